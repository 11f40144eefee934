// Multi-GPU boundary: replicas of the path over the GPUs of one node.
// The reference has no distributed layer at all; what exists here is the minimum the
// partitioning of independent predict() calls needs -- broadcast of inputs, all-gather of
// labels / AutoTune scalars, a max-reduce for timing -- on RCCL over xGMI, reached from the
// Python host through this C ABI (no PyTorch).  librccl is opened lazily (dlopen) so that
// single-GPU users and the CPU-side checks never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "handle.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t,
                            ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // The ROCm installation's RCCL by absolute path first: a bare soname would resolve to
    // whatever librccl.so.1 the process already holds (PyTorch wheels bundle their own,
    // bound to their own HIP runtime -- its communicators cannot use this library's
    // streams and buffers).
    std::string root = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
    const std::string names[] = {root + "/lib/librccl.so.1", "/opt/rocm/lib/librccl.so.1",
                                 "librccl.so.1", "librccl.so"};
    for (const std::string& nm : names) {
      r.lib = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.err = std::string("cannot open librccl: ") + dlerror();
      return;
    }
    auto sym = [&](const char* nm) {
      void* p = dlsym(r.lib, nm);
      if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + nm;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &r;
}

}  // namespace

struct sc_comm_s {
  sc_handle h = nullptr;  // device + stream the collectives run on (not owned)
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  DevBuf send, recv;      // device staging of the host buffers
  std::string err;
};

static int comm_fail(sc_comm c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define SC_NCCL(c, call)                                                                 \
  do {                                                                                   \
    ncclResult_t r_ = (call);                                                            \
    if (r_ != ncclSuccess)                                                               \
      return comm_fail(c, SC_ERR_HIP, std::string(#call) + ": " + rccl()->GetErrorString(r_)); \
  } while (0)
#define SC_CHIP(c, call)                                                                 \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return comm_fail(c, e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP,           \
                       std::string(#call) + ": " + hipGetErrorString(e_));               \
  } while (0)

static int stage(sc_comm c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return SC_OK;
  SC_CHIP(c, hipStreamSynchronize(c->h->stream));
  if (b.p) SC_CHIP(c, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  const size_t want = std::max<size_t>(bytes, 1 << 16);
  SC_CHIP(c, hipMalloc(&b.p, want));
  b.bytes = want;
  return SC_OK;
}

extern "C" int sc_comm_available(void) {
  Rccl* r = rccl();
  return (r->lib && r->err.empty()) ? 1 : 0;
}

extern "C" int sc_comm_unique_id(unsigned char* id) {
  if (!id) return SC_ERR_INVALID;
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return SC_ERR_UNSUPPORTED;
  ncclUniqueId uid;
  if (r->GetUniqueId(&uid) != ncclSuccess) return SC_ERR_HIP;
  static_assert(sizeof(uid) == SC_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id, &uid, sizeof(uid));
  return SC_OK;
}

extern "C" int sc_comm_init_rank(sc_handle h, int world_size, int rank,
                                 const unsigned char* id, sc_comm* out) {
  if (!h || !out || !id || world_size < 1 || rank < 0 || rank >= world_size)
    return SC_ERR_INVALID;
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return fail(h, SC_ERR_UNSUPPORTED, r->err);
  SC_HIP(h, hipSetDevice(h->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  ncclResult_t rc = r->CommInitRank(&comm, world_size, uid, rank);
  if (rc != ncclSuccess)
    return fail(h, SC_ERR_HIP, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  sc_comm c = new sc_comm_s;
  c->h = h;
  c->comm = comm;
  c->rank = rank;
  c->world = world_size;
  *out = c;
  return SC_OK;
}

extern "C" int sc_comm_init_all(sc_handle* handles, int ndev, sc_comm* out) {
  if (!handles || !out || ndev < 1) return SC_ERR_INVALID;
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return fail(handles[0], SC_ERR_UNSUPPORTED, r->err);
  std::vector<int> devs(ndev);
  for (int i = 0; i < ndev; ++i) {
    if (!handles[i]) return SC_ERR_INVALID;
    devs[i] = handles[i]->device;
  }
  std::vector<ncclComm_t> comms(ndev, nullptr);
  ncclResult_t rc = r->CommInitAll(comms.data(), ndev, devs.data());
  if (rc != ncclSuccess)
    return fail(handles[0], SC_ERR_HIP,
                std::string("ncclCommInitAll: ") + r->GetErrorString(rc));
  for (int i = 0; i < ndev; ++i) {
    sc_comm c = new sc_comm_s;
    c->h = handles[i];
    c->comm = comms[i];
    c->rank = i;
    c->world = ndev;
    out[i] = c;
  }
  return SC_OK;
}

extern "C" int sc_comm_destroy(sc_comm c) {
  if (!c) return SC_OK;
  hipSetDevice(c->h->device);
  hipStreamSynchronize(c->h->stream);
  if (c->comm) rccl()->CommDestroy(c->comm);
  if (c->send.p) hipFree(c->send.p);
  if (c->recv.p) hipFree(c->recv.p);
  delete c;
  return SC_OK;
}

extern "C" int sc_comm_rank(sc_comm c) { return c ? c->rank : -1; }
extern "C" int sc_comm_size(sc_comm c) { return c ? c->world : 0; }
extern "C" const char* sc_comm_last_error(sc_comm c) { return c ? c->err.c_str() : ""; }

extern "C" int sc_comm_broadcast(sc_comm c, void* buf, size_t bytes, int root) {
  if (!c || (!buf && bytes) || root < 0 || root >= c->world) return SC_ERR_INVALID;
  if (bytes == 0) return SC_OK;
  hipStream_t s = c->h->stream;
  SC_CHIP(c, hipSetDevice(c->h->device));
  SC_TRY(stage(c, c->send, bytes));
  if (c->rank == root)
    SC_CHIP(c, hipMemcpyAsync(c->send.p, buf, bytes, hipMemcpyHostToDevice, s));
  SC_NCCL(c, rccl()->Broadcast(c->send.p, c->send.p, bytes, ncclUint8, root, c->comm, s));
  if (c->rank != root)
    SC_CHIP(c, hipMemcpyAsync(buf, c->send.p, bytes, hipMemcpyDeviceToHost, s));
  SC_CHIP(c, hipStreamSynchronize(s));
  return SC_OK;
}

extern "C" int sc_comm_allgather(sc_comm c, const void* send, void* recv, size_t bytes) {
  if (!c || ((!send || !recv) && bytes)) return SC_ERR_INVALID;
  if (bytes == 0) return SC_OK;
  hipStream_t s = c->h->stream;
  SC_CHIP(c, hipSetDevice(c->h->device));
  SC_TRY(stage(c, c->send, bytes));
  SC_TRY(stage(c, c->recv, bytes * (size_t)c->world));
  SC_CHIP(c, hipMemcpyAsync(c->send.p, send, bytes, hipMemcpyHostToDevice, s));
  SC_NCCL(c, rccl()->AllGather(c->send.p, c->recv.p, bytes, ncclUint8, c->comm, s));
  SC_CHIP(c, hipMemcpyAsync(recv, c->recv.p, bytes * (size_t)c->world,
                            hipMemcpyDeviceToHost, s));
  SC_CHIP(c, hipStreamSynchronize(s));
  return SC_OK;
}

extern "C" int sc_comm_allreduce_max(sc_comm c, double* values, int count) {
  if (!c || !values || count < 1) return SC_ERR_INVALID;
  hipStream_t s = c->h->stream;
  const size_t bytes = (size_t)count * sizeof(double);
  SC_CHIP(c, hipSetDevice(c->h->device));
  SC_TRY(stage(c, c->send, bytes));
  SC_CHIP(c, hipMemcpyAsync(c->send.p, values, bytes, hipMemcpyHostToDevice, s));
  SC_NCCL(c, rccl()->AllReduce(c->send.p, c->send.p, (size_t)count, ncclFloat64, ncclMax,
                               c->comm, s));
  SC_CHIP(c, hipMemcpyAsync(values, c->send.p, bytes, hipMemcpyDeviceToHost, s));
  SC_CHIP(c, hipStreamSynchronize(s));
  return SC_OK;
}

// every rank's stream has drained and every rank has arrived
extern "C" int sc_comm_barrier(sc_comm c) {
  double v = 0.0;
  return sc_comm_allreduce_max(c, &v, 1);
}
