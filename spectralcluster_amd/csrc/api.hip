// C ABI (include/spectralcluster_amd.h) and host-side orchestration: device arena,
// the refinement / Laplacian / eigen / k-means pipeline, the block-Lanczos control
// loop, the eigengap scalar loop and the MT19937 stream that seeds k-means++.
// Host code only decides and launches; every O(n) or larger computation runs in
// the HIP kernels of this library.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sc_internal.h"

using namespace sc;

// ------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct sc_handle_s {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // current problem
  int n = 0, d = 0, ldn = 0, ldx = 0;
  bool have_x = false, have_affinity = false;
  bool have_cropval = false;  // cropval = CropDiagonal fill values of A0 (affinity GEMM epilogue)
  int n_vec = 0;          // eigenvector columns resident in E
  // matrices
  DevBuf X, Xn, A0, B1, B2;
  // n-vectors
  DevBuf rowmax, rowsum, cvec, pvec, tvec, deg, dvec, cut, rmpart, splitk, tilemap;
  DevBuf cropval, statp;  // fused GEMM row statistics: result + per-tile partials
  // constraints: Cq (resident constraint matrix), Neumann-product work matrices, flag word
  DevBuf Cq, cp[5], symflag;
  bool have_constraint = false, constraint_symmetric = false, constraint_applied = false;
  bool affinity_symmetric = true;
  bool affinity_from_embeddings = false;  // symflag[1] then says whether a row was NaN
  int qn = 0;
  int tilemap_nt = 0;     // tile-grid size the resident tilemap was built for
  DevBuf blurw;           // device copy of the blur weights
  // eigen workspace
  DevBuf Q, Q2, Vs, W, partial, T, Y, Yt, theta, resid, G, Rinv, Hbuf, hsq, colnorm,
      flags;
  DevBuf E, Ek, Eio;      // eigenvectors (col-major), renormed copy, row-major I/O staging
  // general (non-symmetric) eigen path: right scaling, Im(theta), complex Ritz vectors
  // (column-major), residual partials, restart codes, dense Laplacian scratch
  DevBuf crvec, thetai, Vre, Vim, gpart, gsrc, genL;
  const double* vs_scale = nullptr;  // Vs = vs_scale .* V in orthonormalize (default cvec)
  DevBuf ahc_size, ahc_chain, ahc_Z, ahc_lab, ahc_cent;  // size reduction (AHC) scratch
  DevBuf fb_part, fb_small, fb_x, fb_cent, fb_int;      // fallback decisions scratch
  // k-means workspace
  DevBuf kXc, kxsq, kclosest, kcand, kenorm, krnd, kcent, klab32, klab64, kinfo;
  // pinned host scratch
  double* h_theta = nullptr;  // 3 * kLdq doubles (theta, resid, Im theta)
  int* h_flags = nullptr;
  hipEvent_t ev[48];
  int nev = 0;
};

static constexpr int kMaxCols = 128;  // eigenvector columns the arena can hold
// Leading dimension of the n x n matrices.  A row stride that is a multiple of 4 KiB maps
// the 128 rows of an operand panel onto the same few L2 sets (the GEMM reads one 128-byte
// line per row and K-tile): such strides get one extra 128-byte line.
static inline int matrix_ld(int n) {
  int ld = round_up(n, 16);
  if (ld % 512 == 0) ld += 16;
  return ld;
}

// eigenvectors are column-major on the device: column j at E + j * lde, lde = round_up(n, 16)

#define SC_HIP(h, call)                                                         \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
      return e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP;               \
    }                                                                           \
  } while (0)

#define SC_TRY(expr)                \
  do {                              \
    int rc_ = (expr);               \
    if (rc_ != SC_OK) return rc_;   \
  } while (0)

static int fail(sc_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

static int grow(sc_handle h, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return SC_OK;
  if (b.p) {
    SC_HIP(h, hipStreamSynchronize(h->stream));
    SC_HIP(h, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  SC_HIP(h, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return SC_OK;
}
template <typename T>
static T* ptr(const DevBuf& b) {
  return reinterpret_cast<T*>(b.p);
}

static int ensure_matrices(sc_handle h, int n, int d) {
  const size_t ldn = matrix_ld(n);
  const size_t nn = (size_t)n * ldn * sizeof(double);
  SC_TRY(grow(h, h->A0, nn));
  SC_TRY(grow(h, h->B1, nn));
  SC_TRY(grow(h, h->B2, nn));
  if (d > 0) {
    const size_t ldx = round_up(d, 16);
    SC_TRY(grow(h, h->X, (size_t)n * ldx * sizeof(double)));
    SC_TRY(grow(h, h->Xn, (size_t)n * ldx * sizeof(double)));
  }
  const size_t nv = (size_t)round_up(n, 16) * sizeof(double);
  SC_TRY(grow(h, h->rowmax, nv));
  SC_TRY(grow(h, h->rowsum, nv));
  SC_TRY(grow(h, h->cvec, nv));
  SC_TRY(grow(h, h->pvec, nv));
  SC_TRY(grow(h, h->tvec, nv));
  SC_TRY(grow(h, h->deg, nv));
  SC_TRY(grow(h, h->splitk, gemm_splitk_workspace_bytes()));
  SC_TRY(grow(h, h->dvec, nv));
  SC_TRY(grow(h, h->cut, nv));
  SC_TRY(grow(h, h->cropval, nv));
  SC_TRY(grow(h, h->statp, (size_t)2 * n * gemm_tile_dim(n) * sizeof(double)));
  SC_TRY(grow(h, h->rmpart, (size_t)n * blur_tile_columns(n, 8) * sizeof(double)));
  SC_TRY(grow(h, h->blurw, (2 * SC_MAX_BLUR_RADIUS + 1) * sizeof(double)));
  return SC_OK;
}

// (ti, tj) order of the symmetric GEMM tiles for problems of n rows (cached per handle)
static int ensure_tilemap(sc_handle h, int n) {
  const int nt = gemm_tile_dim(n);
  if (h->tilemap_nt == nt) return SC_OK;
  std::vector<int2> map;
  gemm_build_sym_tilemap(nt, &map);
  SC_TRY(grow(h, h->tilemap, map.size() * sizeof(int2)));
  SC_HIP(h, hipMemcpyAsync(h->tilemap.p, map.data(), map.size() * sizeof(int2),
                           hipMemcpyHostToDevice, h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));  // `map` is a local
  h->tilemap_nt = nt;
  return SC_OK;
}

static int ensure_eig(sc_handle h, int n) {
  const size_t nq = (size_t)n * kLdq * sizeof(double);
  SC_TRY(grow(h, h->Q, nq));
  SC_TRY(grow(h, h->Q2, nq));
  SC_TRY(grow(h, h->Vs, (size_t)n * kEigBlock * sizeof(double)));
  SC_TRY(grow(h, h->W, (size_t)n * kEigBlock * sizeof(double)));
  SC_TRY(grow(h, h->partial, (size_t)kProjBlocks * kLdq * kEigBlock * sizeof(double)));
  SC_TRY(grow(h, h->T, (size_t)kLdq * kLdq * sizeof(double)));
  SC_TRY(grow(h, h->Y, (size_t)kLdq * kLdq * sizeof(double)));
  SC_TRY(grow(h, h->Yt, (size_t)kLdq * kLdq * sizeof(double)));
  SC_TRY(grow(h, h->theta, kLdq * sizeof(double)));
  SC_TRY(grow(h, h->resid, kLdq * sizeof(double)));
  SC_TRY(grow(h, h->G, 256 * sizeof(double)));
  SC_TRY(grow(h, h->Rinv, 256 * sizeof(double)));
  SC_TRY(grow(h, h->Hbuf, (size_t)kLdq * kEigBlock * sizeof(double)));
  SC_TRY(grow(h, h->hsq, 16 * sizeof(double)));
  SC_TRY(grow(h, h->colnorm, (size_t)kProjBlocks * kMaxVectors * sizeof(double)));
  SC_TRY(grow(h, h->flags, 16 * sizeof(int)));
  SC_TRY(grow(h, h->E, (size_t)round_up(n, 16) * kMaxCols * sizeof(double)));
  SC_TRY(grow(h, h->Eio, (size_t)n * kMaxCols * sizeof(double)));
  return SC_OK;
}

static int ensure_gen(sc_handle h, int n) {
  const size_t ldv = round_up(n, 16);
  const size_t nv = ldv * sizeof(double);
  SC_TRY(grow(h, h->crvec, nv));
  SC_TRY(grow(h, h->thetai, kLdq * sizeof(double)));
  SC_TRY(grow(h, h->Vre, ldv * kGenMax * sizeof(double)));
  SC_TRY(grow(h, h->Vim, ldv * kGenMax * sizeof(double)));
  SC_TRY(grow(h, h->gpart, (size_t)gen_residual_blocks(n) * 32 * sizeof(double)));
  SC_TRY(grow(h, h->gsrc, 16 * sizeof(int)));
  SC_TRY(grow(h, h->genL, (size_t)kGenMax * kGenMax * sizeof(double)));
  return SC_OK;
}

static int ensure_kmeans(sc_handle h, int n) {
  SC_TRY(grow(h, h->Ek, (size_t)round_up(n, 16) * kMaxCols * sizeof(double)));
  SC_TRY(grow(h, h->Eio, (size_t)n * kMaxCols * sizeof(double)));
  SC_TRY(grow(h, h->kXc, (size_t)n * kMaxVectors * sizeof(double)));
  SC_TRY(grow(h, h->kxsq, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->kclosest, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->kcand, (size_t)8 * n * sizeof(double)));
  SC_TRY(grow(h, h->kenorm, (size_t)n * sizeof(double)));
  SC_TRY(grow(h, h->krnd, 1024 * sizeof(double)));
  SC_TRY(grow(h, h->kcent, (size_t)kMaxVectors * kMaxVectors * sizeof(double)));
  SC_TRY(grow(h, h->klab32, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->klab64, (size_t)n * sizeof(long long)));
  SC_TRY(grow(h, h->kinfo, 16 * sizeof(int)));
  return SC_OK;
}

static int check_last(sc_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return SC_ERR_HIP;
  }
  return SC_OK;
}

// ------------------------------------------------------------------------------
// library / device
// ------------------------------------------------------------------------------
extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" int sc_struct_sizes(int* config_bytes, int* diag_bytes) {
  if (config_bytes) *config_bytes = (int)sizeof(sc_config);
  if (diag_bytes) *diag_bytes = (int)sizeof(sc_diag);
  return SC_OK;
}

extern "C" int sc_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

extern "C" int sc_device_info(int device, char* name, int name_len, char* arch,
                              int arch_len, int* compute_units,
                              int64_t* total_mem_bytes) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SC_ERR_HIP;
  if (name && name_len > 0) snprintf(name, name_len, "%s", prop.name);
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_mem_bytes) *total_mem_bytes = (int64_t)prop.totalGlobalMem;
  return SC_OK;
}

extern "C" int sc_create(int device, sc_handle* out) {
  if (!out) return SC_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SC_ERR_HIP;
  if (device < 0 || device >= count) return SC_ERR_INVALID;
  sc_handle h = new sc_handle_s();
  h->device = device;
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return SC_ERR_HIP;
  }
  for (int i = 0; i < 48; ++i) {
    if (hipEventCreate(&h->ev[i]) != hipSuccess) {
      delete h;
      return SC_ERR_HIP;
    }
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&h->h_theta), 3 * kLdq * sizeof(double)) !=
          hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&h->h_flags), 16 * sizeof(int)) !=
          hipSuccess) {
    delete h;
    return SC_ERR_HIP;
  }
  *out = h;
  return SC_OK;
}

extern "C" int sc_destroy(sc_handle h) {
  if (!h) return SC_OK;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  DevBuf* bufs[] = {&h->X,     &h->Xn,    &h->A0,     &h->B1,      &h->B2,    &h->rowmax,
                    &h->rowsum, &h->cvec,  &h->pvec,   &h->tvec,    &h->deg,   &h->blurw, &h->dvec, &h->cut, &h->rmpart, &h->splitk, &h->tilemap, &h->cropval, &h->statp, &h->crvec, &h->thetai, &h->Vre, &h->Vim, &h->gpart, &h->gsrc, &h->genL, &h->ahc_size, &h->ahc_chain, &h->ahc_Z, &h->ahc_lab, &h->ahc_cent, &h->fb_part, &h->fb_small, &h->fb_x, &h->fb_cent, &h->fb_int, &h->Cq, &h->cp[0], &h->cp[1], &h->cp[2], &h->cp[3], &h->cp[4], &h->symflag,
                    &h->Q,     &h->Q2,    &h->Vs,     &h->W,       &h->partial, &h->T,
                    &h->Y,     &h->Yt,    &h->theta,  &h->resid,   &h->G,     &h->Rinv,
                    &h->Hbuf,  &h->hsq,   &h->colnorm, &h->flags,  &h->E,     &h->Ek,   &h->Eio,
                    &h->kXc,   &h->kxsq,  &h->kclosest, &h->kcand, &h->kenorm, &h->krnd,
                    &h->kcent, &h->klab32, &h->klab64, &h->kinfo};
  for (DevBuf* b : bufs)
    if (b->p) hipFree(b->p);
  for (int i = 0; i < 48; ++i) hipEventDestroy(h->ev[i]);
  if (h->h_theta) hipHostFree(h->h_theta);
  if (h->h_flags) hipHostFree(h->h_flags);
  hipStreamDestroy(h->stream);
  delete h;
  return SC_OK;
}

extern "C" const char* sc_last_error(sc_handle h) { return h ? h->err.c_str() : ""; }

extern "C" int sc_synchronize(sc_handle h) {
  if (!h) return SC_ERR_INVALID;
  SC_HIP(h, hipStreamSynchronize(h->stream));
  return SC_OK;
}

extern "C" int sc_reserve(sc_handle h, int n_max, int d_max) {
  if (!h || n_max <= 0 || d_max < 0) return SC_ERR_INVALID;
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n_max, d_max));
  SC_TRY(ensure_eig(h, n_max));
  SC_TRY(ensure_kmeans(h, n_max));
  return SC_OK;
}

// numpy pairwise sum for short arrays (n < 128): 8 running sums, then the tail
static double numpy_sum_short(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

extern "C" int sc_gaussian_weights(double sigma, int32_t* radius, double* weights) {
  if (!radius || !weights || !(sigma >= 0.0)) return SC_ERR_INVALID;
  if (sigma <= 1e-15) {  // gaussian_filter skips the axis: plain copy
    *radius = 0;
    weights[0] = 1.0;
    return SC_OK;
  }
  const int r = (int)(4.0 * sigma + 0.5);  // truncate = 4.0
  if (r > SC_MAX_BLUR_RADIUS) return SC_ERR_UNSUPPORTED;
  const double s2 = sigma * sigma;
  for (int x = -r; x <= r; ++x) weights[x + r] = std::exp(-0.5 / s2 * (double)(x * x));
  const double sum = numpy_sum_short(weights, 2 * r + 1);
  for (int i = 0; i < 2 * r + 1; ++i) weights[i] /= sum;
  *radius = r;
  return SC_OK;
}

extern "C" int sc_config_default(sc_config* cfg) {
  if (!cfg) return SC_ERR_INVALID;
  memset(cfg, 0, sizeof(*cfg));
  sc_gaussian_weights(1.0, &cfg->blur_radius, cfg->blur_weights);
  cfg->p_percentile = 0.95;
  cfg->soft_multiplier = 0.01;
  cfg->threshold_type = SC_THRESHOLD_ROW_MAX;
  cfg->symmetrize_type = SC_SYMMETRIZE_MAX;
  cfg->laplacian_type = SC_LAPLACIAN_NONE;
  cfg->stop_eigenvalue = 1e-2;
  cfg->eigengap_type = SC_EIGENGAP_RATIO;
  cfg->max_iter = 300;
  cfg->constraint_name = SC_CONSTRAINT_NONE;
  cfg->integration_type = SC_INTEGRATION_MAX;
  cfg->constraint_alpha = 0.6;  // constraint.py:41
  return SC_OK;
}

// ------------------------------------------------------------------------------
// E2: eigengap (reference utils.py:74-130)
// ------------------------------------------------------------------------------
static void eigengap_core(const double* w, int count, int max_clusters,
                          double stop_eigenvalue, int eigengap_type, int descend,
                          double wmax, int* n_clusters, double* max_delta) {
  const double eps = 1e-10;  // utils.py:7
  double best = 0.0;
  int best_k = 0;
  int end = count;
  if (max_clusters > 0 && max_clusters + 1 < end) end = max_clusters + 1;
  if (descend) {
    for (int i = 1; i < end; ++i) {
      if (w[i - 1] < stop_eigenvalue) break;
      const double d = eigengap_type == SC_EIGENGAP_RATIO ? w[i - 1] / (w[i] + eps)
                                                          : (w[i - 1] - w[i]) / wmax;
      if (d > best) { best = d; best_k = i; }
    }
  } else {
    for (int i = 1; i < end - 1; ++i) {
      const double d = eigengap_type == SC_EIGENGAP_RATIO ? w[i + 1] / (w[i] + eps)
                                                          : (w[i + 1] - w[i]) / wmax;
      if (d > best) { best = d; best_k = i + 1; }
    }
  }
  *n_clusters = best_k;
  *max_delta = best;
}

extern "C" int sc_eigengap(const double* w, int count, int max_clusters,
                           double stop_eigenvalue, int eigengap_type, int descend,
                           int* n_clusters, double* max_delta) {
  if (!w || count < 0 || !n_clusters || !max_delta) return SC_ERR_INVALID;
  if (eigengap_type != SC_EIGENGAP_RATIO && eigengap_type != SC_EIGENGAP_NORMALIZED_DIFF)
    return SC_ERR_INVALID;
  double wmax = 0.0;
  if (count > 0) {
    wmax = w[0];
    for (int i = 1; i < count; ++i) wmax = w[i] > wmax ? w[i] : wmax;  // np.max(eigenvalues)
  }
  eigengap_core(w, count, max_clusters, stop_eigenvalue, eigengap_type, descend, wmax,
                n_clusters, max_delta);
  return SC_OK;
}

// ------------------------------------------------------------------------------
// MT19937 as numpy's legacy RandomState(seed) drives it (k-means++ seeding)
// ------------------------------------------------------------------------------
namespace {
struct Mt19937 {
  uint32_t mt[624];
  int pos;
  explicit Mt19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i)
      mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    pos = 624;
  }
  void twist() {
    for (int k = 0; k < 624; ++k) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      mt[k] = v;
    }
    pos = 0;
  }
  uint32_t next_u32() {
    if (pos >= 624) twist();
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  double next_double() {
    const uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};
}  // namespace

extern "C" int sc_random_state_doubles(uint32_t seed, int count, double* out) {
  if (!out || count < 0) return SC_ERR_INVALID;
  Mt19937 rng(seed);
  for (int i = 0; i < count; ++i) out[i] = rng.next_double();
  return SC_OK;
}

extern "C" int sc_uniform_choice(int n, double u) {
  if (n <= 0) return SC_ERR_INVALID;
  const double p = 1.0 / (double)n;
  std::vector<double> cdf(n);
  double run = 0.0;
  for (int i = 0; i < n; ++i) {
    run += p;
    cdf[i] = run;
  }
  const double last = cdf[n - 1];
  for (int i = 0; i < n; ++i)
    if (cdf[i] / last > u) return i;  // searchsorted(..., side="right")
  return n - 1;
}

// ------------------------------------------------------------------------------
// data movement helpers
// ------------------------------------------------------------------------------
static int h2d_matrix(sc_handle h, const double* src, int rows, int cols, double* dst,
                      int ld) {
  SC_HIP(h, hipMemcpy2DAsync(dst, (size_t)ld * sizeof(double), src,
                             (size_t)cols * sizeof(double), (size_t)cols * sizeof(double),
                             rows, hipMemcpyHostToDevice, h->stream));
  return SC_OK;
}
static int d2h_matrix(sc_handle h, const double* src, int ld, int rows, int cols,
                      double* dst) {
  SC_HIP(h, hipMemcpy2DAsync(dst, (size_t)cols * sizeof(double), src,
                             (size_t)ld * sizeof(double), (size_t)cols * sizeof(double),
                             rows, hipMemcpyDeviceToHost, h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  return SC_OK;
}

static int validate_config(sc_handle h, const sc_config* cfg) {
  if (!cfg) return fail(h, SC_ERR_INVALID, "config is NULL");
  if (cfg->n_ops < 0 || cfg->n_ops > SC_MAX_OPS)
    return fail(h, SC_ERR_INVALID, "n_ops out of range");
  for (int i = 0; i < cfg->n_ops; ++i)
    if (cfg->ops[i] < SC_OP_CROP_DIAGONAL || cfg->ops[i] > SC_OP_ROW_WISE_NORMALIZE)
      return fail(h, SC_ERR_INVALID, "Unknown refinement operation");
  if (cfg->blur_radius < 0 || cfg->blur_radius > SC_MAX_BLUR_RADIUS)
    return fail(h, SC_ERR_UNSUPPORTED, "gaussian blur radius > 32");
  if (cfg->laplacian_type < SC_LAPLACIAN_NONE || cfg->laplacian_type > SC_LAPLACIAN_GRAPH_CUT)
    return fail(h, SC_ERR_INVALID, "laplacian_type must be a LaplacianType");
  if (cfg->eigengap_type != SC_EIGENGAP_RATIO &&
      cfg->eigengap_type != SC_EIGENGAP_NORMALIZED_DIFF)
    return fail(h, SC_ERR_INVALID, "eigengap_type must be a EigenGapType");
  if (cfg->symmetrize_type != SC_SYMMETRIZE_MAX &&
      cfg->symmetrize_type != SC_SYMMETRIZE_AVERAGE)
    return fail(h, SC_ERR_INVALID, "Unsupported symmetrize_type.");
  if (cfg->threshold_type != SC_THRESHOLD_ROW_MAX &&
      cfg->threshold_type != SC_THRESHOLD_PERCENTILE)
    return fail(h, SC_ERR_INVALID, "Unsupported thresholding_type");
  return SC_OK;
}

// run one refinement op `in` -> `out` (distinct buffers)
static int run_refine_op(sc_handle h, int op, const sc_config* cfg, const double* in,
                         double* out, int n, int ld) {
  hipStream_t s = h->stream;
  switch (op) {
    case SC_OP_CROP_DIAGONAL:
      launch_crop_diagonal(s, in, out, n, ld);
      break;
    case SC_OP_GAUSSIAN_BLUR:
      if (cfg->blur_radius > 0)
        SC_HIP(h, hipMemcpyAsync(h->blurw.p, cfg->blur_weights,
                                 (2 * cfg->blur_radius + 1) * sizeof(double),
                                 hipMemcpyHostToDevice, s));
      launch_gaussian_blur(s, in, out, n, ld, cfg->blur_radius, ptr<double>(h->blurw));
      break;
    case SC_OP_ROW_WISE_THRESHOLD:
      if (cfg->threshold_type == SC_THRESHOLD_PERCENTILE) {
        launch_cut_percentile(s, in, n, ld, cfg->p_percentile, ptr<double>(h->cut),
                              cfg->preserve_diagonal);
        launch_row_threshold_cut(s, in, out, n, ld, ptr<double>(h->cut), cfg->soft_multiplier,
                                 cfg->binarize, cfg->preserve_diagonal);
      } else {
        launch_row_threshold(s, in, out, n, ld, cfg->p_percentile, cfg->soft_multiplier,
                             cfg->binarize, cfg->preserve_diagonal);
      }
      break;
    case SC_OP_SYMMETRIZE:
      launch_symmetrize(s, in, out, n, ld, cfg->symmetrize_type);
      break;
    case SC_OP_DIFFUSE:
      SC_TRY(ensure_tilemap(h, n));
      launch_gemm_nt(s, in, ld, in, ld, out, ld, n, n, n, kEpiNone, true, ptr<double>(h->splitk),
                     ptr<int2>(h->tilemap));
      break;
    case SC_OP_ROW_WISE_NORMALIZE:
      launch_row_normalize(s, in, out, n, ld);
      break;
    default:
      return fail(h, SC_ERR_INVALID, "Unknown refinement operation");
  }
  return check_last(h, "refinement kernel launch");
}

// ------------------------------------------------------------------------------
// N3: constraints (reference constraint.py:95-164)
// ------------------------------------------------------------------------------
// exact symmetry of a resident (n, ld) matrix; one 4-byte D2H + stream sync
static int device_is_symmetric(sc_handle h, const double* m, int n, int ld, bool* out) {
  SC_TRY(grow(h, h->symflag, 16));
  const int one = 1;
  int result = 0;
  SC_HIP(h, hipMemcpyAsync(h->symflag.p, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
  launch_symmetry_flag(h->stream, m, n, ld, ptr<int>(h->symflag));
  SC_HIP(h, hipMemcpyAsync(&result, h->symflag.p, sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  *out = result != 0;
  return SC_OK;
}

// ConstraintPropagation.adjust_affinity (constraint.py:138-164):  out may alias a.
//   P = alpha D^-1/2 A D^-1/2,  T = (I - P)^-1 = prod_{j>=0} (I + P^(2^j))  (rho(P) <= |alpha|),
//   F = (1 - alpha)^2 T Q T,  out = F > 0 ? 1 - (1 - F)(1 - A) : (1 + F) A.
// Every product runs on the fp64 MFMA GEMM (C = X Y^T).  For a symmetric A all factors
// are symmetric and commute, so squarings and T updates compute the upper tile triangle
// only; a general A carries explicit transposes instead.
static int constraint_propagation(sc_handle h, const double* a, bool sym_a, const double* q,
                                  bool sym_q, double alpha, double* out, int n, int ld) {
  hipStream_t s = h->stream;
  const double mag = fabs(alpha);
  if (!(mag < 1.0))
    return fail(h, SC_ERR_UNSUPPORTED,
                "ConstraintPropagation on the device needs |constraint_propagation_alpha| < 1");
  // factors (I + P^(2^j)), j = 0 .. steps-1, leave a remainder of P^(2^steps)
  int steps = 0;
  if (mag > 0.0) {
    double rem = mag;
    while (rem > 1e-18 && steps < 18) {
      rem *= rem;
      ++steps;
    }
    if (rem > 1e-18)
      return fail(h, SC_ERR_UNSUPPORTED,
                  "constraint_propagation_alpha too close to 1 for the Neumann product");
  }
  const size_t bytes = (size_t)n * ld * sizeof(double);
  for (int i = 0; i < 5; ++i) SC_TRY(grow(h, h->cp[i], bytes));
  SC_TRY(ensure_tilemap(h, n));
  double* P = ptr<double>(h->cp[0]);
  double* T = ptr<double>(h->cp[1]);
  double* Pn = ptr<double>(h->cp[2]);
  double* Tn = ptr<double>(h->cp[3]);
  double* X = ptr<double>(h->cp[4]);  // transposes (general A), then T Q^T
  double* ws = ptr<double>(h->splitk);
  const int2* tm = ptr<int2>(h->tilemap);
  launch_row_stats(s, a, n, ld, ptr<double>(h->cut), ptr<double>(h->deg));  // deg = rowsum
  launch_cp_prepare(s, a, ptr<double>(h->deg), alpha, P, T, n, ld);          // T = I + P
  for (int j = 1; j < steps; ++j) {
    // Pn = P P
    if (sym_a) {
      launch_gemm_nt(s, P, ld, P, ld, Pn, ld, n, n, n, kEpiNone, true, ws, tm);
    } else {
      launch_transpose(s, P, X, n, ld);
      launch_gemm_nt(s, P, ld, X, ld, Pn, ld, n, n, n, kEpiNone, false, ws, nullptr);
    }
    std::swap(P, Pn);
    // Tn = T + T P
    if (sym_a) {
      launch_gemm_nt(s, T, ld, P, ld, Tn, ld, n, n, n, kEpiAdd, true, ws, tm, nullptr, T);
    } else {
      launch_transpose(s, P, X, n, ld);
      launch_gemm_nt(s, T, ld, X, ld, Tn, ld, n, n, n, kEpiAdd, false, ws, nullptr, nullptr, T);
    }
    std::swap(T, Tn);
  }
  // G^T = T^T Q^T  (X),  T Q T = T (G^T)^T  (Pn)
  const double* Tt = T;
  if (!sym_a) {
    launch_transpose(s, T, Tn, n, ld);
    Tt = Tn;
  }
  launch_gemm_nt(s, Tt, ld, q, ld, X, ld, n, n, n, kEpiNone, false, ws, nullptr);
  const bool sym_f = sym_a && sym_q;
  launch_gemm_nt(s, T, ld, X, ld, Pn, ld, n, n, n, kEpiNone, sym_f, ws, sym_f ? tm : nullptr);
  launch_cp_adjust(s, Pn, a, (1.0 - alpha) * (1.0 - alpha), out, n, ld);
  return check_last(h, "constraint propagation launch");
}

// cfg's constraint operator on `a` with the resident constraint matrix; out may alias a
static int adjust_affinity(sc_handle h, const sc_config* cfg, const double* a, bool sym_a,
                           double* out, int n, int ld) {
  if (cfg->constraint_name == SC_CONSTRAINT_AFFINITY_INTEGRATION) {
    if (cfg->integration_type != SC_INTEGRATION_MAX &&
        cfg->integration_type != SC_INTEGRATION_AVERAGE)
      return fail(h, SC_ERR_INVALID, "Unsupported integration type");
    launch_affinity_integration(h->stream, a, ptr<double>(h->Cq), out, n, ld,
                                cfg->integration_type);
    return check_last(h, "affinity integration launch");
  }
  if (cfg->constraint_name == SC_CONSTRAINT_PROPAGATION)
    return constraint_propagation(h, a, sym_a, ptr<double>(h->Cq), h->constraint_symmetric,
                                  cfg->constraint_alpha, out, n, ld);
  return fail(h, SC_ERR_INVALID, "constraint_name must be a ConstraintName");
}

extern "C" int sc_set_constraint(sc_handle h, const double* q, int n) {
  if (!h) return SC_ERR_INVALID;
  if (!q || n <= 0) return fail(h, SC_ERR_INVALID, "constraint matrix must be (n, n)");
  SC_HIP(h, hipSetDevice(h->device));
  const int ld = matrix_ld(n);
  SC_TRY(grow(h, h->Cq, (size_t)n * ld * sizeof(double)));
  SC_TRY(h2d_matrix(h, q, n, n, ptr<double>(h->Cq), ld));
  SC_TRY(device_is_symmetric(h, ptr<double>(h->Cq), n, ld, &h->constraint_symmetric));
  h->have_constraint = true;
  h->qn = n;
  return SC_OK;
}

extern "C" int sc_clear_constraint(sc_handle h) {
  if (!h) return SC_ERR_INVALID;
  h->have_constraint = false;
  h->qn = 0;
  return SC_OK;
}

static bool constraint_active(sc_handle h, const sc_config* cfg, bool before) {
  return cfg->constraint_name != SC_CONSTRAINT_NONE && h->have_constraint &&
         (cfg->constraint_before_refinement != 0) == before;
}

extern "C" int sc_apply_constraint(sc_handle h, const sc_config* cfg) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  if (!h->have_constraint) return fail(h, SC_ERR_INVALID, "no constraint matrix resident");
  if (cfg->constraint_name == SC_CONSTRAINT_NONE)
    return fail(h, SC_ERR_INVALID, "no constraint operation configured");
  if (h->qn != h->n)
    return fail(h, SC_ERR_INVALID, "affinity and constraint matrix must have the same shape");
  if (h->constraint_applied)
    return fail(h, SC_ERR_INVALID, "the resident affinity is already constraint-adjusted");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, h->n, 0));
  SC_TRY(adjust_affinity(h, cfg, ptr<double>(h->A0), h->affinity_symmetric, ptr<double>(h->A0),
                         h->n, h->ldn));
  h->affinity_symmetric = h->affinity_symmetric && h->constraint_symmetric;
  h->have_cropval = false;
  h->constraint_applied = true;
  h->n_vec = 0;
  return SC_OK;
}

extern "C" int sc_stage_constraint(sc_handle h, const sc_config* cfg, const double* affinity,
                                   const double* q, int n, double* out) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!affinity || !q || !out || n <= 0)
    return fail(h, SC_ERR_INVALID, "affinity and constraint matrix must be (n, n)");
  SC_TRY(sc_set_affinity(h, affinity, n));
  SC_TRY(sc_set_constraint(h, q, n));
  const int rc = sc_apply_constraint(h, cfg);
  sc_clear_constraint(h);
  SC_TRY(rc);
  return d2h_matrix(h, ptr<double>(h->A0), h->ldn, n, n, out);
}

// ------------------------------------------------------------------------------
// embeddings / affinity
// ------------------------------------------------------------------------------
extern "C" int sc_set_embeddings(sc_handle h, const double* x, int n, int d) {
  if (!h) return SC_ERR_INVALID;
  if (!x || n <= 0 || d <= 0) return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, d));
  h->n = n;
  h->d = d;
  h->ldn = matrix_ld(n);
  h->ldx = round_up(d, 16);
  h->have_affinity = h->have_cropval = false;
  h->n_vec = 0;
  SC_TRY(h2d_matrix(h, x, n, d, ptr<double>(h->X), h->ldx));
  SC_HIP(h, hipStreamSynchronize(h->stream));  // caller may reuse x immediately
  h->have_x = true;
  return SC_OK;
}

extern "C" int sc_compute_affinity(sc_handle h) {
  if (!h) return SC_ERR_INVALID;
  if (!h->have_x) return fail(h, SC_ERR_INVALID, "no embeddings resident");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_tilemap(h, h->n));
  SC_TRY(grow(h, h->symflag, 16));
  SC_HIP(h, hipMemsetAsync(ptr<int>(h->symflag) + 1, 0, sizeof(int), h->stream));
  launch_normalize_rows(h->stream, ptr<double>(h->X), h->ldx, h->n, h->d,
                        ptr<double>(h->Xn), ptr<int>(h->symflag) + 1);
  h->affinity_from_embeddings = true;
  // CropDiagonal's fill value (max_{j != i} A_ij, >= 0) comes out of the GEMM epilogue
  GemmRowStats rs{2, ptr<double>(h->statp), nullptr, ptr<double>(h->cropval), nullptr};
  launch_gemm_nt(h->stream, ptr<double>(h->Xn), h->ldx, ptr<double>(h->Xn), h->ldx,
                 ptr<double>(h->A0), h->ldn, h->n, h->n, h->d, kEpiAffinity, true,
                 ptr<double>(h->splitk), ptr<int2>(h->tilemap), &rs);
  SC_TRY(check_last(h, "affinity launch"));
  h->have_affinity = true;
  h->have_cropval = true;
  h->affinity_symmetric = true;
  h->constraint_applied = false;
  h->n_vec = 0;
  return SC_OK;
}

extern "C" int sc_set_affinity(sc_handle h, const double* a, int n) {
  if (!h) return SC_ERR_INVALID;
  if (!a || n <= 0) return fail(h, SC_ERR_INVALID, "affinity must be (n, n)");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  h->n = n;
  h->ldn = matrix_ld(n);
  h->have_x = false;
  h->n_vec = 0;
  SC_TRY(h2d_matrix(h, a, n, n, ptr<double>(h->A0), h->ldn));
  SC_TRY(device_is_symmetric(h, ptr<double>(h->A0), n, h->ldn, &h->affinity_symmetric));
  h->have_affinity = true;
  h->have_cropval = false;
  h->constraint_applied = false;
  h->affinity_from_embeddings = false;
  return SC_OK;
}

// ------------------------------------------------------------------------------
// symmetric top-k eigensolver driver
// ------------------------------------------------------------------------------
struct EigRequest {
  int descend;          // 1: report largest first (w = theta); 0: w = -theta ascending
  int max_clusters;     // 0 = None
  int min_clusters;     // 0 = None
  double stop_eigenvalue;
  int eigengap_type;
  int use_stop;         // stop_eigenvalue only on the descending branch
  double value_tol, vector_tol;
  int max_cycles;
  int fixed_count;      // > 0: plain "count extreme eigenpairs" request (stage API)
  // General path only.  Consumed eigenvalues deep in a dense bulk converge arbitrarily
  // slowly in a small Krylov basis, yet cannot influence the result: only the two values
  // that form the maximum gap (and the normaliser of NormalizedDiff) are held to value_tol;
  // the others must be accurate enough that, with their residual intervals, no other gap
  // can reach the maximum and no comparison with stop_eigenvalue can flip.
  int decision_aware = 0;
};

struct EigDecision {
  bool enough = false;     // basis large enough to take a decision
  bool converged = false;
  int kw = 0;              // eigenvalues reported
  int kvec = 0;            // vectors that must be accurate
  int n_clusters_raw = 0;
  double max_delta = 0.0;
  double max_resid = 0.0;
  bool unsupported = false;
  int fail_kind = 0;       // 1 consumed value, 2 far-end value, 3 vector (trace only)
  int fail_index = -1;
};

// Inspect Ritz values theta[0..m) (descending) + residual estimates.
static EigDecision analyze(const EigRequest& rq, const double* theta, const double* resid,
                           int m, int n, bool exact) {
  EigDecision dc;
  std::vector<double> w(m);
  for (int i = 0; i < m; ++i) w[i] = rq.descend ? theta[i] : -theta[i];
  const double scale = std::max(std::fabs(theta[0]), std::fabs(theta[m - 1]));
  int kw;
  if (rq.fixed_count > 0) {
    kw = std::min(rq.fixed_count, n);
  } else if (rq.max_clusters > 0) {
    kw = std::min(n, rq.max_clusters + 1);
  } else if (rq.descend) {
    // max_clusters None: everything >= stop_eigenvalue, plus the first one below
    int c = 0;
    while (c < m && !(w[c] < rq.stop_eigenvalue)) ++c;
    if (c >= m && m < n) return dc;  // have not reached the stop value yet
    kw = std::min(c + 1, n);
  } else {
    kw = n;  // ascending without max_clusters reads every eigenvalue
  }
  if (kw > m) {
    if (kw > kEigBasisCap / 2 && !exact) dc.unsupported = true;
    return dc;
  }
  dc.enough = true;
  dc.kw = kw;
  if (rq.fixed_count > 0) {
    dc.kvec = kw;
  } else {
    // np.max(eigenvalues) is taken over the WHOLE spectrum (utils.py:110,123): the
    // first value when descending, the far end of the Ritz spectrum when ascending.
    const double wmax = rq.descend ? w[0] : w[m - 1];
    eigengap_core(w.data(), kw, rq.max_clusters, rq.use_stop ? rq.stop_eigenvalue : 0.0,
                  rq.eigengap_type, rq.descend, wmax, &dc.n_clusters_raw, &dc.max_delta);
    dc.kvec = std::max(dc.n_clusters_raw, rq.min_clusters);
    if (dc.kvec < 1) dc.kvec = 1;
    if (dc.kvec > m) dc.kvec = m;
  }
  if (exact) {
    dc.converged = true;
    return dc;
  }
  bool ok = true;
  const double floor_abs = 1e-14 * scale;
  // values actually read by the eigengap loop
  int first = rq.descend ? 0 : 1, last = kw - 1;
  if (rq.fixed_count > 0) first = 0;
  if (rq.descend && rq.use_stop && rq.fixed_count == 0) {
    for (int i = 0; i < kw; ++i)
      if (w[i] < rq.stop_eigenvalue) { last = i; break; }
  }
  const bool aware = rq.decision_aware && rq.fixed_count == 0;
  const int kb = dc.n_clusters_raw;  // the maximum gap sits between w[kb - 1] and w[kb]
  for (int i = first; i <= last; ++i) {
    const bool decisive = !aware || i == kb - 1 || i == kb ||
                          (rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF && rq.descend && i == 0);
    const double rel = decisive ? rq.value_tol : std::max(rq.value_tol, 1e-3);
    const double tol = std::max(rel * std::fabs(w[i]), floor_abs);
    if (!(resid[i] <= tol)) {
      if (ok) { dc.fail_kind = 1; dc.fail_index = i; }
      ok = false;
    }
    dc.max_resid = std::max(dc.max_resid, resid[i]);
  }
  if (aware && ok) {
    // interval check: eigenvalue i lies within err(i) of w[i] (10 x residual: a safety
    // factor for the departure from normality)
    auto err = [&](int i) { return 10.0 * resid[i]; };
    const double eps = 1e-10;
    const double wmax = rq.descend ? w[0] : w[m - 1];
    auto gap_bounds = [&](int lo_i, int hi_i, double* lower, double* upper) {
      // Ratio: w[hi_i] / (w[lo_i] + eps); NormalizedDiff: (w[hi_i] - w[lo_i]) / wmax,
      // where hi_i is the numerator index
      const double a = w[hi_i], b = w[lo_i], ea = err(hi_i), eb = err(lo_i);
      if (rq.eigengap_type == SC_EIGENGAP_RATIO) {
        const double den_lo = b - eb + eps, den_hi = b + eb + eps;
        *upper = den_lo > 0.0 ? (a + ea) / den_lo : 1e300;
        *lower = den_hi > 0.0 ? (a - ea) / den_hi : -1e300;
      } else {
        *upper = (a - b + ea + eb) / wmax;
        *lower = (a - b - ea - eb) / wmax;
      }
    };
    double best_lo = 0.0, dummy;
    if (kb >= 1) {
      if (rq.descend) gap_bounds(kb, kb - 1, &best_lo, &dummy);
      else gap_bounds(kb - 1, kb, &best_lo, &dummy);
    }
    const int end = kw;
    if (rq.descend) {
      for (int i = 1; i < end && ok; ++i) {
        if (rq.use_stop) {
          if (std::fabs(w[i - 1] - rq.stop_eigenvalue) <= err(i - 1)) {
            ok = false; dc.fail_kind = 4; dc.fail_index = i - 1;
            break;
          }
          if (w[i - 1] < rq.stop_eigenvalue) break;
        }
        if (i == kb) continue;
        double lo, up;
        gap_bounds(i, i - 1, &lo, &up);
        if (!(up < best_lo) && !(kb == 0 && up <= 0.0)) {
          ok = false; dc.fail_kind = 4; dc.fail_index = i;
        }
      }
    } else {
      for (int i = 1; i < end - 1 && ok; ++i) {
        if (i + 1 == kb) continue;
        double lo, up;
        gap_bounds(i, i + 1, &lo, &up);
        if (!(up < best_lo) && !(kb == 0 && up <= 0.0)) {
          ok = false; dc.fail_kind = 4; dc.fail_index = i;
        }
      }
    }
  }
  if (!rq.descend && rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF && rq.fixed_count == 0) {
    // np.max(eigenvalues): the far end of the spectrum only normalises the gaps (it
    // cannot change n_clusters) and sits on the edge of a dense bulk where Krylov
    // methods converge like 1/degree^2: accept a 1e-4 residual bound there.
    const double tol = std::max(std::max(rq.value_tol, 1e-4) * std::fabs(w[m - 1]), floor_abs);
    if (!(resid[m - 1] <= tol)) {
      if (ok) { dc.fail_kind = 2; dc.fail_index = m - 1; }
      ok = false;
    }
  }
  for (int i = 0; i < dc.kvec; ++i) {
    if (!(resid[i] <= std::max(rq.vector_tol * scale, floor_abs))) {
      if (ok) { dc.fail_kind = 3; dc.fail_index = i; }
      ok = false;
    }
    dc.max_resid = std::max(dc.max_resid, resid[i]);
  }
  dc.converged = ok;
  return dc;
}

// One CholQR pass on W (n x 8) with the orthonormality-defect flag of its input armed
// (flags[10]); stores the result into Q[:, store_col ...] and Vs when store_col >= 0.
static int cholqr_pass(sc_handle h, int n, int store_col) {
  hipStream_t s = h->stream;
  double* W = ptr<double>(h->W);
  launch_proj_partial(s, W, kEigBlock, kEigBlock, W, n, ptr<double>(h->partial));
  launch_reduce_chol(s, ptr<double>(h->partial), proj_blocks(n), ptr<double>(h->Rinv), nullptr,
                     nullptr, ptr<int>(h->flags), ptr<int>(h->flags) + 10, 2);
  launch_apply_rinv(s, W, n, ptr<double>(h->Rinv), store_col >= 0 ? ptr<double>(h->Q) : nullptr,
                    kLdq, store_col >= 0 ? store_col : 0,
                    h->vs_scale ? h->vs_scale : ptr<double>(h->cvec), ptr<double>(h->Vs));
  return SC_OK;
}

static const char kNonFiniteMessage[] = "Array must not contain infs or NaNs";

// Orthonormalise W (n x 16) against Q[:, 0:m] and within itself.
//   record: accumulate the projection coefficients into T columns [col0, col0+16)
//   store_col: column of Q to receive the result (< 0: do not store)
static int orthonormalize(sc_handle h, int n, int m, bool record, int col0, int store_col,
                          bool save_gram) {
  hipStream_t s = h->stream;
  double* Q = ptr<double>(h->Q);
  double* W = ptr<double>(h->W);
  double* part = ptr<double>(h->partial);
  double* hsq = ptr<double>(h->hsq);
  SC_HIP(h, hipMemsetAsync(hsq, 0, 16 * sizeof(double), s));
  if (m > 0) {
    for (int pass = 0; pass < 2; ++pass) {
      launch_proj_partial(s, Q, kLdq, m, W, n, part);
      launch_reduce_H(s, part, proj_blocks(n), m, ptr<double>(h->Hbuf),
                      record ? ptr<double>(h->T) : nullptr, kLdq, col0, pass, hsq);
      launch_update_block(s, Q, kLdq, m, ptr<double>(h->Hbuf), W, n);
    }
  }
  // CholQR2
  launch_proj_partial(s, W, kEigBlock, kEigBlock, W, n, part);
  launch_reduce_chol(s, part, proj_blocks(n), ptr<double>(h->Rinv),
                     save_gram ? ptr<double>(h->G) : nullptr, hsq, ptr<int>(h->flags),
                     ptr<int>(h->flags) + 11, 1);
  launch_apply_rinv(s, W, n, ptr<double>(h->Rinv), nullptr, 0, 0, nullptr, nullptr);
  SC_TRY(cholqr_pass(h, n, store_col));
  return check_last(h, "orthonormalize launch");
}

static int read_flags(sc_handle h, int* mask) {
  SC_HIP(h, hipMemcpyAsync(h->h_flags, h->flags.p, 13 * sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  *mask = h->h_flags[0];
  if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
  if (h->h_flags[1] > 0 && getenv("SC_EIG_TRACE")) {
    fprintf(stderr, "[sc] jacobi sweeps=%d  %.1f us  %.0f MHz shader clock\n", h->h_flags[1],
            h->h_flags[2] * 0.01, h->h_flags[3] * 1024.0 / (h->h_flags[2] * 0.01));
    fprintf(stderr, "[sc]   thread-0 kcycles: param %d  barrier1 %d  update %d  barrier2 %d\n",
            h->h_flags[4], h->h_flags[5], h->h_flags[6], h->h_flags[7]);
    hipMemsetAsync(ptr<int>(h->flags) + 1, 0, 2 * sizeof(int), h->stream);
  }
  return SC_OK;
}

// Make sure the block in W is a full-rank orthonormal block; repairs dependent
// columns with random vectors (bounded retries).
static int finish_block(sc_handle h, int n, int m, int store_col, uint64_t* seed) {
  for (int attempt = 0; attempt < 4; ++attempt) {
    int mask = 0;
    SC_TRY(read_flags(h, &mask));
    // CholQR2 only orthonormalises blocks of condition < ~1e8; a numerically low-rank
    // operator produces worse ones: keep passing until the input Gram matrix was near I
    for (int extra = 0;
         extra < 3 && mask == 0 && (h->h_flags[10] != 0 || h->h_flags[11] != 0); ++extra) {
      // (the projection coefficients already recorded in T stay: this round only removes
      // rounding-level components)
      SC_TRY(orthonormalize(h, n, m, false, 0, store_col, false));
      SC_TRY(read_flags(h, &mask));
    }
    if (mask == 0) return SC_OK;
    launch_refill_deficient(h->stream, ptr<double>(h->W), n, ptr<int>(h->flags), ++(*seed));
    SC_TRY(orthonormalize(h, n, m, false, 0, store_col, false));
  }
  return fail(h, SC_ERR_NOT_CONVERGED, "could not build a full-rank Krylov block");
}

static void back_transform_cols(sc_handle h, int n, int cols) {
  launch_back_transform(h->stream, ptr<double>(h->E), round_up(n, 16), n, cols,
                        ptr<double>(h->tvec));
}

// S (n x n, ld) symmetric on the device; cvec/pvec/tvec already set.
static int sym_topk(sc_handle h, const double* S, int ld, int n, const EigRequest& rq,
                    sc_diag* diag, EigDecision* out_dc, std::vector<double>* out_w) {
  hipStream_t s = h->stream;
  SC_TRY(ensure_eig(h, n));
  double* theta_d = ptr<double>(h->theta);
  double* resid_d = ptr<double>(h->resid);
  const double* cvec = ptr<double>(h->cvec);
  const double* pvec = ptr<double>(h->pvec);
  EigDecision dc;
  int m = 0, passes = 0, cycles = 0;

  if (n <= kDenseMax) {
    // ---- direct dense path: every eigenpair, one Jacobi launch
    launch_jacobi(s, S, ld, n, 1, cvec, pvec, nullptr, theta_d, ptr<double>(h->Y), kLdq,
                  nullptr, ptr<double>(h->Yt), ptr<int>(h->flags));
    SC_TRY(check_last(h, "jacobi launch"));
    SC_HIP(h, hipMemcpyAsync(h->h_theta, theta_d, n * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                             hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
    for (int i = 0; i < n; ++i) h->h_theta[kLdq + i] = 0.0;
    dc = analyze(rq, h->h_theta, h->h_theta + kLdq, n, n, true);
    if (!dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
    m = n;
    const int cols = n;  // all eigenvectors, like np.linalg.eig
    launch_rowmajor_to_colmajor(s, ptr<double>(h->Y), kLdq, n, cols, ptr<double>(h->E),
                                round_up(n, 16));
    back_transform_cols(h, n, cols);
    h->n_vec = cols;
    if (diag) diag->eig_path = SC_EIG_PATH_DENSE_JACOBI;
    dc.kw = n;
  } else {
    if (rq.fixed_count == 0 && rq.max_clusters == 0 && !rq.descend)
      return fail(h, SC_ERR_UNSUPPORTED,
                  "max_clusters=None with a Laplacian needs every eigenvalue; only "
                  "supported for n <= 128 on the device path");
    uint64_t seed = 0x5eed5eedull;
    // ---- start block
    launch_random_block(s, ptr<double>(h->W), n, seed);
    SC_TRY(orthonormalize(h, n, 0, false, 0, 0, false));
    SC_TRY(finish_block(h, n, 0, 0, &seed));
    SC_HIP(h, hipMemsetAsync(h->T.p, 0, (size_t)kLdq * kLdq * sizeof(double), s));
    // basis cap: LDS Jacobi limit, and basis + next block must fit in R^n
    const int cap = std::min(kEigBasisCap, ((n - kEigBlock) / kEigBlock) * kEigBlock);
    const int first_check = std::min(3 * kEigBlock, cap);
    bool done = false;
    while (!done) {
      // block V_j lives in Q[:, m : m + 16]; Vs = c .* V_j
      launch_block_matvec(s, S, ld, n, cvec, pvec, ptr<double>(h->Q) + m, kLdq,
                          ptr<double>(h->Vs), ptr<double>(h->W));
      ++passes;
      m += kEigBlock;
      SC_TRY(orthonormalize(h, n, m, true, m - kEigBlock, m, true));
      // Rayleigh-Ritz is the expensive serial step: every block early on (where
      // convergence is expected), then sparser, then once per restart cycle.
      const bool check = cycles == 0 ? (m >= first_check && (m <= 4 * kEigBlock ||
                                                               m % (2 * kEigBlock) == 0 ||
                                                               m + kEigBlock > cap))
                                     : (m + kEigBlock > cap);
      if (check) {
        launch_jacobi(s, ptr<double>(h->T), kLdq, m, 0, nullptr, nullptr, ptr<double>(h->G),
                      theta_d, ptr<double>(h->Y), kLdq, resid_d, ptr<double>(h->Yt),
                      ptr<int>(h->flags));
        SC_TRY(check_last(h, "jacobi launch"));
        SC_HIP(h, hipMemcpyAsync(h->h_theta, theta_d, m * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
        SC_HIP(h, hipMemcpyAsync(h->h_theta + kLdq, resid_d, m * sizeof(double),
                                 hipMemcpyDeviceToHost, s));
      }
      SC_TRY(finish_block(h, n, m, m, &seed));  // syncs the stream
      if (check) {
        dc = analyze(rq, h->h_theta, h->h_theta + kLdq, m, n, false);
        if (getenv("SC_EIG_TRACE")) {
          int worst = 0;
          double wr = 0.0;
          for (int i = 0; i < std::min(m, dc.kw > 0 ? dc.kw : m); ++i) {
            const double r = h->h_theta[kLdq + i] / std::max(std::fabs(h->h_theta[i]), 1e-300);
            if (r > wr) { wr = r; worst = i; }
          }
          fprintf(stderr, "[sc] lanczos pass %d m=%d cycle %d: enough=%d conv=%d kw=%d kvec=%d "
                  "fail kind %d at %d (theta %.6g resid %.2e); far end theta=%.6g resid=%.2e\n",
                  passes, m, cycles, dc.enough, dc.converged, dc.kw, dc.kvec, dc.fail_kind,
                  dc.fail_index, dc.fail_index >= 0 ? h->h_theta[dc.fail_index] : 0.0,
                  dc.fail_index >= 0 ? h->h_theta[kLdq + dc.fail_index] : 0.0, h->h_theta[m - 1],
                  h->h_theta[kLdq + m - 1]);
          (void)wr; (void)worst;
        }
        if (dc.unsupported)
          return fail(h, SC_ERR_UNSUPPORTED,
                      "more than 64 eigenvalues are needed (max_clusters=None with a "
                      "slowly decaying spectrum); set max_clusters");
        if (dc.enough && dc.converged) {
          done = true;
          break;
        }
      }
      if (m + kEigBlock > cap) {
        // ---- thick restart: keep the leading Ritz vectors + the new block
        if (++cycles > rq.max_cycles)
          return fail(h, SC_ERR_NOT_CONVERGED, "block Lanczos did not converge");
        int want = dc.enough ? std::max(dc.kw, dc.kvec) : cap / 4;
        int keep = round_up(want + kEigBlock, kEigBlock);
        keep = std::max(kEigBlock, std::min(keep, cap - 2 * kEigBlock));
        if (!rq.descend && rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF &&
            rq.fixed_count == 0 && keep < m)
          // np.max(eigenvalues) is the far end of the spectrum: keep that Ritz pair too
          launch_swap_ritz(s, ptr<double>(h->Y), kLdq, m, theta_d, keep - 1, m - 1);
        launch_basis_times_Y(s, ptr<double>(h->Q), kLdq, m, ptr<double>(h->Y), kLdq, keep,
                             ptr<double>(h->Q2), kLdq, n, 0);
        launch_copy_block(s, ptr<double>(h->Q) + m, kLdq, ptr<double>(h->Q2) + keep, kLdq, n,
                          kEigBlock);
        std::swap(h->Q, h->Q2);
        launch_set_diag_T(s, ptr<double>(h->T), kLdq, kLdq, theta_d, keep);
        SC_TRY(check_last(h, "restart launch"));
        m = keep;
      }
    }
    const int cols = std::min(std::max(dc.kw, dc.kvec), kMaxVectors);
    launch_basis_times_Y(s, ptr<double>(h->Q), kLdq, m, ptr<double>(h->Y), kLdq, cols,
                         ptr<double>(h->E), round_up(n, 16), n, 1);
    back_transform_cols(h, n, cols);
    SC_TRY(check_last(h, "ritz vector launch"));
    h->n_vec = cols;
    if (diag) diag->eig_path = SC_EIG_PATH_BLOCK_LANCZOS;
  }
  if (out_w) {
    out_w->resize(dc.kw);
    for (int i = 0; i < dc.kw; ++i) (*out_w)[i] = rq.descend ? h->h_theta[i] : -h->h_theta[i];
  }
  if (diag) {
    diag->eig_matvec_passes = passes;
    diag->eig_block = kEigBlock;
    diag->eig_basis = m;
    diag->eig_cycles = cycles;
    diag->eig_max_residual = dc.max_resid;
  }
  *out_dc = dc;
  return SC_OK;
}

// ------------------------------------------------------------------------------
// general (non-symmetric) top-k eigensolver driver
// ------------------------------------------------------------------------------
// M (n x n, ld): the refined matrix, NOT diagonally similar to a symmetric one.
// Operator  Op x = p .* x + cl .* (M (cr .* x))  (= M, or minus the Laplacian), whose
// eigenvalues of largest real part are wanted; eigenvectors are those of the reference's
// matrix itself (no similarity transform).  n <= 64: the dense solver on the materialised
// matrix (every eigenpair, like np.linalg.eig).  Larger n: block Arnoldi with full
// re-orthogonalisation, explicit Rayleigh-Ritz H = Q^T Op Q (basis <= 64), explicit
// residuals ||Op v - theta v||, explicit restart from the wanted Ritz vectors (real and
// imaginary parts of complex pairs).
static int gen_topk(sc_handle h, const double* M, int ld, int n, int laplacian_type,
                    const EigRequest& rq, sc_diag* diag, EigDecision* out_dc,
                    std::vector<double>* out_w) {
  hipStream_t s = h->stream;
  SC_TRY(ensure_eig(h, n));
  SC_TRY(ensure_gen(h, n));
  double* theta_d = ptr<double>(h->theta);
  double* thetai_d = ptr<double>(h->thetai);
  double* resid_d = ptr<double>(h->resid);
  double* Yre = ptr<double>(h->Y);
  double* Yim = ptr<double>(h->Yt);
  double* Vre = ptr<double>(h->Vre);
  double* Vim = ptr<double>(h->Vim);
  int* info_d = ptr<int>(h->flags) + 8;
  const int ldv = round_up(n, 16);
  double* th = h->h_theta;             // [0, kLdq): Re theta, [kLdq, 2 kLdq): resid
  double* thi = h->h_theta + 2 * kLdq;  // Im theta
  EigDecision dc;
  int m = 0, passes = 0, cycles = 0;
  const bool is_lap = laplacian_type >= SC_LAPLACIAN_UNNORMALIZED;
  const bool far_end = !rq.descend && rq.eigengap_type == SC_EIGENGAP_NORMALIZED_DIFF &&
                       rq.fixed_count == 0;

  auto fetch_ritz = [&](int count) -> int {
    SC_HIP(h, hipMemcpyAsync(th, theta_d, count * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(thi, thetai_d, count * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 8, info_d, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipMemcpyAsync(h->h_flags + 12, ptr<int>(h->flags) + 12, sizeof(int),
                             hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    if (h->h_flags[12] != 0) return fail(h, SC_ERR_NON_FINITE, kNonFiniteMessage);
    if (h->h_flags[8] != 0)
      return fail(h, SC_ERR_NOT_CONVERGED, "QR iteration of the projected eigenproblem failed");
    return SC_OK;
  };

  if (n <= kGenMax) {
    // ---- dense: eigen-decomposition of the reference's own matrix
    const double* src = M;
    if (is_lap) {
      launch_laplacian(s, M, ptr<double>(h->genL), n, ld, laplacian_type, ptr<double>(h->deg));
      src = ptr<double>(h->genL);
    }
    launch_gen_eig(s, src, ld, n, is_lap ? -1.0 : 1.0, n, theta_d, thetai_d, Yre, Yim, kLdq,
                   info_d);
    SC_TRY(check_last(h, "dense general eigensolver launch"));
    SC_TRY(fetch_ritz(n));
    for (int i = 0; i < n; ++i) th[kLdq + i] = 0.0;
    dc = analyze(rq, th, th + kLdq, n, n, true);
    if (!dc.enough) return fail(h, SC_ERR_UNSUPPORTED, "eigen request cannot be satisfied");
    launch_gen_ritz(s, nullptr, 0, n, n, Yre, Yim, kLdq, n, Vre, Vim, ldv);
    launch_gen_phase(s, Vre, Vim, ldv, n, n, ptr<double>(h->E), ldv);
    SC_TRY(check_last(h, "eigenvector normalisation launch"));
    h->n_vec = n;
    m = n;
    dc.kw = n;
    if (diag) diag->eig_path = SC_EIG_PATH_DENSE_GENERAL;
  } else {
    if (rq.fixed_count == 0 && rq.max_clusters == 0 && !rq.descend)
      return fail(h, SC_ERR_UNSUPPORTED,
                  "max_clusters=None with a Laplacian needs every eigenvalue; only "
                  "supported for n <= 64 on the general eigen path");
    const double* cl = ptr<double>(h->cvec);
    const double* cr = ptr<double>(h->crvec);
    const double* pv = ptr<double>(h->pvec);
    double* Q = ptr<double>(h->Q);
    double* OpQ = ptr<double>(h->Q2);
    double* W = ptr<double>(h->W);
    h->vs_scale = cr;
    struct Restore {
      sc_handle h;
      ~Restore() { h->vs_scale = nullptr; }
    } restore{h};
    const int cap = std::min(kGenMax, ((n - kEigBlock) / kEigBlock) * kEigBlock);
    const int first_check = std::min(3 * kEigBlock, cap);
    uint64_t seed = 0x9e3779b97f4a7c15ull;
    std::vector<std::vector<int>> start_blocks;  // restart: codes 2*col+part, -1 = noise
    size_t next_start = 0;
    const int kMaxCheck = 32;  // Ritz pairs whose residual is evaluated
    while (true) {
      // ---- next block into W
      if (next_start < start_blocks.size()) {
        SC_HIP(h, hipMemcpyAsync(h->gsrc.p, start_blocks[next_start].data(),
                                 kEigBlock * sizeof(int), hipMemcpyHostToDevice, s));
        launch_gen_gather(s, Vre, Vim, ldv, n, ptr<int>(h->gsrc), ++seed, W);
        SC_HIP(h, hipStreamSynchronize(s));  // the code vector is host memory
        ++next_start;
      } else if (m == 0) {
        launch_random_block(s, W, n, seed);
      } else {
        launch_copy_block(s, OpQ + (m - kEigBlock), kLdq, W, kEigBlock, n, kEigBlock);
      }
      SC_TRY(orthonormalize(h, n, m, false, 0, m, false));
      SC_TRY(finish_block(h, n, m, m, &seed));
      launch_block_matvec(s, M, ld, n, cl, pv, Q + m, kLdq, ptr<double>(h->Vs), W);
      launch_copy_block(s, W, kEigBlock, OpQ + m, kLdq, n, kEigBlock);
      ++passes;
      m += kEigBlock;
      // Rayleigh-Ritz (a serial ~m^3 solve in one wavefront) is the expensive step: every
      // block early in the first cycle, where convergence is expected, then every other
      // block, and only with a full basis once restarts have begun
      const bool check = next_start >= start_blocks.size() && m >= first_check &&
                         (cycles == 0 ? (m <= 4 * kEigBlock || m % (2 * kEigBlock) == 0 ||
                                         m + kEigBlock > cap)
                                      : (m + kEigBlock > cap));
      if (check) {
        // H = Q^T (Op Q), one 8-column block at a time
        for (int jb = 0; jb < m; jb += kEigBlock) {
          launch_copy_block(s, OpQ + jb, kLdq, W, kEigBlock, n, kEigBlock);
          launch_proj_partial(s, Q, kLdq, m, W, n, ptr<double>(h->partial));
          launch_reduce_H(s, ptr<double>(h->partial), proj_blocks(n), m, ptr<double>(h->Hbuf),
                          nullptr, 0, 0, 0, ptr<double>(h->hsq));
          launch_copy_block(s, ptr<double>(h->Hbuf), kEigBlock, ptr<double>(h->T) + jb, kLdq,
                            m, kEigBlock);
        }
        launch_gen_eig(s, ptr<double>(h->T), kLdq, m, 1.0, m, theta_d, thetai_d, Yre, Yim, kLdq,
                       info_d);
        const int c1 = std::min(m, kMaxCheck);
        launch_gen_residual(s, Q, OpQ, kLdq, m, n, Yre, Yim, kLdq, theta_d, thetai_d, c1,
                            ptr<double>(h->gpart), resid_d);
        if (far_end && m - 1 >= c1)
          launch_gen_residual(s, Q, OpQ, kLdq, m, n, Yre + (m - 1), Yim + (m - 1), kLdq,
                              theta_d + (m - 1), thetai_d + (m - 1), 1, ptr<double>(h->gpart),
                              resid_d + (m - 1));
        SC_TRY(check_last(h, "Rayleigh-Ritz launch"));
        for (int i = 0; i < m; ++i) th[kLdq + i] = 1e300;  // not evaluated = not converged
        SC_HIP(h, hipMemcpyAsync(th + kLdq, resid_d, c1 * sizeof(double), hipMemcpyDeviceToHost,
                                 s));
        if (far_end && m - 1 >= c1)
          SC_HIP(h, hipMemcpyAsync(th + kLdq + m - 1, resid_d + (m - 1), sizeof(double),
                                   hipMemcpyDeviceToHost, s));
        SC_TRY(fetch_ritz(m));
        dc = analyze(rq, th, th + kLdq, m, n, false);
        if (getenv("SC_EIG_TRACE"))
          fprintf(stderr, "[sc] arnoldi pass %d m=%d cycle %d sweeps %d: enough=%d conv=%d kw=%d "
                  "kvec=%d fail kind %d at %d (resid %.2e)\n", passes, m, cycles, h->h_flags[9],
                  dc.enough, dc.converged, dc.kw, dc.kvec, dc.fail_kind, dc.fail_index,
                  dc.fail_index >= 0 ? th[kLdq + dc.fail_index] : 0.0);
        if (getenv("SC_EIG_TRACE") && atoi(getenv("SC_EIG_TRACE")) > 1) {
          for (int i = 0; i < std::min(m, 12); ++i)
            fprintf(stderr, "[sc]    ritz %2d  re %.12g  im %.3e  resid %.3e\n", i, th[i], thi[i],
                    th[kLdq + i]);
        }
        if (dc.unsupported || (dc.enough && std::max(dc.kw, dc.kvec) > kMaxCheck))
          return fail(h, SC_ERR_UNSUPPORTED,
                      "the general eigen path reports at most 32 eigenpairs for n > 64; set "
                      "max_clusters <= 31");
        if (dc.enough && dc.converged) break;
      }
      if (m + kEigBlock > cap) {
        // ---- explicit restart from the wanted Ritz vectors
        if (++cycles > rq.max_cycles)
          return fail(h, SC_ERR_NOT_CONVERGED, "block Arnoldi did not converge");
        // (thick restart: the new basis is [wanted Ritz vectors | residual block], after
        // which the Arnoldi recurrence continues from the residual block)
        constexpr int kStash = 48;  // Vre columns [48, 56) hold the residual block
        launch_copy_block(s, OpQ + (m - kEigBlock), kLdq, W, kEigBlock, n, kEigBlock);
        SC_TRY(orthonormalize(h, n, m, false, 0, -1, false));
        SC_TRY(finish_block(h, n, m, -1, &seed));
        launch_rowmajor_to_colmajor(s, W, kEigBlock, n, kEigBlock, Vre + (size_t)kStash * ldv,
                                    ldv);
        const int want = dc.enough ? std::max(dc.kw, dc.kvec) : cap / 4;
        const int avail = std::min(m, 40);  // Ritz vectors materialised: columns [0, avail)
        launch_gen_ritz(s, Q, kLdq, m, n, Yre, Yim, kLdq, avail, Vre, Vim, ldv);
        int vcols = avail;
        const bool far_kept = far_end && m - 1 >= avail;
        if (far_kept) {  // Ritz vector m-1 -> column `avail`
          launch_gen_ritz(s, Q, kLdq, m, n, Yre + (m - 1), Yim + (m - 1), kLdq, 1,
                          Vre + (size_t)avail * ldv, Vim + (size_t)avail * ldv, ldv);
          vcols = avail + 1;
        }
        launch_gen_phase(s, Vre, Vim, ldv, n, vcols, nullptr, 0);
        SC_TRY(check_last(h, "restart launch"));
        const double scale = std::max(std::fabs(th[0]), std::fabs(th[m - 1]));
        auto is_complex = [&](int i) { return std::fabs(thi[i]) > 1e-12 * std::max(scale, 1e-300); };
        std::vector<int> codes;
        if (far_kept) {
          codes.push_back(2 * avail);
          if (is_complex(m - 1)) codes.push_back(2 * avail + 1);
        }
        // Kept vectors must fill whole blocks: a random pad column r would break the
        // relation Op [kept] in span(kept, residual block) -- (I - Q Q^T) Op r is not in the
        // basis -- and the Ritz pairs then stall at the size of their component along r.
        // So the kept set is extended, never padded.
        const int max_cols = (std::max(1, cap / kEigBlock - 3)) * kEigBlock;
        const int target = std::min(round_up(want + kEigBlock / 2, kEigBlock), max_cols);
        for (int i = 0; i < avail; ++i) {
          if ((int)codes.size() >= target && codes.size() % kEigBlock == 0) break;
          if (!is_complex(i)) {
            codes.push_back(2 * i);
            continue;
          }
          bool partner_kept = false;  // its conjugate, earlier in the list
          for (int j = 0; j < i; ++j)
            if (is_complex(j) && std::fabs(th[j] - th[i]) <= 1e-9 * scale &&
                std::fabs(thi[j] + thi[i]) <= 1e-9 * scale)
              partner_kept = true;
          if (partner_kept) continue;
          codes.push_back(2 * i);
          codes.push_back(2 * i + 1);
        }
        if ((int)codes.size() > max_cols) codes.resize(max_cols);
        while (codes.size() % kEigBlock) codes.push_back(-1);  // last resort (tiny bases)
        for (int j = 0; j < kEigBlock; ++j) codes.push_back(2 * (kStash + j));
        start_blocks.clear();
        for (size_t b = 0; b * kEigBlock < codes.size(); ++b)
          start_blocks.emplace_back(codes.begin() + b * kEigBlock,
                                    codes.begin() + (b + 1) * kEigBlock);
        next_start = 0;
        m = 0;
      }
    }
    const int cols = std::min(std::max(dc.kw, dc.kvec), kMaxCheck);
    launch_gen_ritz(s, Q, kLdq, m, n, Yre, Yim, kLdq, cols, Vre, Vim, ldv);
    launch_gen_phase(s, Vre, Vim, ldv, n, cols, ptr<double>(h->E), ldv);
    SC_TRY(check_last(h, "ritz vector launch"));
    h->n_vec = cols;
    if (diag) diag->eig_path = SC_EIG_PATH_BLOCK_ARNOLDI;
  }
  if (out_w) {
    out_w->resize(dc.kw);
    for (int i = 0; i < dc.kw; ++i) (*out_w)[i] = rq.descend ? th[i] : -th[i];
  }
  if (diag) {
    diag->eig_matvec_passes = passes;
    diag->eig_block = kEigBlock;
    diag->eig_basis = m;
    diag->eig_cycles = cycles;
    diag->eig_max_residual = dc.max_resid;
  }
  *out_dc = dc;
  return SC_OK;
}

// ------------------------------------------------------------------------------
// _compute_eigenvectors_ncluster
// ------------------------------------------------------------------------------
static void ev_rec(sc_handle h, int* slot) {
  if (h->nev < 48) {
    hipEventRecord(h->ev[h->nev], h->stream);
    *slot = h->nev++;
  } else {
    *slot = -1;
  }
}
static float ev_ms(sc_handle h, int a, int b) {
  if (a < 0 || b < 0) return 0.f;
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->ev[a], h->ev[b]);
  return ms;
}

static int eig_ncluster_impl(sc_handle h, const sc_config* cfg, sc_diag* diag) {
  const int n = h->n, ld = h->ldn;
  hipStream_t s = h->stream;
  const double* cur = ptr<double>(h->A0);
  double* bufs[2] = {ptr<double>(h->B1), ptr<double>(h->B2)};
  int which = 0;
  bool symmetric = h->affinity_symmetric;  // cosine affinity: always
  const bool constrain_after = constraint_active(h, cfg, false);
  bool folded_rownorm = false;
  int e_begin, e_after_refine;
  float diffuse_ms_events[SC_MAX_OPS][2];
  int n_diffuse = 0;
  ev_rec(h, &e_begin);
  // Fusions (identical arithmetic, fewer passes over the n x n matrix):
  //   CropDiagonal + GaussianBlur      -> crop value vector + blur with diagonal override
  //   GaussianBlur -> RowWiseThreshold -> row maxima come out of the blur epilogue
  //   RowWiseThreshold(RowMax) + Symmetrize -> one tile-pair kernel
  const bool blur_fast = (cfg->blur_radius == 4 || cfg->blur_radius == 8) && n >= 128;
  const double* pending_diag = nullptr;
  bool have_partials = false;
  bool have_row_stats = false;
  for (int i = 0; i < cfg->n_ops; ++i) {
    const int op = cfg->ops[i];
    const int next = i + 1 < cfg->n_ops ? cfg->ops[i + 1] : 0;
    const int next2 = i + 2 < cfg->n_ops ? cfg->ops[i + 2] : 0;
    const bool thr_sym_fusable = true;  // RowMax or Percentile, with or without diagonal
    const bool partials_usable = cfg->threshold_type == SC_THRESHOLD_ROW_MAX &&
                                 !cfg->preserve_diagonal;
    if (op == SC_OP_ROW_WISE_NORMALIZE && symmetric && i == cfg->n_ops - 1 &&
        !constrain_after) {
      folded_rownorm = true;  // W = diag(1/rowmax) S is never materialised
      continue;
    }
    if (op == SC_OP_CROP_DIAGONAL && next == SC_OP_GAUSSIAN_BLUR && blur_fast) {
      if (cur == ptr<double>(h->A0) && h->have_cropval) {
        pending_diag = ptr<double>(h->cropval);
      } else {
        launch_crop_value(s, cur, n, ld, ptr<double>(h->dvec));
        pending_diag = ptr<double>(h->dvec);
      }
      have_partials = false;
      continue;  // symmetry unchanged; the blur applies the new diagonal on load
    }
    double* out = bufs[which];
    which ^= 1;
    int e0 = -1, e1 = -1;
    if (op == SC_OP_DIFFUSE) ev_rec(h, &e0);
    if (op == SC_OP_GAUSSIAN_BLUR && blur_fast) {
      const bool want = next == SC_OP_ROW_WISE_THRESHOLD && next2 == SC_OP_SYMMETRIZE &&
                        partials_usable;
      SC_HIP(h, hipMemcpyAsync(h->blurw.p, cfg->blur_weights,
                               (2 * cfg->blur_radius + 1) * sizeof(double),
                               hipMemcpyHostToDevice, s));
      have_partials = launch_gaussian_blur_fused(s, cur, out, n, ld, cfg->blur_radius,
                                                 ptr<double>(h->blurw), pending_diag,
                                                 want ? ptr<double>(h->rmpart) : nullptr);
      pending_diag = nullptr;
      SC_TRY(check_last(h, "blur launch"));
    } else if (op == SC_OP_ROW_WISE_THRESHOLD && next == SC_OP_SYMMETRIZE && thr_sym_fusable) {
      if (cfg->threshold_type == SC_THRESHOLD_PERCENTILE)
        launch_cut_percentile(s, cur, n, ld, cfg->p_percentile, ptr<double>(h->cut),
                              cfg->preserve_diagonal);
      else if (have_partials && partials_usable)
        launch_cut_from_partials(s, ptr<double>(h->rmpart), n,
                                 blur_tile_columns(n, cfg->blur_radius),
                                 cfg->p_percentile, ptr<double>(h->cut));
      else
        launch_cut_from_rows(s, cur, n, ld, cfg->p_percentile, ptr<double>(h->cut),
                             cfg->preserve_diagonal);
      launch_threshold_symmetrize(s, cur, out, n, ld, ptr<double>(h->cut),
                                  cfg->soft_multiplier, cfg->binarize, cfg->symmetrize_type,
                                  cfg->preserve_diagonal);
      SC_TRY(check_last(h, "threshold+symmetrize launch"));
      have_partials = false;
      cur = out;
      symmetric = true;
      ++i;  // Symmetrize consumed
      continue;
    } else if (op == SC_OP_DIFFUSE) {
      // when Diffuse is the last materialised matrix its row max / row sum (for the
      // RowWiseNormalize fold and the Laplacian scaling) come out of the GEMM epilogue
      const bool last = i == cfg->n_ops - 1 ||
                        (i == cfg->n_ops - 2 && next == SC_OP_ROW_WISE_NORMALIZE);
      GemmRowStats rs{1, ptr<double>(h->statp), ptr<double>(h->statp) + (size_t)n * gemm_tile_dim(n),
                      ptr<double>(h->rowmax), ptr<double>(h->rowsum)};
      SC_TRY(ensure_tilemap(h, n));
      launch_gemm_nt(s, cur, ld, cur, ld, out, ld, n, n, n, kEpiNone, true,
                     ptr<double>(h->splitk), ptr<int2>(h->tilemap), last ? &rs : nullptr);
      SC_TRY(check_last(h, "diffuse launch"));
      have_row_stats = last;
      have_partials = false;
    } else {
      SC_TRY(run_refine_op(h, op, cfg, cur, out, n, ld));
      have_partials = false;
    }
    if (op == SC_OP_DIFFUSE) {
      ev_rec(h, &e1);
      diffuse_ms_events[n_diffuse][0] = (float)e0;
      diffuse_ms_events[n_diffuse][1] = (float)e1;
      ++n_diffuse;
    }
    cur = out;
    switch (op) {
      case SC_OP_CROP_DIAGONAL:
      case SC_OP_GAUSSIAN_BLUR:
        break;  // symmetry preserved (blur: up to rounding)
      case SC_OP_ROW_WISE_THRESHOLD:
      case SC_OP_ROW_WISE_NORMALIZE:
        symmetric = false;
        break;
      case SC_OP_SYMMETRIZE:
      case SC_OP_DIFFUSE:
        symmetric = true;
        break;
    }
  }
  if (constrain_after) {  // spectral_clusterer.py:137-142
    if (h->qn != n)
      return fail(h, SC_ERR_INVALID,
                  "affinity and constraint matrix must have the same shape");
    double* out = bufs[which];
    which ^= 1;
    SC_TRY(adjust_affinity(h, cfg, cur, symmetric, out, n, ld));
    cur = out;
    symmetric = symmetric && h->constraint_symmetric;
    have_row_stats = false;
  }
  ev_rec(h, &e_after_refine);
  // ---- scaling vectors (RowWiseNormalize fold + Laplacian)
  if (!symmetric) {
    // general matrix (e.g. RowWiseThreshold without a later Symmetrize / Diffuse):
    // Op x = p .* x + cl .* (M (cr .* x)), no similarity transform
    SC_TRY(ensure_gen(h, n));
    launch_row_stats(s, cur, n, ld, ptr<double>(h->rowmax), ptr<double>(h->deg));
    launch_scaling_general(s, ptr<double>(h->deg), n, cfg->laplacian_type,
                           ptr<double>(h->cvec), ptr<double>(h->crvec), ptr<double>(h->pvec));
  } else {
    if (!have_row_stats)
      launch_row_stats(s, cur, n, ld, ptr<double>(h->rowmax), ptr<double>(h->rowsum));
    launch_scaling_vectors(s, ptr<double>(h->rowmax), ptr<double>(h->rowsum), n,
                           cfg->laplacian_type, folded_rownorm ? 1 : 0, ptr<double>(h->cvec),
                           ptr<double>(h->pvec), ptr<double>(h->tvec));
  }
  // a NaN / inf anywhere in the refined matrix (zero embedding rows, an all-zero refined row
  // under RowWiseNormalize, ...) reaches its row sums, hence c / p: np.linalg.eig raises on
  // such input; the flag is read with the solver's first host sync
  SC_TRY(ensure_eig(h, n));
  if (h->affinity_from_embeddings)  // a zero embedding row: its NaNs may have been dropped
    SC_HIP(h, hipMemcpyAsync(ptr<int>(h->flags) + 12, ptr<int>(h->symflag) + 1, sizeof(int),
                             hipMemcpyDeviceToDevice, s));
  else
    SC_HIP(h, hipMemsetAsync(ptr<int>(h->flags) + 12, 0, sizeof(int), s));
  launch_check_finite(s, ptr<double>(h->cvec), ptr<double>(h->pvec), n, ptr<int>(h->flags) + 12);
  SC_TRY(check_last(h, "scaling launch"));
  if (getenv("SC_EIG_TRACE") && atoi(getenv("SC_EIG_TRACE")) > 2 && symmetric) {
    std::vector<double> rm(n), rs(n);
    hipMemcpyAsync(rm.data(), h->rowmax.p, n * sizeof(double), hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(rs.data(), h->rowsum.p, n * sizeof(double), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    int imin = 0, imax = 0;
    for (int i = 0; i < n; ++i) {
      if (rm[i] < rm[imin]) imin = i;
      if (rm[i] > rm[imax]) imax = i;
    }
    fprintf(stderr, "[sc] scaling: fused stats %d; rowmax min %.6g at %d, max %.6g at %d; "
            "rowsum[%d] %.6g\n", (int)have_row_stats, rm[imin], imin, rm[imax], imax, imin,
            rs[imin]);
  }
  int e_after_scaling;
  ev_rec(h, &e_after_scaling);
  // ---- eigen + eigengap
  EigRequest rq;
  rq.descend = (cfg->laplacian_type == SC_LAPLACIAN_NONE ||
                cfg->laplacian_type == SC_LAPLACIAN_AFFINITY);
  rq.max_clusters = cfg->max_clusters;
  rq.min_clusters = cfg->min_clusters;
  rq.stop_eigenvalue = cfg->stop_eigenvalue;
  rq.eigengap_type = cfg->eigengap_type;
  rq.use_stop = rq.descend;  // spectral_clusterer.py:163-167: not passed when ascending
  rq.value_tol = cfg->eig_value_tol > 0 ? cfg->eig_value_tol : 1e-6;
  rq.vector_tol = cfg->eig_vector_tol > 0 ? cfg->eig_vector_tol : 1e-10;
  rq.max_cycles = cfg->eig_max_cycles > 0 ? cfg->eig_max_cycles : 40;
  rq.fixed_count = 0;
  EigDecision dc;
  std::vector<double> w;
  if (symmetric) {
    SC_TRY(sym_topk(h, cur, ld, n, rq, diag, &dc, &w));
  } else {
    rq.decision_aware = 1;
    SC_TRY(gen_topk(h, cur, ld, n, cfg->laplacian_type, rq, diag, &dc, &w));
  }
  int e_after_eig;
  ev_rec(h, &e_after_eig);
  SC_HIP(h, hipStreamSynchronize(s));
  if (diag) {
    diag->n = n;
    diag->n_clusters_raw = dc.n_clusters_raw;
    diag->max_delta = dc.max_delta;
    diag->eig_descending = rq.descend;
    diag->n_eigenvalues = std::min((int)w.size(), SC_MAX_EIG);
    for (int i = 0; i < diag->n_eigenvalues; ++i) diag->eigenvalues[i] = w[i];
    diag->symmetry_state = !symmetric ? 3 : (folded_rownorm ? 2 : 1);
    float dms = 0.f;
    for (int i = 0; i < n_diffuse; ++i)
      dms += ev_ms(h, (int)diffuse_ms_events[i][0], (int)diffuse_ms_events[i][1]);
    diag->stage_ms[SC_STAGE_DIFFUSE] = dms;
    diag->stage_ms[SC_STAGE_REFINE] = ev_ms(h, e_begin, e_after_refine) - dms;
    diag->stage_ms[SC_STAGE_SCALING] = ev_ms(h, e_after_refine, e_after_scaling);
    diag->stage_ms[SC_STAGE_EIG] = ev_ms(h, e_after_scaling, e_after_eig);
  }
  return SC_OK;
}

extern "C" int sc_eig_ncluster(sc_handle h, const sc_config* cfg, sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  SC_HIP(h, hipSetDevice(h->device));
  h->nev = 0;
  if (diag) memset(diag, 0, sizeof(*diag));
  return eig_ncluster_impl(h, cfg, diag);
}

extern "C" int sc_num_eigenvectors(sc_handle h) { return h ? h->n_vec : 0; }

extern "C" int sc_get_eigenvectors(sc_handle h, double* out, int n, int ncols) {
  if (!h || !out) return SC_ERR_INVALID;
  if (n != h->n || ncols <= 0 || ncols > h->n_vec)
    return fail(h, SC_ERR_INVALID, "eigenvector request out of range");
  SC_HIP(h, hipSetDevice(h->device));
  launch_colmajor_to_rowmajor(h->stream, ptr<double>(h->E), round_up(n, 16), n, ncols,
                              ptr<double>(h->Eio), ncols);
  return d2h_matrix(h, ptr<double>(h->Eio), ncols, n, ncols, out);
}

// ------------------------------------------------------------------------------
// k-means tail
// ------------------------------------------------------------------------------
static int kmeans_on_device(sc_handle h, const double* E, int lde, int n, int k, int max_iter,
                            int64_t* labels, double* centroids_out, int* iterations,
                            int metric = kKmeansCosine) {
  if (metric != kKmeansCosine && metric != kKmeansEuclidean && metric != kKmeansSqeuclidean &&
      metric != kKmeansCityblock && metric != kKmeansChebyshev)
    return fail(h, SC_ERR_UNSUPPORTED,
                "custom_dist on the device: cosine, euclidean, sqeuclidean, cityblock, "
                "chebyshev");
  if (max_iter <= 0)
    return fail(h, SC_ERR_INVALID, "Number of iterations should be a positive number");
  if (n < k) return fail(h, SC_ERR_INVALID, "n_samples should be >= n_clusters");
  if (k < 1 || k > kMaxVectors)
    return fail(h, SC_ERR_UNSUPPORTED, "n_clusters must be in [1, 64] on the device path");
  SC_TRY(ensure_kmeans(h, n));
  // RandomState(0): first centre via choice(n, p=uniform) = cdf.searchsorted(u, 'right')
  Mt19937 rng(0);
  const double u = rng.next_double();
  const int first = sc_uniform_choice(n, u);
  const int trials = 2 + (int)std::log((double)k);
  std::vector<double> rnd((size_t)std::max(1, (k - 1) * trials));
  for (size_t i = 0; i < rnd.size(); ++i) rnd[i] = rng.next_double();
  if (rnd.size() > 1024) return fail(h, SC_ERR_UNSUPPORTED, "too many k-means++ trials");
  SC_HIP(h, hipMemcpyAsync(h->krnd.p, rnd.data(), rnd.size() * sizeof(double),
                           hipMemcpyHostToDevice, h->stream));
  KmeansWorkspace ws;
  ws.Xc = ptr<double>(h->kXc);
  ws.xsq = ptr<double>(h->kxsq);
  ws.closest = ptr<double>(h->kclosest);
  ws.cand = ptr<double>(h->kcand);
  ws.enorm = ptr<double>(h->kenorm);
  ws.rnd = ptr<double>(h->krnd);
  ws.centroids = ptr<double>(h->kcent);
  ws.labels32 = ptr<int>(h->klab32);
  ws.labels64 = ptr<long long>(h->klab64);
  ws.info = ptr<int>(h->kinfo);
  SC_HIP(h, hipMemsetAsync(h->kinfo.p, 0, 8 * sizeof(int), h->stream));
  launch_kmeans(h->stream, E, lde, n, k, max_iter, first, trials, ws, metric);
  SC_TRY(check_last(h, "kmeans launch"));
  SC_HIP(h, hipMemcpyAsync(labels, h->klab64.p, (size_t)n * sizeof(int64_t),
                           hipMemcpyDeviceToHost, h->stream));
  int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SC_HIP(h, hipMemcpyAsync(info, h->kinfo.p, 6 * sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
  if (centroids_out)
    SC_HIP(h, hipMemcpyAsync(centroids_out, h->kcent.p, (size_t)k * k * sizeof(double),
                             hipMemcpyDeviceToHost, h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  if (iterations) *iterations = info[0];
  if (getenv("SC_KMEANS_TRACE"))
    fprintf(stderr, "[sc] kmeans n=%d k=%d iters=%d  us: centre %.1f  kmeans++ %.1f  lloyd %.1f"
            "  cosine-loop %.1f\n", n, k, info[0], info[1] * 0.01, (info[2] - info[1]) * 0.01,
            (info[3] - info[2]) * 0.01, (info[4] - info[3]) * 0.01);
  return SC_OK;
}

extern "C" int sc_cluster(sc_handle h, const sc_config* cfg, int n_clusters, int64_t* labels,
                          sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  if (!cfg || !labels) return fail(h, SC_ERR_INVALID, "NULL argument");
  if (h->n_vec <= 0) return fail(h, SC_ERR_INVALID, "no eigenvectors resident");
  if (n_clusters < 1 || n_clusters > h->n_vec)
    return fail(h, SC_ERR_INVALID, "n_clusters exceeds the resident eigenvectors");
  SC_HIP(h, hipSetDevice(h->device));
  const int n = h->n;
  int e0, e1;
  ev_rec(h, &e0);
  const double* E = ptr<double>(h->E);
  const int lde = round_up(n, 16);
  if (cfg->row_wise_renorm) {
    SC_TRY(ensure_kmeans(h, n));
    SC_HIP(h, hipMemcpyAsync(h->Ek.p, h->E.p, (size_t)lde * n_clusters * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    launch_row_renorm(h->stream, ptr<double>(h->Ek), lde, n, n_clusters);
    E = ptr<double>(h->Ek);
  }
  int iters = 0;
  SC_TRY(kmeans_on_device(h, E, lde, n, n_clusters, cfg->max_iter, labels, nullptr, &iters,
                          cfg->kmeans_metric));
  ev_rec(h, &e1);
  SC_HIP(h, hipStreamSynchronize(h->stream));
  if (diag) {
    diag->n_clusters = n_clusters;
    diag->kmeans_iterations = iters;
    diag->stage_ms[SC_STAGE_KMEANS] = ev_ms(h, e0, e1);
  }
  return SC_OK;
}

// ------------------------------------------------------------------------------
// whole path
// ------------------------------------------------------------------------------
extern "C" int sc_run_resident(sc_handle h, const sc_config* cfg, int64_t* labels,
                               sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!labels) return fail(h, SC_ERR_INVALID, "labels is NULL");
  if (!h->have_x) return fail(h, SC_ERR_INVALID, "no embeddings resident");
  SC_HIP(h, hipSetDevice(h->device));
  sc_diag local;
  sc_diag* dg = diag ? diag : &local;
  memset(dg, 0, sizeof(*dg));
  h->nev = 0;
  int e0, e1, e2;
  ev_rec(h, &e0);
  SC_TRY(sc_compute_affinity(h));
  if (constraint_active(h, cfg, true)) SC_TRY(sc_apply_constraint(h, cfg));  // :259-264
  ev_rec(h, &e1);
  SC_TRY(eig_ncluster_impl(h, cfg, dg));
  int k = dg->n_clusters_raw;
  if (cfg->min_clusters > 0 && k < cfg->min_clusters) k = cfg->min_clusters;  // :295-296
  SC_TRY(sc_cluster(h, cfg, k, labels, dg));
  ev_rec(h, &e2);
  SC_HIP(h, hipStreamSynchronize(h->stream));
  dg->stage_ms[SC_STAGE_AFFINITY] = ev_ms(h, e0, e1);
  dg->stage_ms[SC_STAGE_TOTAL] = ev_ms(h, e0, e2);
  return SC_OK;
}

extern "C" int sc_predict(sc_handle h, const double* x, int n, int d, const sc_config* cfg,
                          int64_t* labels, sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  SC_TRY(sc_set_embeddings(h, x, n, d));
  return sc_run_resident(h, cfg, labels, diag);
}

extern "C" int sc_predict_batch(sc_handle h, const double* const* xs, const int* ns, int d,
                                int count, const sc_config* cfg, int64_t* const* labels,
                                sc_diag* diags) {
  if (!h) return SC_ERR_INVALID;
  if (!xs || !ns || !labels || count < 0) return fail(h, SC_ERR_INVALID, "NULL argument");
  int nmax = 0;
  for (int i = 0; i < count; ++i) nmax = std::max(nmax, ns[i]);
  if (nmax > 0) SC_TRY(sc_reserve(h, nmax, d));  // one arena sized for the largest member
  for (int i = 0; i < count; ++i)
    SC_TRY(sc_predict(h, xs[i], ns[i], d, cfg, labels[i], diags ? diags + i : nullptr));
  return SC_OK;
}

// ------------------------------------------------------------------------------
// N4: size reduction -- agglomerative clustering + centroids
// ------------------------------------------------------------------------------
namespace {
// CPython heapq (Lib/heapq.py) on ints: sklearn's _hc_cut enumerates the heap ARRAY, so
// the exact sift order defines the label numbering.
void heap_siftdown(std::vector<long long>& heap, size_t startpos, size_t pos) {
  const long long newitem = heap[pos];
  while (pos > startpos) {
    const size_t parentpos = (pos - 1) >> 1;
    const long long parent = heap[parentpos];
    if (newitem < parent) {
      heap[pos] = parent;
      pos = parentpos;
      continue;
    }
    break;
  }
  heap[pos] = newitem;
}
void heap_siftup(std::vector<long long>& heap, size_t pos) {
  const size_t endpos = heap.size(), startpos = pos;
  const long long newitem = heap[pos];
  size_t childpos = 2 * pos + 1;
  while (childpos < endpos) {
    const size_t rightpos = childpos + 1;
    if (rightpos < endpos && !(heap[childpos] < heap[rightpos])) childpos = rightpos;
    heap[pos] = heap[childpos];
    pos = childpos;
    childpos = 2 * pos + 1;
  }
  heap[pos] = newitem;
  heap_siftdown(heap, startpos, pos);
}
void heap_push(std::vector<long long>& heap, long long item) {
  heap.push_back(item);
  heap_siftdown(heap, 0, heap.size() - 1);
}
void heap_pushpop(std::vector<long long>& heap, long long item) {
  if (!heap.empty() && heap[0] < item) {
    std::swap(item, heap[0]);
    heap_siftup(heap, 0);
  }
}
}  // namespace

// sklearn.cluster.AgglomerativeClustering(metric="cosine", linkage=complete|average,
// n_clusters=... | distance_threshold=...).fit_predict(X), label numbering included.
extern "C" int sc_ahc(sc_handle h, const double* x, int n, int d, int linkage, int n_clusters,
                      double distance_threshold, int64_t* labels, int* n_clusters_out) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !labels || d <= 0) return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  if (n < 2)
    return fail(h, SC_ERR_INVALID,
                "Found array with " + std::to_string(std::max(n, 0)) +
                    " sample(s) while a minimum of 2 is required by AgglomerativeClustering.");
  if (linkage != SC_LINKAGE_COMPLETE && linkage != SC_LINKAGE_AVERAGE)
    return fail(h, SC_ERR_INVALID, "linkage must be complete or average");
  if (n_clusters < 0 || n_clusters > n)
    return fail(h, SC_ERR_INVALID, "Cannot extract more clusters than samples");
  SC_HIP(h, hipSetDevice(h->device));
  // cosine distances: the affinity stage's normalise + symmetric GEMM, then 1 - clip(c)
  SC_TRY(sc_set_embeddings(h, x, n, d));
  SC_TRY(ensure_tilemap(h, n));
  hipStream_t s = h->stream;
  const int ld = h->ldn;
  launch_normalize_rows(s, ptr<double>(h->X), h->ldx, n, d, ptr<double>(h->Xn));
  launch_gemm_nt(s, ptr<double>(h->Xn), h->ldx, ptr<double>(h->Xn), h->ldx, ptr<double>(h->B1),
                 ld, n, n, d, kEpiNone, true, ptr<double>(h->splitk), ptr<int2>(h->tilemap));
  launch_cosine_distance(s, ptr<double>(h->B1), n, ld);
  SC_TRY(grow(h, h->ahc_size, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_chain, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_Z, (size_t)n * 4 * sizeof(double)));
  launch_ahc_nn_chain(s, ptr<double>(h->B1), ld, n, linkage, ptr<int>(h->ahc_size),
                      ptr<int>(h->ahc_chain), ptr<double>(h->ahc_Z));
  SC_TRY(check_last(h, "agglomerative clustering launch"));
  std::vector<double> Z((size_t)(n - 1) * 4);
  SC_HIP(h, hipMemcpyAsync(Z.data(), h->ahc_Z.p, Z.size() * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  h->have_affinity = h->have_cropval = false;
  // ---- scipy: stable sort by height, union-find relabelling (hierarchy.pyx `label`)
  std::vector<int> order(n - 1);
  for (int i = 0; i < n - 1; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return Z[(size_t)a * 4 + 2] < Z[(size_t)b * 4 + 2]; });
  std::vector<int> parent(2 * (size_t)n - 1);
  for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
  auto find = [&](int v) {
    int r = v;
    while (parent[r] != r) r = parent[r];
    while (parent[v] != r) {
      const int next = parent[v];
      parent[v] = r;
      v = next;
    }
    return r;
  };
  std::vector<std::array<long long, 2>> children(n - 1);
  std::vector<double> heights(n - 1);
  int next_label = n;
  for (int i = 0; i < n - 1; ++i) {
    const int m = order[i];
    const int xr = find((int)Z[(size_t)m * 4]), yr = find((int)Z[(size_t)m * 4 + 1]);
    children[i] = {std::min(xr, yr), std::max(xr, yr)};
    heights[i] = Z[(size_t)m * 4 + 2];
    parent[xr] = next_label;
    parent[yr] = next_label;
    ++next_label;
  }
  // ---- sklearn: number of clusters, then _hc_cut
  int k = n_clusters;
  if (k == 0) {  // distance_threshold mode
    k = 1;
    for (int i = 0; i < n - 1; ++i) k += heights[i] >= distance_threshold;
  }
  if (n_clusters_out) *n_clusters_out = k;
  std::vector<long long> nodes;
  nodes.push_back(-(std::max(children[n - 2][0], children[n - 2][1]) + 1));
  for (int it = 0; it < k - 1; ++it) {
    const auto c = children[(size_t)(-nodes[0] - n)];
    heap_push(nodes, -c[0]);
    heap_pushpop(nodes, -c[1]);
  }
  std::vector<long long> stack;
  for (size_t i = 0; i < nodes.size(); ++i) {
    stack.assign(1, -nodes[i]);
    while (!stack.empty()) {
      const long long v = stack.back();
      stack.pop_back();
      if (v < n) {
        labels[v] = (int64_t)i;
      } else {
        stack.push_back(children[(size_t)(v - n)][0]);
        stack.push_back(children[(size_t)(v - n)][1]);
      }
    }
  }
  return SC_OK;
}

// utils.get_cluster_centroids (reference utils.py:159-176): (k, d) means, k = max(labels)+1
extern "C" int sc_cluster_centroids(sc_handle h, const double* x, int n, int d,
                                    const int64_t* labels, int k, double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !labels || !out || n <= 0 || d <= 0 || k <= 0)
    return fail(h, SC_ERR_INVALID, "embeddings must be (n, d), labels (n,)");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->ahc_lab, (size_t)n * sizeof(int)));
  SC_TRY(grow(h, h->ahc_cent, (size_t)(n + k) * d * sizeof(double)));
  std::vector<int> lab32(n);
  for (int i = 0; i < n; ++i) lab32[i] = (int)labels[i];
  double* xd = ptr<double>(h->ahc_cent);
  double* cd_ = xd + (size_t)n * d;
  SC_HIP(h, hipMemcpyAsync(xd, x, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, s));
  SC_HIP(h, hipMemcpyAsync(h->ahc_lab.p, lab32.data(), (size_t)n * sizeof(int),
                           hipMemcpyHostToDevice, s));
  launch_cluster_centroids(s, xd, d, n, d, ptr<int>(h->ahc_lab), k, cd_);
  SC_TRY(check_last(h, "centroid launch"));
  SC_HIP(h, hipMemcpyAsync(out, cd_, (size_t)k * d * sizeof(double), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  return SC_OK;
}

// ------------------------------------------------------------------------------
// N4: fallback decisions (reference fallback_clusterer.py, naive_clusterer.py)
// ------------------------------------------------------------------------------
// out = {affinity.min(), np.diag(affinity, k=1).min(), mean, np.std(affinity)} of the
// resident affinity (single-cluster conditions AllAffinity / NeighborAffinity / AffinityStd)
extern "C" int sc_affinity_stats(sc_handle h, double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!out) return fail(h, SC_ERR_INVALID, "out is NULL");
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  SC_HIP(h, hipSetDevice(h->device));
  const int n = h->n;
  SC_TRY(grow(h, h->fb_part, (size_t)n * 8 * sizeof(double)));
  SC_TRY(grow(h, h->fb_small, 32 * sizeof(double)));
  launch_affinity_stats(h->stream, ptr<double>(h->A0), n, h->ldn, ptr<double>(h->fb_part),
                        ptr<double>(h->fb_small));
  SC_TRY(check_last(h, "affinity statistics launch"));
  SC_HIP(h, hipMemcpyAsync(out, h->fb_small.p, 4 * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  return SC_OK;
}

// BIC of a 1- and a 2-component Gaussian mixture fitted to affinity[i][j], j >= i + offset
// (fallback_clusterer.py:154-173).  sklearn's GaussianMixture defaults: full covariance,
// reg_covar 1e-6, tol 1e-3 on the mean log-likelihood, max_iter 100, k-means start.  The
// reference's k-means start is randomly seeded; here it is the deterministic 1-D 2-means
// from (min, max), which is the fixed point those seeds reach on separable data.
extern "C" int sc_affinity_gmm_bic(sc_handle h, int diagonal_offset, double* bic1,
                                   double* bic2) {
  if (!h) return SC_ERR_INVALID;
  if (!bic1 || !bic2) return fail(h, SC_ERR_INVALID, "NULL output");
  if (!h->have_affinity) return fail(h, SC_ERR_INVALID, "no affinity resident");
  const int n = h->n;
  if (diagonal_offset < 0 || diagonal_offset >= n - 1)
    return fail(h, SC_ERR_INVALID,
                "single_cluster_affinity_diagonal_offset must be significantly smaller than "
                "affinity matrix dimension");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->fb_part, (size_t)n * 8 * sizeof(double)));
  SC_TRY(grow(h, h->fb_small, 32 * sizeof(double)));
  double* params_d = ptr<double>(h->fb_small);
  double* sums_d = params_d + 8;
  const double* a = ptr<double>(h->A0);
  const int ld = h->ldn;
  const double m = (double)(n - diagonal_offset);
  const double count = m * (m + 1.0) / 2.0;
  const double reg = 1e-6, tiny = 10.0 * 2.220446049250313e-16;
  double sums[8];
  auto pass = [&](int components, int mode, const double* params) -> int {
    SC_HIP(h, hipMemcpyAsync(params_d, params, 6 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_gmm_pass(s, a, n, ld, diagonal_offset, components, mode, params_d,
                    ptr<double>(h->fb_part), sums_d);
    SC_HIP(h, hipMemcpyAsync(sums, sums_d, 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    SC_HIP(h, hipStreamSynchronize(s));
    return SC_OK;
  };
  // M-step of sklearn's _estimate_gaussian_parameters from the pass sums
  auto m_step = [&](int components, double* params) {
    double wsum = 0.0;
    for (int c = 0; c < components; ++c) {
      const double nk = sums[3 * c] + tiny;
      const double mu = sums[3 * c + 1] / nk;
      const double var = (sums[3 * c + 2] - 2.0 * mu * sums[3 * c + 1] + mu * mu * sums[3 * c]) / nk;
      params[3 * c] = nk / count;
      params[3 * c + 1] = mu;
      params[3 * c + 2] = var + reg;
      wsum += params[3 * c];
    }
    for (int c = 0; c < components; ++c) params[3 * c] /= wsum;
  };
  auto fit = [&](int components, double* bic) -> int {
    double params[6] = {1.0, 0.0, 1.0, 0.0, 0.0, 1.0};
    if (components == 1) {
      SC_TRY(pass(1, 1, params));  // r0 = 1 everywhere: plain moments
      m_step(1, params);
    } else {
      // 2-means start from the extremes, Lloyd steps until the inertia stops moving
      launch_gmm_range(s, a, n, ld, diagonal_offset, ptr<double>(h->fb_part), sums_d);
      double range[2];
      SC_HIP(h, hipMemcpyAsync(range, sums_d, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
      SC_HIP(h, hipStreamSynchronize(s));
      params[1] = range[0];
      params[4] = range[1];
      double prev_inertia = -1.0;
      for (int it = 0; it < 300; ++it) {
        SC_TRY(pass(2, 0, params));
        const double inertia = sums[6];
        if (sums[0] > 0.0) params[1] = sums[1] / sums[0];
        if (sums[3] > 0.0) params[4] = sums[4] / sums[3];
        if (inertia == prev_inertia) break;
        prev_inertia = inertia;
      }
      SC_TRY(pass(2, 0, params));  // responsibilities = the final hard labels
      m_step(2, params);
    }
    double prev = -__builtin_huge_val();
    for (int it = 0; it < 100; ++it) {
      SC_TRY(pass(components, 1, params));  // E-step under params (+ sums of the M-step)
      const double lower_bound = sums[6] / count;
      m_step(components, params);
      if (std::fabs(lower_bound - prev) < 1e-3) break;
      prev = lower_bound;
    }
    SC_TRY(pass(components, 1, params));
    const double n_params = components == 1 ? 2.0 : 5.0;
    *bic = -2.0 * sums[6] + n_params * std::log(count);
    return SC_OK;
  };
  SC_TRY(fit(1, bic1));
  SC_TRY(fit(2, bic2));
  return SC_OK;
}

// NaiveClusterer.predict (naive_clusterer.py:57-105) continuing from the given state:
// centroids (capacity x d, the first *n_centroids rows valid), counts, labels out.
extern "C" int sc_naive_cluster(sc_handle h, const double* x, int n, int d, double threshold,
                                double adaptation_threshold, double* centroids, int32_t* counts,
                                int32_t* n_centroids, int capacity, int64_t* labels) {
  if (!h) return SC_ERR_INVALID;
  if (!x || !centroids || !counts || !n_centroids || !labels || n <= 0 || d <= 0)
    return fail(h, SC_ERR_INVALID, "embeddings must be (n, d)");
  if (*n_centroids < 0 || *n_centroids + n > capacity)
    return fail(h, SC_ERR_INVALID, "centroid capacity must cover n_centroids + n");
  SC_HIP(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SC_TRY(grow(h, h->fb_x, (size_t)n * d * sizeof(double)));
  SC_TRY(grow(h, h->fb_cent, (size_t)capacity * d * sizeof(double)));
  SC_TRY(grow(h, h->fb_int, ((size_t)capacity + n + 4) * sizeof(int)));
  int* counts_d = ptr<int>(h->fb_int);
  int* k_d = counts_d + capacity;
  int* labels_d = k_d + 4;
  const int k0 = *n_centroids;
  SC_HIP(h, hipMemcpyAsync(h->fb_x.p, x, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, s));
  if (k0 > 0) {
    SC_HIP(h, hipMemcpyAsync(h->fb_cent.p, centroids, (size_t)k0 * d * sizeof(double),
                             hipMemcpyHostToDevice, s));
    SC_HIP(h, hipMemcpyAsync(counts_d, counts, (size_t)k0 * sizeof(int), hipMemcpyHostToDevice, s));
  }
  SC_HIP(h, hipMemcpyAsync(k_d, n_centroids, sizeof(int), hipMemcpyHostToDevice, s));
  launch_naive_cluster(s, ptr<double>(h->fb_x), n, d, threshold, adaptation_threshold,
                       ptr<double>(h->fb_cent), counts_d, k_d, labels_d);
  SC_TRY(check_last(h, "naive clusterer launch"));
  std::vector<int> lab(n);
  SC_HIP(h, hipMemcpyAsync(lab.data(), labels_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(n_centroids, k_d, sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  const int k1 = *n_centroids;
  SC_HIP(h, hipMemcpyAsync(centroids, h->fb_cent.p, (size_t)k1 * d * sizeof(double),
                           hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipMemcpyAsync(counts, counts_d, (size_t)k1 * sizeof(int), hipMemcpyDeviceToHost, s));
  SC_HIP(h, hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) labels[i] = lab[i];
  return SC_OK;
}

// ------------------------------------------------------------------------------
// single stages
// ------------------------------------------------------------------------------
extern "C" int sc_stage_affinity(sc_handle h, const double* x, int n, int d, double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!out) return fail(h, SC_ERR_INVALID, "out is NULL");
  SC_TRY(sc_set_embeddings(h, x, n, d));
  SC_TRY(sc_compute_affinity(h));
  return d2h_matrix(h, ptr<double>(h->A0), h->ldn, n, n, out);
}

extern "C" int sc_stage_refine(sc_handle h, int op, const sc_config* cfg, const double* in,
                               int n, double* out) {
  if (!h) return SC_ERR_INVALID;
  SC_TRY(validate_config(h, cfg));
  if (!in || !out || n <= 0) return fail(h, SC_ERR_INVALID, "affinity must be (n, n)");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  const int ld = matrix_ld(n);
  h->have_affinity = h->have_cropval = false;
  h->have_x = false;
  h->n_vec = 0;
  SC_TRY(h2d_matrix(h, in, n, n, ptr<double>(h->B1), ld));
  SC_TRY(run_refine_op(h, op, cfg, ptr<double>(h->B1), ptr<double>(h->B2), n, ld));
  return d2h_matrix(h, ptr<double>(h->B2), ld, n, n, out);
}

extern "C" int sc_stage_laplacian(sc_handle h, int laplacian_type, const double* in, int n,
                                  double* out) {
  if (!h) return SC_ERR_INVALID;
  if (!in || !out || n <= 0) return fail(h, SC_ERR_INVALID, "affinity must be (n, n)");
  if (laplacian_type < SC_LAPLACIAN_AFFINITY || laplacian_type > SC_LAPLACIAN_GRAPH_CUT)
    return fail(h, SC_ERR_INVALID, "laplacian_type must be a LaplacianType");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  const int ld = matrix_ld(n);
  h->have_affinity = h->have_cropval = false;
  h->have_x = false;
  h->n_vec = 0;
  SC_TRY(h2d_matrix(h, in, n, n, ptr<double>(h->B1), ld));
  if (laplacian_type == SC_LAPLACIAN_AFFINITY)
    return d2h_matrix(h, ptr<double>(h->B1), ld, n, n, out);
  launch_laplacian(h->stream, ptr<double>(h->B1), ptr<double>(h->B2), n, ld, laplacian_type,
                   ptr<double>(h->deg));
  SC_TRY(check_last(h, "laplacian launch"));
  return d2h_matrix(h, ptr<double>(h->B2), ld, n, n, out);
}

__global__ void k_negate(const double* in, double* out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = -in[e];
}

__global__ void k_fill(double* p, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

extern "C" int sc_stage_sym_eig(sc_handle h, const double* m, int n, int count, int descend,
                                double* values, double* vectors, sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  if (!m || n <= 0 || count <= 0 || count > n || !values)
    return fail(h, SC_ERR_INVALID, "bad eigen request");
  if (n > kDenseMax && count > kMaxVectors)
    return fail(h, SC_ERR_UNSUPPORTED, "at most 64 eigenpairs for n > 128");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  const int ld = matrix_ld(n);
  h->n = n;
  h->ldn = ld;
  h->have_affinity = h->have_cropval = false;
  h->have_x = false;
  SC_TRY(h2d_matrix(h, m, n, n, ptr<double>(h->B1), ld));
  // Op = +M (descend) or -M (ascend): c = 1, p = 0, t = 1; the sign is folded by
  // running on sigma * M through c = 1 and a negated copy when ascending.
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->cvec), n, 1.0);
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->pvec), n, 0.0);
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->tvec), n, 1.0);
  const double* S = ptr<double>(h->B1);
  if (!descend) {  // smallest eigenpairs of M = largest of -M
    hipLaunchKernelGGL(k_negate, dim3(2048), dim3(256), 0, h->stream, ptr<double>(h->B1),
                       ptr<double>(h->B2), (size_t)n * ld);
    S = ptr<double>(h->B2);
  }
  EigRequest rq;
  rq.descend = descend ? 1 : 0;
  rq.max_clusters = 0;
  rq.min_clusters = 0;
  rq.stop_eigenvalue = 0.0;
  rq.eigengap_type = SC_EIGENGAP_RATIO;
  rq.use_stop = 0;
  rq.value_tol = 1e-10;
  rq.vector_tol = 1e-11;
  rq.max_cycles = 60;
  rq.fixed_count = count;
  EigDecision dc;
  std::vector<double> w;
  sc_diag local;
  sc_diag* dg = diag ? diag : &local;
  memset(dg, 0, sizeof(*dg));
  h->nev = 0;
  SC_TRY(ensure_eig(h, n));
  SC_HIP(h, hipMemsetAsync(ptr<int>(h->flags) + 12, 0, sizeof(int), h->stream));
  SC_TRY(sym_topk(h, S, ld, n, rq, dg, &dc, &w));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < count; ++i) values[i] = w[i];
  if (vectors) {
    launch_colmajor_to_rowmajor(h->stream, ptr<double>(h->E), round_up(n, 16), n, count,
                                ptr<double>(h->Eio), count);
    SC_TRY(d2h_matrix(h, ptr<double>(h->Eio), count, n, count, vectors));
  }
  return SC_OK;
}

extern "C" int sc_stage_eig(sc_handle h, const double* m, int n, int count, int descend,
                            double* values, double* vectors, sc_diag* diag) {
  if (!h) return SC_ERR_INVALID;
  if (!m || n <= 0 || count <= 0 || count > n || !values)
    return fail(h, SC_ERR_INVALID, "bad eigen request");
  if (n > kGenMax && count > 32)
    return fail(h, SC_ERR_UNSUPPORTED, "at most 32 eigenpairs for n > 64 on the general path");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_matrices(h, n, 0));
  SC_TRY(ensure_gen(h, n));
  const int ld = matrix_ld(n);
  h->n = n;
  h->ldn = ld;
  h->have_affinity = h->have_cropval = false;
  h->have_x = false;
  SC_TRY(h2d_matrix(h, m, n, n, ptr<double>(h->B1), ld));
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->cvec), n, 1.0);
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->crvec), n, 1.0);
  hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, h->stream, ptr<double>(h->pvec), n, 0.0);
  const double* S = ptr<double>(h->B1);
  if (!descend) {  // smallest real parts of M = largest of -M
    hipLaunchKernelGGL(k_negate, dim3(2048), dim3(256), 0, h->stream, ptr<double>(h->B1),
                       ptr<double>(h->B2), (size_t)n * ld);
    S = ptr<double>(h->B2);
  }
  EigRequest rq;
  rq.descend = descend ? 1 : 0;
  rq.max_clusters = 0;
  rq.min_clusters = 0;
  rq.stop_eigenvalue = 0.0;
  rq.eigengap_type = SC_EIGENGAP_RATIO;
  rq.use_stop = 0;
  rq.value_tol = 1e-10;
  rq.vector_tol = 1e-11;
  rq.max_cycles = 100;
  rq.fixed_count = count;
  EigDecision dc;
  std::vector<double> w;
  sc_diag local;
  sc_diag* dg = diag ? diag : &local;
  memset(dg, 0, sizeof(*dg));
  h->nev = 0;
  SC_TRY(ensure_eig(h, n));
  SC_HIP(h, hipMemsetAsync(ptr<int>(h->flags) + 12, 0, sizeof(int), h->stream));
  SC_TRY(gen_topk(h, S, ld, n, SC_LAPLACIAN_NONE, rq, dg, &dc, &w));
  SC_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < count; ++i) values[i] = w[i];
  if (vectors) {
    launch_colmajor_to_rowmajor(h->stream, ptr<double>(h->E), round_up(n, 16), n, count,
                                ptr<double>(h->Eio), count);
    SC_TRY(d2h_matrix(h, ptr<double>(h->Eio), count, n, count, vectors));
  }
  return SC_OK;
}

extern "C" int sc_stage_kmeans_metric(sc_handle h, const double* e, int n, int k, int max_iter,
                                      int metric, int64_t* labels, double* centroids_out,
                                      int* iterations) {
  if (!h) return SC_ERR_INVALID;
  if (!e || !labels || n <= 0 || k <= 0) return fail(h, SC_ERR_INVALID, "bad k-means input");
  if (k > kMaxVectors)
    return fail(h, SC_ERR_UNSUPPORTED, "n_clusters must be <= 64 on the device path");
  SC_HIP(h, hipSetDevice(h->device));
  SC_TRY(ensure_kmeans(h, n));
  SC_HIP(h, hipMemcpyAsync(h->Eio.p, e, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
  launch_to_colmajor(h->stream, ptr<double>(h->Eio), n, k, ptr<double>(h->Ek),
                     round_up(n, 16));
  return kmeans_on_device(h, ptr<double>(h->Ek), round_up(n, 16), n, k, max_iter, labels,
                          centroids_out, iterations, metric);
}

extern "C" int sc_stage_kmeans(sc_handle h, const double* e, int n, int k, int max_iter,
                               int64_t* labels, double* centroids_out, int* iterations) {
  return sc_stage_kmeans_metric(h, e, n, k, max_iter, SC_KMEANS_COSINE, labels, centroids_out,
                                iterations);
}
