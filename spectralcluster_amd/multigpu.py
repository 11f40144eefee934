"""Partitioning of independent predict() calls over the GPUs of one node.

The hot path has no exchange step inside a predict(): a
problem of <= 537 MB per matrix fits one MI355X and its stages are sequentially
dependent.  What shards are the *units* around it:

  * batched utterances  -> `predict_batch_sharded`  (LPT by an n^3 cost model)
  * AutoTune p sweeps   -> `autotune_sharded`       (p-grid round-robin)

One process per GPU.  Communication goes through a `Comm`: on the GPU box that is
`RcclComm` -- RCCL over xGMI behind the C ABI (`sc_comm_*` in
include/spectralcluster_amd.h), no PyTorch.  The collectives are the bare minimum:
broadcast of inputs from rank 0, all-gather of results (labels / two scalars per p),
a max-reduce for timing.  Every function here takes the `Comm` and the per-unit
compute as arguments, so the partitioning logic is testable on CPU with any
byte-level transport (tests/ plugs in a gloo adapter for world size 2).
"""

from __future__ import annotations

import ctypes
import os
import socket
import struct
import time
import typing

import numpy as np


def cost_model(n: int) -> float:
  """Cost of one utterance of n samples inside a grouped batch (predict_batch(group=16), the
  execution the partition schedules), in microseconds on one MI355X.  Calibrated on measured
  grouped times at d=256 (tests/probes/cost_model_fit.py, profiles/r30_cost_fit.txt -- round 6's
  tree: 55 us at n=300, 79 at 650, 122 at 1200, 168 at 1535, 143 at 1536, 197 at 2000, 386 at
  3000; non-negative least squares on the relative error): a fixed per-utterance share of the
  group's launch chains and the O(n^2) passes.  From n = 1536 on members take the matrix-free
  Diffuse, whose digit product now runs over a skip list: what it costs depends on how the
  utterance's speakers fall into tiles, not on n alone -- the upper branch fits its record to 8 %
  (the branches below to 1.5 %), where round 5's, with every tile computed, fitted to 1 %.
  (Round 3's single cubic, calibrated on the explicit route, was 2x too high everywhere and
  had no kink: the 8 shares of config 5 came out 10-14 % apart.  Round 2's n^3 + 64 n^2 put a
  factor 900 between n=300 and n=3000 where the measured factor is 7.)"""
  n = float(n)
  if n < 512.0:
    return 50.0 + 6.5e-5 * n * n
  if n < 1536.0:
    return 65.38 + 2.665e-5 * n * n + 1.06e-8 * n * n * n
  return 66.5 + 3.529e-5 * n * n


def lpt_assignment(sizes: typing.Sequence[int], world: int) -> typing.List[typing.List[int]]:
  """Longest-processing-time-first: sort by cost descending, give each item to
  the least-loaded rank.  Returns, per rank, the item indices it owns (in the
  order it will run them).  Deterministic: ties go to the lowest rank."""
  order = sorted(range(len(sizes)), key=lambda i: (-cost_model(sizes[i]), i))
  load = [0.0] * world
  owned = [[] for _ in range(world)]
  for i in order:
    r = min(range(world), key=lambda q: (load[q], q))
    owned[r].append(i)
    load[r] += cost_model(sizes[i])
  return owned


# --------------------------------------------------------------------------------
# communicators
# --------------------------------------------------------------------------------
class Comm:
  """Byte-level collectives between `size` ranks (this process is `rank`)."""
  rank = 0
  size = 1

  def broadcast_bytes(self, data: typing.Optional[bytes], nbytes: int, root: int = 0) -> bytes:
    """`data` (nbytes long) on `root`; returns root's bytes on every rank."""
    raise NotImplementedError

  def allgather_bytes(self, data: bytes) -> typing.List[bytes]:
    """Every rank contributes the same number of bytes; returns them in rank order."""
    raise NotImplementedError

  def allreduce_max(self, value: float) -> float:
    raise NotImplementedError

  def barrier(self) -> None:
    self.allreduce_max(0.0)

  def close(self) -> None:
    pass


class LocalComm(Comm):
  """World of one: every collective is the identity."""

  def broadcast_bytes(self, data, nbytes, root=0):
    return bytes(data)

  def allgather_bytes(self, data):
    return [bytes(data)]

  def allreduce_max(self, value):
    return float(value)


_ID_MAGIC = b"SCCOMM1\0"


def _id_file() -> str:
  """Where rank 0 leaves the rendezvous blob for the other ranks of this launch.  All ranks of
  one `torch.distributed.run` launch share TORCHELASTIC_RUN_ID (when exported), a MASTER_PORT
  and a parent process, which makes the name unique per launch on the node; it lives in a
  per-user 0700 directory.  Launchers whose ranks have no common parent (some srun set-ups)
  must export SC_COMM_ID_FILE."""
  explicit = os.environ.get("SC_COMM_ID_FILE")
  if explicit:
    return explicit
  base = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sc_comm_%d" % os.getuid())
  try:
    os.mkdir(base, 0o700)
  except FileExistsError:
    pass
  st = os.lstat(base)
  import stat as _stat
  if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
    raise PermissionError("%s is not a private directory of this user" % base)
  run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
  run_id = "".join(ch for ch in run_id if ch.isalnum())[:32]
  tag = "%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), run_id or "x", os.getppid())
  return os.path.join(base, "%s.id" % tag)


def _collective_timeout() -> typing.Optional[float]:
  """Timeout of an established TCP collective.  None (the default) blocks: one rank may
  compute minutes longer than another between two collectives (an unbalanced batch share, a
  large-n AutoTune level), and a dead peer is noticed through its closed socket anyway.
  SC_COMM_TIMEOUT_S overrides."""
  v = os.environ.get("SC_COMM_TIMEOUT_S")
  return float(v) if v else None


class SocketComm(Comm):
  """The same byte collectives over TCP on one node, rank 0 as the hub: the fallback of
  `RcclComm.from_env` when RCCL cannot be brought up, and the channel the ranks agree on
  that through.  The replicas of this package exchange a few hundred bytes per job (timing
  reductions, labels of a sharded batch, AutoTune scalars): nothing here is on a data path."""

  def __init__(self, rank: int, size: int, conns, server=None):
    self.rank, self.size = int(rank), int(size)
    self._conns = conns    # rank 0: {rank: socket} of the others; else: {0: socket}
    self._server = server
    self.note = ""

  @staticmethod
  def _send(sock, blob: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(blob)) + blob)

  @staticmethod
  def _recv(sock) -> bytes:
    def exactly(k):
      parts = []
      while k:
        chunk = sock.recv(min(k, 1 << 20))
        if not chunk:
          raise ConnectionError("peer closed the connection")
        parts.append(chunk)
        k -= len(chunk)
      return b"".join(parts)
    (count,) = struct.unpack("<Q", exactly(8))
    return exactly(count)

  @classmethod
  def from_env(cls, rank: int, size: int, timeout_s: float = 120.0) -> "SocketComm":
    """Rank 0 listens on an ephemeral port of MASTER_ADDR (127.0.0.1 by default) and publishes
    the port through the launch's id file; the others connect and say who they are."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if rank == 0:
      server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
      server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
      server.bind((addr, 0))
      server.listen(size)
      server.settimeout(timeout_s)
      port = server.getsockname()[1]
      RcclComm.exchange_id(0, lambda: struct.pack("<I", port) + bytes(124), timeout_s)
      conns = {}
      while len(conns) < size - 1:
        peer, _ = server.accept()
        peer.settimeout(timeout_s)  # the handshake only ...
        peer.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        (who,) = struct.unpack("<I", cls._recv(peer))
        peer.settimeout(_collective_timeout())  # ... collectives wait for the slowest rank
        conns[who] = peer
      try:
        os.unlink(_id_file())
      except OSError:
        pass
      return cls(rank, size, conns, server)
    blob = RcclComm.exchange_id(rank, lambda: b"", timeout_s)
    (port,) = struct.unpack("<I", blob[:4])
    deadline = time.monotonic() + timeout_s
    while True:
      try:
        sock = socket.create_connection((addr, port), timeout=timeout_s)
        break
      except OSError:
        if time.monotonic() > deadline:
          raise
        time.sleep(0.05)
    sock.settimeout(timeout_s)
    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    cls._send(sock, struct.pack("<I", rank))
    sock.settimeout(_collective_timeout())
    return cls(rank, size, {0: sock})

  def _exchange(self, blob: bytes) -> typing.List[bytes]:
    """Every rank contributes a blob (any length); every rank gets all of them, rank order."""
    if self.rank == 0:
      blobs = [bytes(blob)] + [self._recv(self._conns[r]) for r in range(1, self.size)]
      packed = b"".join(struct.pack("<Q", len(b)) + b for b in blobs)
      for r in range(1, self.size):
        self._send(self._conns[r], packed)
      return blobs
    self._send(self._conns[0], bytes(blob))
    packed = self._recv(self._conns[0])
    out, pos = [], 0
    for _ in range(self.size):
      (count,) = struct.unpack_from("<Q", packed, pos)
      out.append(packed[pos + 8:pos + 8 + count])
      pos += 8 + count
    return out

  def broadcast_bytes(self, data, nbytes, root=0):
    return self._exchange(bytes(data) if self.rank == root else b"")[root]

  def allgather_bytes(self, data):
    return self._exchange(bytes(data))

  def allreduce_max(self, value):
    return max(struct.unpack("<d", b)[0] for b in self._exchange(struct.pack("<d", float(value))))

  def close(self):
    for sock in self._conns.values():
      try:
        sock.close()
      except OSError:
        pass
    self._conns = {}
    if self._server is not None:
      self._server.close()
      self._server = None


class RcclComm(Comm):
  """RCCL communicator behind the C ABI (`sc_comm_*`), one rank per GPU."""

  def __init__(self, handle, rank: int, size: int, unique_id: bytes):
    from spectralcluster_amd import _lib
    self._lib = handle.lib
    self._handle = handle  # keeps the device handle (stream) alive
    self.rank, self.size = int(rank), int(size)
    c = ctypes.c_void_p()
    rc = self._lib.sc_comm_init_rank(handle.raw, self.size, self.rank, unique_id,
                                     ctypes.byref(c))
    if rc != _lib.SC_OK:
      raise _lib.DeviceLibraryError("sc_comm_init_rank failed (%d): %s"
                                    % (rc, handle.last_error()))
    self._c = c

  @staticmethod
  def new_unique_id() -> bytes:
    from spectralcluster_amd import _lib
    buf = ctypes.create_string_buffer(128)
    rc = _lib.load().sc_comm_unique_id(buf)
    if rc != _lib.SC_OK:
      raise _lib.DeviceLibraryError("sc_comm_unique_id failed (%d): is librccl present?" % rc)
    return buf.raw

  @staticmethod
  def exchange_id(rank: int, make_id: typing.Callable[[], bytes],
                  timeout_s: float = 120.0) -> bytes:
    """Out-of-band hand-over of the 128-byte RCCL unique id on one node: rank 0 creates it
    and leaves it in `_id_file()` (written to a temporary name and renamed: readers never
    see a partial id); the other ranks poll for the file."""
    path = _id_file()
    if rank == 0:
      uid = make_id()
      # magic | creation time | payload, written under a private temporary name (O_EXCL and
      # O_NOFOLLOW: nothing pre-planted is followed or reused) and renamed into place
      blob = _ID_MAGIC + struct.pack("<d", time.time()) + uid
      tmp = "%s.%d.tmp" % (path, os.getpid())
      try:
        os.unlink(tmp)
      except OSError:
        pass
      fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
      with os.fdopen(fd, "wb") as f:
        f.write(blob)
      os.replace(tmp, path)
      return uid
    deadline = time.monotonic() + timeout_s
    while True:
      try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
        with os.fdopen(fd, "rb") as f:
          owner = os.fstat(f.fileno()).st_uid
          blob = f.read()
        if (owner == os.getuid() and len(blob) == 144 and blob[:8] == _ID_MAGIC and
            # a file a crashed earlier launch left behind is not this launch's
            time.time() - struct.unpack("<d", blob[8:16])[0] < timeout_s + 60.0):
          return blob[16:]
      except OSError:  # not there yet (or a symlink: O_NOFOLLOW refuses it)
        pass
      if time.monotonic() > deadline:
        raise TimeoutError("rank %d: no RCCL unique id at %s" % (rank, path))
      time.sleep(0.01)

  @classmethod
  def from_env(cls, handle, timeout_s: float = 120.0) -> "Comm":
    """Communicator of the launch described by RANK / WORLD_SIZE (what
    `python -m torch.distributed.run` exports).  World size 1 needs no RCCL."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if size == 1:
      return LocalComm()
    # The ranks first meet over TCP (always possible on one node), hand the RCCL unique id
    # over that way and then agree on whether RCCL came up on EVERY rank; if not, the TCP
    # communicator itself carries the job's few hundred bytes (`comm.note` says why).
    sock = SocketComm.from_env(rank, size, timeout_s)
    head, why = b"\0" + bytes(128), ""
    if rank == 0:
      try:
        head = b"\1" + cls.new_unique_id()
      except Exception as exc:  # pylint: disable=broad-except
        why = "rank 0: %s" % exc
    head = sock.broadcast_bytes(head, 129)
    # ncclCommInitRank blocks until EVERY rank has entered it: a rank whose local preflight
    # fails (device not settable, librccl not loadable) must say so BEFORE anyone enters
    pre = ""
    try:
      from spectralcluster_amd import _lib
      if not handle.lib.sc_comm_available():
        pre = "rank %d: librccl cannot be loaded" % rank
      elif handle.lib.sc_synchronize(handle.raw) != _lib.SC_OK:
        pre = "rank %d: device not usable: %s" % (rank, handle.last_error())
    except Exception as exc:  # pylint: disable=broad-except
      pre = "rank %d: %s" % (rank, exc)
    ready = sock.allgather_bytes((b"\0" if pre else b"\1") + pre.encode()[:200])
    comm = None
    if head[:1] == b"\1" and all(r[:1] == b"\1" for r in ready):
      # ncclCommInitRank cannot be cancelled: it runs on a daemon thread, and a rank that does
      # not get its communicator within SC_COMM_INIT_TIMEOUT seconds (default 90: a first
      # communicator of 8 ranks takes a few seconds) reports that instead of hanging the job --
      # the ranks then agree on the TCP communicator below.  (No N > 1 RCCL communicator has
      # ever been brought up in this project's test pool: the watchdog is for the day one is.)
      import threading
      box = {}

      def bring_up():
        try:
          box["comm"] = cls(handle, rank, size, head[1:])
        except Exception as exc:  # pylint: disable=broad-except
          box["why"] = "rank %d: %s" % (rank, exc)

      worker = threading.Thread(target=bring_up, daemon=True)
      worker.start()
      limit = float(os.environ.get("SC_COMM_INIT_TIMEOUT", "90"))
      worker.join(limit)
      if worker.is_alive():
        why = "rank %d: ncclCommInitRank did not return within %g s" % (rank, limit)
      else:
        comm = box.get("comm")
        why = box.get("why", why)
    elif pre:
      why = pre
    reports = sock.allgather_bytes((b"\1" if comm is not None else b"\0") + why.encode()[:200])
    if all(r[:1] == b"\1" for r in reports):
      comm.barrier()
      sock.close()
      return comm
    if comm is not None:
      comm.close()
    sock.note = "; ".join(r[1:].decode(errors="replace") for r in reports if r[1:])
    return sock

  def _check(self, rc, what):
    if rc != 0:
      from spectralcluster_amd import _lib
      msg = self._lib.sc_comm_last_error(self._c)
      raise _lib.DeviceLibraryError("%s failed (%d): %s"
                                    % (what, rc, msg.decode() if msg else ""))

  def broadcast_bytes(self, data, nbytes, root=0):
    buf = ctypes.create_string_buffer(int(nbytes))
    if self.rank == root:
      buf.raw = bytes(data)
    self._check(self._lib.sc_comm_broadcast(self._c, buf, int(nbytes), int(root)),
                "sc_comm_broadcast")
    return buf.raw

  def allgather_bytes(self, data):
    data = bytes(data)
    out = ctypes.create_string_buffer(len(data) * self.size)
    self._check(self._lib.sc_comm_allgather(self._c, data, out, len(data)),
                "sc_comm_allgather")
    raw = out.raw
    return [raw[r * len(data):(r + 1) * len(data)] for r in range(self.size)]

  def allreduce_max(self, value):
    v = (ctypes.c_double * 1)(float(value))
    self._check(self._lib.sc_comm_allreduce_max(self._c, v, 1), "sc_comm_allreduce_max")
    return float(v[0])

  def close(self):
    if getattr(self, "_c", None):
      self._lib.sc_comm_destroy(self._c)
      self._c = None


# --------------------------------------------------------------------------------
# collectives on arrays
# --------------------------------------------------------------------------------
_DTYPES = [np.float64, np.int64, np.int32, np.int8, np.uint8]


def launch_local_ranks(argv: typing.Sequence[str], nproc: int,
                       timeout_s: typing.Optional[float] = None) -> int:
  """One process per GPU of THIS node without an external launcher: starts `nproc` copies of
  `argv` with what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would export
  (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT)
  and waits for them.  Rank 0 keeps this process's stdout (its ONE JSON line); the other ranks'
  stdout goes to stderr.  The ranks share this process as parent, which is what names their
  rendezvous file (`_id_file`).  Returns the largest exit code (124 on timeout: every rank still
  running is killed -- by PID, they are ours)."""
  import subprocess
  import sys
  if nproc < 1:
    raise ValueError("nproc must be >= 1")
  probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
  probe.bind(("127.0.0.1", 0))
  port = probe.getsockname()[1]
  probe.close()
  procs = []
  for rank in range(nproc):
    env = dict(os.environ)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc),
                "LOCAL_WORLD_SIZE": str(nproc), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(port)})
    # (the host driver only supports dmabuf IPC: RCCL needs this on the GPU boxes of this pool)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs.append(subprocess.Popen(list(argv), env=env,
                                  stdout=None if rank == 0 else sys.stderr))
  deadline = None if timeout_s is None else time.monotonic() + timeout_s
  worst = 0
  for p in procs:
    try:
      rc = p.wait(None if deadline is None else max(0.0, deadline - time.monotonic()))
    except subprocess.TimeoutExpired:
      for q in procs:
        if q.poll() is None:
          q.kill()
      for q in procs:
        q.wait()
      return 124
    worst = max(worst, rc if rc >= 0 else 128 - rc)
  return worst


def broadcast_array(comm: Comm, arr: typing.Optional[np.ndarray], root: int = 0) -> np.ndarray:
  """Broadcast an ndarray from `root` (a 72-byte header -- ndim, dtype, shape -- then the
  data).  Used for embeddings (n*d*8 bytes) and packed configs."""
  if comm.rank == root:
    a = np.ascontiguousarray(arr)
    code = [i for i, t in enumerate(_DTYPES) if a.dtype == t]
    if not code or a.ndim > 7:
      raise TypeError("broadcast_array: unsupported dtype/rank %s %d" % (a.dtype, a.ndim))
    header = struct.pack("<9q", a.ndim, code[0], *(list(a.shape) + [0] * (7 - a.ndim)))
  else:
    a, header = None, None
  header = comm.broadcast_bytes(header, 72, root)
  meta = struct.unpack("<9q", header)
  shape = tuple(int(v) for v in meta[2:2 + meta[0]])
  dtype = np.dtype(_DTYPES[meta[1]])
  nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
  if nbytes == 0:
    return np.empty(shape, dtype=dtype)
  payload = comm.broadcast_bytes(a.tobytes() if comm.rank == root else None, nbytes, root)
  return np.frombuffer(payload, dtype=dtype).reshape(shape).copy()


def predict_batch_sharded(
    comm: Comm,
    predict_fn: typing.Optional[typing.Callable[[np.ndarray], np.ndarray]],
    utterances,
    predict_many_fn: typing.Optional[typing.Callable] = None,
    sizes: typing.Optional[typing.Sequence[int]] = None) -> typing.List[np.ndarray]:
  """Every rank holds (or has been broadcast) the same utterance list; rank r runs
  `predict_fn` on its LPT share (or `predict_many_fn(list_of_arrays) -> list_of_labels`
  on the whole share at once, e.g. a multi-stream batch).  With `sizes` given (the
  n_samples of every utterance), `utterances` only needs to hold this rank's share
  (`utterances[i]` for the i it owns) -- the partition depends on the sizes alone.  The
  labels travel as one ragged int32 slab per rank (the concatenation of its share, padded
  to the longest slab: O(total samples), not O(slots x longest utterance)).  Every rank
  returns the complete, input-ordered result."""
  world, rank = comm.size, comm.rank
  sizes = ([int(u.shape[0]) for u in utterances] if sizes is None
           else [int(v) for v in sizes])
  owned = lpt_assignment(sizes, world)
  slab_len = max((sum(sizes[i] for i in o) for o in owned), default=0)
  if predict_many_fn is not None:
    local = predict_many_fn([utterances[idx] for idx in owned[rank]])
  else:
    local = [predict_fn(utterances[idx]) for idx in owned[rank]]
  mine = np.full(slab_len, -1, dtype=np.int32)
  pos = 0
  for idx, lab in zip(owned[rank], local):
    lab = np.asarray(lab)
    if lab.shape[0] != sizes[idx]:
      raise ValueError("predict returned %d labels for an utterance of %d"
                       % (lab.shape[0], sizes[idx]))
    mine[pos:pos + sizes[idx]] = lab
    pos += sizes[idx]
  slabs = comm.allgather_bytes(mine.tobytes())
  out = [None] * len(sizes)
  for r in range(world):
    block = np.frombuffer(slabs[r], dtype=np.int32)
    pos = 0
    for idx in owned[r]:
      out[idx] = block[pos:pos + sizes[idx]].astype(np.int64)
      pos += sizes[idx]
  return out


def sweep_owner(index: int, world: int) -> int:
  """Rank that evaluates grid[index] of an AutoTune level: the ONE dealing rule
  (`autotune_sharded` deals by it, `predict_autotune_distributed` finds the winner's rank by
  it)."""
  return index % world


def autotune_sharded(
    comm: Comm,
    evaluate_fn: typing.Optional[typing.Callable[[float], typing.Tuple[float, int]]],
    grid: typing.Sequence[float],
    evaluate_share_fn: typing.Optional[typing.Callable] = None
) -> typing.Tuple[np.ndarray, np.ndarray]:
  """One AutoTune search level (reference autotune.py:98-111): rank r evaluates
  grid[r::world]; `evaluate_fn(p) -> (ratio, n_clusters)`, or `evaluate_share_fn(list of p)
  -> list of (ratio, n_clusters)` for the rank's whole share at once.  Returns the full
  (ratios, n_clusters) arrays on every rank (all-gather of 2 doubles per p)."""
  world, rank = comm.size, comm.rank
  per = (len(grid) + world - 1) // world
  mine = np.full((per, 2), np.nan)
  share = [float(grid[i]) for i in range(len(grid)) if sweep_owner(i, world) == rank]
  if evaluate_share_fn is not None:
    results = evaluate_share_fn(share) if share else []
  else:
    results = [evaluate_fn(p) for p in share]
  for s, (ratio, k) in enumerate(results):
    mine[s, 0] = float(ratio)
    mine[s, 1] = float(k)
  blocks = comm.allgather_bytes(mine.tobytes())
  ratios = np.full(len(grid), np.nan)
  ks = np.zeros(len(grid), dtype=np.int64)
  for r in range(world):
    block = np.frombuffer(blocks[r], dtype=np.float64).reshape(per, 2)
    for s, i in enumerate(i for i in range(len(grid)) if sweep_owner(i, world) == r):
      ratios[i] = block[s, 0]
      ks[i] = int(block[s, 1])
  return ratios, ks


def first_strict_minimum(ratios: np.ndarray) -> int:
  """Index AutoTune.tune keeps: strict `<` scan, first index wins ties
  (reference autotune.py:106-111)."""
  best, best_i = np.inf, -1
  for i, r in enumerate(ratios):
    if r < best:
      best, best_i = r, i
  return best_i


# --------------------------------------------------------------------------------
# thin wrappers around a SpectralClusterer (device compute on this rank's GPU)
# --------------------------------------------------------------------------------
def predict_batch_distributed(comm: Comm, clusterer,
                              utterances: typing.Sequence[np.ndarray],
                              streams: typing.Optional[int] = None,
                              group: typing.Optional[int] = None) -> typing.List[np.ndarray]:
  """BASELINE config 5: utterances partitioned over the ranks (LPT), each rank runs its
  share as one batch on its own GPU (grouped by default, `streams` for the multi-stream
  form: see SpectralClusterer.predict_batch), labels all-gathered."""
  return predict_batch_sharded(
      comm, None, utterances,
      predict_many_fn=lambda share: clusterer.predict_batch(share, streams=streams,
                                                            group=group))


def predict_autotune_distributed(comm: Comm, clusterer, embeddings: np.ndarray,
                                 constraint_matrix=None) -> np.ndarray:
  """BASELINE config 4: every rank holds the embeddings (broadcast them first if only
  rank 0 has them), recomputes the affinity locally (cheaper than shipping n^2 doubles),
  evaluates its share of each AutoTune search level, all-gathers (ratio, n_clusters);
  the rank that evaluated the winner adopts its eigenvectors from the sweep
  (sc_sweep_adopt), runs k-means and broadcasts the labels (n int32): identical labels
  everywhere, no n x n or n x k matrix ever crosses xGMI.  Constraints are handled as in
  predict() (reference spectral_clusterer.py:259-264, 137-142)."""
  from spectralcluster_amd import _lib
  tuner = clusterer.autotune
  if tuner is None:
    raise ValueError("clusterer.autotune is not set")
  handle = clusterer._handle()
  n = embeddings.shape[0]
  constrained = clusterer._set_constraint(handle, n, constraint_matrix)
  clusterer._upload(handle, embeddings)
  if constrained and clusterer.constraint_options.apply_before_refinement:
    handle.check(handle.lib.sc_apply_constraint(handle.raw, clusterer.build_config()))

  def evaluate(ps):
    # this rank's share of a level as one grouped sweep.  A rank that fails mid-sweep must
    # still reach the all-gather, or the others hang: report NaN and raise after the level
    try:
      diags = clusterer._eig_sweep(handle, ps)
      return [(tuner.ratio(p, d.max_delta), int(d.n_clusters_raw)) for p, d in zip(ps, diags)]
    except Exception as exc:  # pylint: disable=broad-except
      evaluate.error = exc
      return [(float("nan"), 0)] * len(ps)

  evaluate.error = None
  evaluate.last_p = clusterer.refinement_options.p_percentile
  evaluate.last_level = []

  def evaluate_many(ps):
    evaluate.last_p = ps[-1]
    evaluate.last_level = list(ps)
    ratios, ks = autotune_sharded(comm, None, ps, evaluate_share_fn=evaluate)
    if evaluate.error is not None:
      raise evaluate.error
    if np.isnan(ratios).any():
      raise RuntimeError("an AutoTune evaluation failed on another rank")
    return [(float(r), p, int(k)) for r, p, k in zip(ratios, ps, ks)]

  _, n_clusters, best_p = tuner.tune(None, evaluate_many=evaluate_many)
  clusterer.last_best_p = best_p
  # the reference's closure leaves p_percentile at the LAST evaluated value
  # (spectral_clusterer.py:277); keep that observable state, like predict()
  clusterer.refinement_options.p_percentile = evaluate.last_p
  if clusterer.min_clusters is not None:
    n_clusters = max(n_clusters, clusterer.min_clusters)

  def finish():
    # the winner's eigenvectors (adopted from this rank's sweep when it evaluated the winner)
    # + k-means
    diag = clusterer._adopt_or_evaluate(handle, best_p)
    labels = np.empty(n, dtype=np.int64)
    handle.check(handle.lib.sc_cluster(handle.raw, clusterer.build_config(best_p), n_clusters,
                                       _lib.as_int64_p(labels), diag))
    return labels

  last_level = [float(p) for p in evaluate.last_level]
  if comm.size == 1 or float(best_p) not in last_level:
    return finish()  # (a winner from an earlier level: every rank re-evaluates it)
  # The rank that evaluated the winner still holds its eigenvectors: it alone runs k-means and
  # broadcasts n int32 labels; the other ranks neither refine nor solve again.
  owner = sweep_owner(last_level.index(float(best_p)), comm.size)
  payload, error = None, None
  if comm.rank == owner:
    try:
      payload = finish().astype(np.int32).tobytes()
    except Exception as exc:  # pylint: disable=broad-except
      error, payload = exc, np.full(n, -2, dtype=np.int32).tobytes()  # the others must not hang
  labels = np.frombuffer(comm.broadcast_bytes(payload, n * 4, root=owner), dtype=np.int32)
  if comm.rank != owner:
    # this rank's resident eigenvectors / diagnostics belong to its own share of the sweep,
    # not to the winner: do not leave them looking like the result
    clusterer.last_diag = None
  if error is not None:
    raise error
  if n and labels[0] == -2:
    raise RuntimeError("the AutoTune winner could not be clustered on rank %d" % owner)
  return labels.astype(np.int64)
