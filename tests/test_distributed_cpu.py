"""CPU, world_size 2: the multi-GPU partitioning layer (`spectralcluster_amd/multigpu.py`).
The per-unit compute is injected (here the CPU oracle on tiny problems) and the byte
transport is a gloo adapter (tests/_gloo_comm.py), so that sharding, broadcast and gather
run exactly the code that sits on RCCL (`multigpu.RcclComm`, C ABI `sc_comm_*`) on the GPU
box."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, out_dir):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  import torch.distributed as dist
  import spectral_oracle as so
  from spectralcluster_amd import multigpu
  import _gloo_comm
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    comm = _gloo_comm.GlooComm()
    assert comm.allreduce_max(float(rank)) == world - 1
    cfg = so.icassp2018_config()
    # --- broadcast: only rank 0 has the data
    x0 = so.blobs(90, 8, 3, seed=1) if rank == 0 else None
    x = multigpu.broadcast_array(comm, x0)
    assert x.shape == (90, 8) and np.array_equal(x, so.blobs(90, 8, 3, seed=1))
    # --- batched utterances: LPT shard + all_gather of ragged labels
    sizes = [60, 35, 80, 50, 45]
    utts = [so.blobs(n, 8, 2 + (i % 2), seed=10 + i) for i, n in enumerate(sizes)]
    ran = []

    def predict_fn(u):
      ran.append(u.shape[0])
      return so.predict(u, cfg)

    got = multigpu.predict_batch_sharded(comm, predict_fn, utts)
    want = [so.predict(u, cfg) for u in utts]
    for g, w in zip(got, want):
      assert g.dtype == np.int64 and np.array_equal(g, w)
    owned = multigpu.lpt_assignment(sizes, world)
    assert sorted(ran) == sorted(sizes[i] for i in owned[rank])
    # --- AutoTune sweep: p-grid round-robin + all_gather of (ratio, n_clusters)
    a = so.affinity(so.blobs(70, 8, 3, seed=3))
    gcfg = so.icassp2018_config(laplacian_type=so.LAPLACIAN_GRAPH_CUT, max_clusters=6)
    grid = so.autotune_range(0.55, 0.95, 0.05)
    import dataclasses

    def evaluate(p):
      _, k, delta = so.eig_ncluster(a, dataclasses.replace(gcfg, p_percentile=p))
      return np.sqrt(1 - p) / delta, k

    ratios, ks = multigpu.autotune_sharded(comm, evaluate, grid)
    _, _, best_p, seen = so.autotune_search(a, gcfg, 0.55, 0.95, 0.05)
    np.testing.assert_allclose(ratios, [seen[p] for p in grid], rtol=1e-12)
    assert grid[multigpu.first_strict_minimum(ratios)] == best_p
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), ratios)
  finally:
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
  import torch.multiprocessing as mp
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r0 = np.load(tmp_path / "ok_0.npy")
  r1 = np.load(tmp_path / "ok_1.npy")
  assert np.array_equal(r0, r1)  # every rank ends with the same, complete answer


def test_local_comm_is_identity():
  """World of one (what bench.py --gpus 1 and single-GPU users get): no transport."""
  sys.path.insert(0, ROOT)
  from spectralcluster_amd import multigpu
  comm = multigpu.LocalComm()
  a = np.arange(12, dtype=np.float64).reshape(3, 4)
  assert np.array_equal(multigpu.broadcast_array(comm, a), a)
  labs = multigpu.predict_batch_sharded(
      comm, lambda u: np.arange(u.shape[0]) % 3, [np.zeros((5, 2)), np.zeros((9, 2))])
  assert [l.tolist() for l in labs] == [[0, 1, 2, 0, 1], [0, 1, 2, 0, 1, 2, 0, 1, 2]]
  assert labs[0].dtype == np.int64
  ratios, ks = multigpu.autotune_sharded(comm, lambda p: (1.0 - p, 3), [0.5, 0.7])
  assert ratios.tolist() == [0.5, 1.0 - 0.7] and ks.tolist() == [3, 3]


def test_product_code_never_imports_torch():
  """north_star: Python host + C ABI, no PyTorch -- neither the package nor bench.py."""
  import re
  offenders = []
  files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
  for dirpath, _, names in os.walk(os.path.join(ROOT, "spectralcluster_amd")):
    files += [os.path.join(dirpath, f) for f in names if f.endswith(".py")]
  for path in files:
    with open(path) as f:
      for line in f:
        if re.match(r"\s*(import|from)\s+torch\b", line):
          offenders.append(path)
  assert not offenders, offenders


def _id_worker(rank, path, out):
  sys.path.insert(0, ROOT)
  os.environ["SC_COMM_ID_FILE"] = path
  from spectralcluster_amd import multigpu
  uid = multigpu.RcclComm.exchange_id(rank, lambda: bytes(range(128)), timeout_s=30.0)
  with open("%s.%d" % (out, rank), "wb") as f:
    f.write(uid)


def test_unique_id_rendezvous_between_processes(tmp_path):
  """The out-of-band half of RcclComm.from_env (no GPU needed): rank 0 publishes the id
  atomically, late and early readers all get the same 128 bytes."""
  import multiprocessing as mp
  ctx = mp.get_context("spawn")
  path, out = str(tmp_path / "id"), str(tmp_path / "got")
  readers = [ctx.Process(target=_id_worker, args=(r, path, out)) for r in (1, 2)]
  for p in readers:
    p.start()          # readers first: they must wait for the file
  writer = ctx.Process(target=_id_worker, args=(0, path, out))
  writer.start()
  for p in readers + [writer]:
    p.join(60)
    assert p.exitcode == 0
  for r in (0, 1, 2):
    assert open("%s.%d" % (out, r), "rb").read() == bytes(range(128))
  sys.path.insert(0, ROOT)
  from spectralcluster_amd import multigpu
  os.environ["SC_COMM_ID_FILE"] = str(tmp_path / "never")
  try:
    with pytest.raises(TimeoutError):
      multigpu.RcclComm.exchange_id(1, lambda: b"", timeout_s=0.2)
  finally:
    del os.environ["SC_COMM_ID_FILE"]


def _fallback_worker(rank, world, path, out):
  sys.path.insert(0, ROOT)
  os.environ.update({"SC_COMM_ID_FILE": path, "WORLD_SIZE": str(world), "RANK": str(rank),
                     "MASTER_ADDR": "127.0.0.1"})
  import numpy as np
  from spectralcluster_amd import multigpu
  # no device handle here: RCCL cannot come up on any rank, every rank must notice and agree
  comm = multigpu.RcclComm.from_env(None, timeout_s=60.0)
  assert isinstance(comm, multigpu.SocketComm) and comm.size == world and comm.note
  got = comm.allgather_bytes(bytes([rank]) * 3)
  assert got == [bytes([r]) * 3 for r in range(world)]
  assert comm.broadcast_bytes(b"hello" if rank == 1 else None, 5, root=1) == b"hello"
  assert comm.allreduce_max(float(rank)) == float(world - 1)
  comm.barrier()
  # the sharded batch driver on top of it: labels of every utterance on every rank
  utts = [np.full((5 + i, 2), float(i)) for i in range(7)]
  labels = multigpu.predict_batch_sharded(
      comm, lambda u: np.full(u.shape[0], int(u[0, 0]), dtype=np.int64), utts)
  assert [int(l[0]) for l in labels] == list(range(7))
  assert [l.shape[0] for l in labels] == [5 + i for i in range(7)]
  ratios, ks = multigpu.autotune_sharded(comm, lambda p: (1.0 - p, 3), [0.5, 0.6, 0.7, 0.8])
  assert np.allclose(ratios, [0.5, 0.4, 0.3, 0.2]) and list(ks) == [3, 3, 3, 3]
  comm.close()
  with open("%s.%d" % (out, rank), "w") as f:
    f.write("ok")


def test_tcp_fallback_when_rccl_cannot_come_up(tmp_path):
  """RcclComm.from_env on a launch where RCCL fails on every rank: the ranks meet over TCP,
  agree that RCCL is out, and the same communicator carries the collectives of the sharded
  drivers (three processes, no GPU)."""
  import multiprocessing as mp
  ctx = mp.get_context("spawn")
  path, out = str(tmp_path / "id"), str(tmp_path / "done")
  procs = [ctx.Process(target=_fallback_worker, args=(r, 3, path, out)) for r in (2, 1, 0)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(120)
    assert p.exitcode == 0
  assert all(os.path.exists("%s.%d" % (out, r)) for r in range(3))


def _hung_init_worker(rank, world, path, out):
  sys.path.insert(0, ROOT)
  os.environ.update({"SC_COMM_ID_FILE": path, "WORLD_SIZE": str(world), "RANK": str(rank),
                     "MASTER_ADDR": "127.0.0.1", "SC_COMM_INIT_TIMEOUT": "1.5"})
  import time
  from spectralcluster_amd import multigpu

  class Lib:  # a library whose preflight passes ...
    @staticmethod
    def sc_comm_available():
      return 1

    @staticmethod
    def sc_synchronize(raw):
      return 0

  class Handle:
    lib, raw = Lib(), None

  def hang(self, handle, rank, size, unique_id):  # ... and whose ncclCommInitRank never returns
    time.sleep(3600)

  multigpu.RcclComm.__init__ = hang
  multigpu.RcclComm.new_unique_id = staticmethod(lambda: bytes(128))
  t0 = time.monotonic()
  comm = multigpu.RcclComm.from_env(Handle(), timeout_s=60.0)
  assert isinstance(comm, multigpu.SocketComm) and "did not return within" in comm.note
  assert time.monotonic() - t0 < 30.0
  assert comm.allgather_bytes(bytes([rank])) == [bytes([r]) for r in range(world)]
  comm.close()
  with open("%s.%d" % (out, rank), "w") as f:
    f.write("ok")
  os._exit(0)  # (the hung daemon thread must not keep the process)


def test_tcp_fallback_when_rccl_init_hangs(tmp_path):
  """A communicator bring-up that never returns (ncclCommInitRank cannot be cancelled) must not
  hang the job: after SC_COMM_INIT_TIMEOUT every rank reports it and the TCP communicator
  carries the collectives."""
  import multiprocessing as mp
  ctx = mp.get_context("spawn")
  path, out = str(tmp_path / "id"), str(tmp_path / "done")
  procs = [ctx.Process(target=_hung_init_worker, args=(r, 2, path, out)) for r in (1, 0)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(120)
    assert p.exitcode == 0
  assert all(os.path.exists("%s.%d" % (out, r)) for r in range(2))


def test_rendezvous_file_rejects_stale_and_planted_files(tmp_path, monkeypatch):
  """the id file of a crashed earlier launch (old timestamp), a truncated blob and
  a pre-planted symlink must not be taken for this launch's rendezvous blob."""
  import struct
  import threading
  import time
  from spectralcluster_amd import multigpu
  path = tmp_path / "launch.id"
  monkeypatch.setenv("SC_COMM_ID_FILE", str(path))
  uid = bytes(range(128))
  # stale: right format, written "an hour ago"
  path.write_bytes(multigpu._ID_MAGIC + struct.pack("<d", time.time() - 3600.0) + bytes(128))
  with pytest.raises(TimeoutError):
    multigpu.RcclComm.exchange_id(1, lambda: b"", timeout_s=0.3)
  # old 128-byte format / truncated
  path.write_bytes(bytes(128))
  with pytest.raises(TimeoutError):
    multigpu.RcclComm.exchange_id(1, lambda: b"", timeout_s=0.2)
  # a symlink somebody planted at the path: never followed by the reader ...
  target = tmp_path / "elsewhere"
  target.write_bytes(multigpu._ID_MAGIC + struct.pack("<d", time.time()) + bytes(128))
  path.unlink()
  path.symlink_to(target)
  with pytest.raises(TimeoutError):
    multigpu.RcclComm.exchange_id(1, lambda: b"", timeout_s=0.2)
  # ... and replaced (not written through) by rank 0; a polling reader then gets the id
  got = {}
  t = threading.Thread(target=lambda: got.setdefault(
      "uid", multigpu.RcclComm.exchange_id(1, lambda: b"", timeout_s=5.0)))
  t.start()
  time.sleep(0.1)
  assert multigpu.RcclComm.exchange_id(0, lambda: uid) == uid
  t.join()
  assert got["uid"] == uid
  assert not path.is_symlink() and target.read_bytes()[16:] == bytes(128)
  assert (path.stat().st_mode & 0o777) == 0o600


def _bench(args, extra_env, timeout=180):
  import subprocess
  env = dict(os.environ)
  for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
    env.pop(name, None)
  env.update(extra_env)
  return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                        capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_launches_its_own_ranks():
  """`python bench.py --gpus N` with no launcher around it starts the N
  ranks itself (multigpu.launch_local_ranks: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
  torch.distributed.run exports them); rank 0's ONE JSON line arrives on the command's stdout.
  SC_BENCH_LAUNCH_ECHO stops the ranks after the TCP rendezvous the real run starts with."""
  import json
  for n in (2, 3):
    res = _bench(["--gpus", str(n), "--steps", "2", "--warmup", "1"],
                 {"SC_BENCH_LAUNCH_ECHO": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks_seen"] == list(range(n))


def test_bench_refuses_a_world_that_is_not_its_gpus_argument():
  """Under a launcher WORLD_SIZE decides how many ranks exist; `--gpus` must say the same
  (round 5's bench.py parsed --gpus and never read it)."""
  res = _bench(["--gpus", "8"], {"SC_BENCH_LAUNCH_ECHO": "1", "WORLD_SIZE": "1", "RANK": "0"})
  assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
  res = _bench([], {"SC_BENCH_LAUNCH_ECHO": "1", "WORLD_SIZE": "1", "RANK": "0"})
  assert res.returncode == 0 and '"n_gpus": 1' in res.stdout


def test_launch_local_ranks_reports_the_worst_exit_code_and_times_out(tmp_path):
  sys.path.insert(0, ROOT)
  from spectralcluster_amd import multigpu
  script = tmp_path / "rank.py"
  script.write_text("import os, sys, time\n"
                    "r = int(os.environ['RANK'])\n"
                    "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                    "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                    "if len(sys.argv) > 1: time.sleep(60)\n"
                    "sys.exit(5 if r == 2 else 0)\n")
  assert multigpu.launch_local_ranks([sys.executable, str(script)], 3) == 5
  assert multigpu.launch_local_ranks([sys.executable, str(script), "hang"], 3, timeout_s=1.0) == 124
