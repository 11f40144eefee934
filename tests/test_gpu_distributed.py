"""The multi-GPU wrappers with REAL device compute.  The pool has one GPU per box, so
(a) two ranks share device 0 over the test-only gloo transport (RCCL refuses two ranks on
one GPU) -- the sharding code is the same one that runs over RCCL with one GPU per rank --
and (b) the RCCL communicator itself (`sc_comm_*`, C ABI) is exercised with world size 1,
every collective going through librccl on the device."""

import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
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
  os.environ["SPECTRALCLUSTER_AMD_DEVICE"] = "0"
  import torch.distributed as dist
  import spectral_oracle as so
  import spectralcluster_amd as sca
  from spectralcluster_amd import multigpu
  import _gloo_comm
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    comm = _gloo_comm.GlooComm()
    # config 5 in miniature
    rng = np.random.default_rng(5)
    utts = [so.blobs(int(n), 32, int(k), seed=i)
            for i, (n, k) in enumerate(zip(rng.integers(130, 700, 14), rng.integers(2, 5, 14)))]
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                              refinement_options=sca.configs.icassp2018_refinement_options)
    got = multigpu.predict_batch_distributed(comm, c, utts, streams=2)
    for u, g in zip(utts, got):
      assert np.array_equal(g, c.predict(u))
    # config 4 in miniature: AutoTune sweep sharded over the ranks
    x0 = so.blobs(512, 64, 6, 512) if rank == 0 else None
    x = multigpu.broadcast_array(comm, x0)
    def make():
      return sca.SpectralClusterer(
          min_clusters=2, max_clusters=20, laplacian_type=sca.LaplacianType.GraphCut,
          refinement_options=sca.RefinementOptions(
              refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
          autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                init_search_step=0.025, search_level=2))
    sharded = multigpu.predict_autotune_distributed(comm, make(), x)
    serial = make().predict(x)
    assert np.array_equal(sharded, serial)
    np.save(os.path.join(out_dir, "labels_%d.npy" % rank), sharded)
  finally:
    dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo(tmp_path):
  import torch.multiprocessing as mp
  mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
  assert np.array_equal(np.load(tmp_path / "labels_0.npy"), np.load(tmp_path / "labels_1.npy"))


_RCCL_WORLD1 = r"""
import os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so
import spectralcluster_amd as sca
from spectralcluster_amd import _lib, multigpu
assert "torch" not in sys.modules
handle = _lib.default_handle()
assert handle.lib.sc_comm_available() == 1
uid = multigpu.RcclComm.new_unique_id()
assert len(uid) == 128 and uid != bytes(128)
comm = multigpu.RcclComm(handle, 0, 1, uid)
assert (comm.rank, comm.size) == (0, 1)
rng = np.random.default_rng(0)
x = rng.standard_normal((300, 17))
assert np.array_equal(multigpu.broadcast_array(comm, x), x)
lab = rng.integers(0, 7, 1000).astype(np.int32)
assert comm.allgather_bytes(lab.tobytes()) == [lab.tobytes()]
assert comm.allreduce_max(3.25) == 3.25
comm.barrier()
big = rng.integers(0, 255, 5 << 20, dtype=np.uint8)  # staging buffer regrowth
assert comm.broadcast_bytes(big.tobytes(), big.size, 0) == big.tobytes()
utts = [so.blobs(n, 16, 3, seed=n) for n in (150, 260, 200)]
c = sca.configs.icassp2018_clusterer
got = multigpu.predict_batch_distributed(comm, c, utts, streams=1)
for u, g in zip(utts, got):
  assert np.array_equal(g, c.predict(u))
# config 4 through the same communicator
x4 = so.blobs(512, 64, 6, 512)
def make():
  return sca.SpectralClusterer(
      min_clusters=2, max_clusters=20, laplacian_type=sca.LaplacianType.GraphCut,
      refinement_options=sca.RefinementOptions(
          refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
      autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                            init_search_step=0.025, search_level=1))
assert np.array_equal(multigpu.predict_autotune_distributed(comm, make(), x4), make().predict(x4))
comm.close()
# the launch-environment path: rank 0 of a world of 1 needs no RCCL; a world of 2 with only
# this rank present times out on the id file instead of hanging
os.environ.update(WORLD_SIZE="1", RANK="0")
assert isinstance(multigpu.RcclComm.from_env(handle), multigpu.LocalComm)
print("RCCL_WORLD1_OK")
"""


def test_rccl_comm_world_1(tmp_path):
  """RCCL behind the C ABI: unique id, ncclCommInitRank, broadcast / all-gather /
  max-reduce staged through the device, the two sharded drivers, destroy.  Runs in a fresh
  interpreter, the way the product runs: no PyTorch in the process (its wheel carries a
  second RCCL + HIP runtime)."""
  import subprocess
  script = tmp_path / "rccl_world1.py"
  script.write_text(_RCCL_WORLD1)
  r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True,
                     timeout=300)
  assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout + r.stderr


def test_rccl_comm_from_env_world_1(handle, monkeypatch):
  from spectralcluster_amd import multigpu
  monkeypatch.setenv("WORLD_SIZE", "1")
  monkeypatch.setenv("RANK", "0")
  assert isinstance(multigpu.RcclComm.from_env(handle), multigpu.LocalComm)
