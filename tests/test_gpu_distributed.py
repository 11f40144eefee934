"""GPU + torch.distributed: the multi-GPU wrappers with REAL device compute.  The pool
has one GPU per box, so two ranks share device 0 and use the gloo backend (NCCL refuses
two ranks on one GPU); the sharding / collective code is the same one that runs over
RCCL with one GPU per rank."""

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
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  os.environ["SPECTRALCLUSTER_AMD_DEVICE"] = "0"
  import torch.distributed as dist
  import spectral_oracle as so
  import spectralcluster_amd as sca
  from spectralcluster_amd import multigpu
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    # config 5 in miniature
    rng = np.random.default_rng(5)
    utts = [so.blobs(int(n), 32, int(k), seed=i)
            for i, (n, k) in enumerate(zip(rng.integers(130, 700, 14), rng.integers(2, 5, 14)))]
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                              refinement_options=sca.configs.icassp2018_refinement_options)
    got = multigpu.predict_batch_distributed(c, utts, streams=2)
    for u, g in zip(utts, got):
      assert np.array_equal(g, c.predict(u))
    # config 4 in miniature: AutoTune sweep sharded over the ranks
    x0 = so.blobs(512, 64, 6, 512) if rank == 0 else None
    x = multigpu.broadcast_array(x0)
    def make():
      return sca.SpectralClusterer(
          min_clusters=2, max_clusters=20, laplacian_type=sca.LaplacianType.GraphCut,
          refinement_options=sca.RefinementOptions(
              refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
          autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                init_search_step=0.025, search_level=2))
    sharded = multigpu.predict_autotune_distributed(make(), x)
    serial = make().predict(x)
    assert np.array_equal(sharded, serial)
    np.save(os.path.join(out_dir, "labels_%d.npy" % rank), sharded)
  finally:
    dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo(tmp_path):
  import torch.multiprocessing as mp
  mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
  assert np.array_equal(np.load(tmp_path / "labels_0.npy"), np.load(tmp_path / "labels_1.npy"))
