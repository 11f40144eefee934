"""CPU, world_size 2, backend gloo: the multi-GPU partitioning layer
(`spectralcluster_amd/multigpu.py`).  The per-unit compute is injected (here the
CPU oracle on tiny problems) so that sharding, broadcast and gather are exercised
exactly as they run over RCCL on the GPU box."""

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
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  import torch.distributed as dist
  import spectral_oracle as so
  from spectralcluster_amd import multigpu
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    cfg = so.icassp2018_config()
    # --- broadcast: only rank 0 has the data
    x0 = so.blobs(90, 8, 3, seed=1) if rank == 0 else None
    x = multigpu.broadcast_array(x0)
    assert x.shape == (90, 8) and np.array_equal(x, so.blobs(90, 8, 3, seed=1))
    # --- batched utterances: LPT shard + all_gather of ragged labels
    sizes = [60, 35, 80, 50, 45]
    utts = [so.blobs(n, 8, 2 + (i % 2), seed=10 + i) for i, n in enumerate(sizes)]
    ran = []

    def predict_fn(u):
      ran.append(u.shape[0])
      return so.predict(u, cfg)

    got = multigpu.predict_batch_sharded(predict_fn, utts)
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

    ratios, ks = multigpu.autotune_sharded(evaluate, grid)
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
