#!/usr/bin/env python
"""Benchmark of the hot path: SpectralClusterer.predict() calls/s on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  One "step" = one
predict() call on device-resident embeddings (H2D of X happens before the timed
region; the label D2H, n*8 bytes, is inside it).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): n=8192 d=256
synthetic Gaussian blobs (8 speakers), ICASSP2018 refinement, GraphCut Laplacian,
eigengap k in [2, 20], cosine k-means.  Multi-GPU = independent replicas
(batched-utterance partitioning): every rank runs the same per-GPU work, no
data-path collective, weak scaling.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SAMPLES, N_FEATURES, N_SPEAKERS, SEED = 8192, 256, 8, 0
MAX_CLUSTERS = 20
PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak
GEMM_TILE = 128               # gemm_f64.hip block tile


def blobs(n, d, k, seed, noise=0.3):
  """SURVEY.md section 8(d) generator (same as oracle/spectral_oracle.blobs)."""
  rng = np.random.default_rng(seed)
  centers = rng.standard_normal((k, d))
  lab = np.sort(rng.integers(0, k, n))
  return np.ascontiguousarray(centers[lab] + noise * rng.standard_normal((n, d))), lab


def ari(a, b):
  """Adjusted Rand index (Hubert & Arabie 1985) from the contingency table."""
  a = np.asarray(a).ravel()
  b = np.asarray(b).ravel()
  _, ai = np.unique(a, return_inverse=True)
  _, bi = np.unique(b, return_inverse=True)
  table = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
  np.add.at(table, (ai, bi), 1)
  pairs = lambda t: int((t * (t - 1) // 2).sum())
  both, rows, cols = pairs(table), pairs(table.sum(axis=1)), pairs(table.sum(axis=0))
  total = a.size * (a.size - 1) // 2
  expected = rows * cols / total if total else 0.0
  top = 0.5 * (rows + cols)
  return 1.0 if top == expected else float((both - expected) / (top - expected))


def cpu_baseline_leg(gpu_clusterer):
  """Oracle (NumPy port of the reference, np.linalg.eig) on the host cores, on a
  bounded sample of the same workload: n=2048 instead of 8192 (the full size costs
  ~160 s per call on 8 cores; cost is ~n^3)."""
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  import spectral_oracle as so
  n_s = 2048
  x = so.blobs(n_s, N_FEATURES, N_SPEAKERS, SEED)
  cfg = so.icassp2018_config(laplacian_type=so.LAPLACIAN_GRAPH_CUT,
                             max_clusters=MAX_CLUSTERS)
  reps, spent, labels = 0, 0.0, None
  while reps < 3 and spent < 20.0:
    t0 = time.perf_counter()
    labels = so.predict(x, cfg)
    spent += time.perf_counter() - t0
    reps += 1
  cpu_s = spent / reps
  # same sample on the GPU, for an apples-to-apples ratio
  gpu_clusterer.predict(x)
  t0 = time.perf_counter()
  for _ in range(5):
    glab = gpu_clusterer.predict(x)
  gpu_s = (time.perf_counter() - t0) / 5
  try:
    from threadpoolctl import threadpool_info
    threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
  except Exception:
    threads = os.cpu_count() or 1
  return {
      "value": 1.0 / cpu_s, "unit": "calls/s", "cores": int(threads), "kind": "port",
      "sample": ("oracle/spectral_oracle.predict (np.linalg.eig) on n=%d d=%d k=%d "
                 "blobs, same config; %d reps; n=8192 extrapolates by (8192/2048)^3"
                 % (n_s, N_FEATURES, N_SPEAKERS, reps)),
      "seconds_per_call": cpu_s,
      "extrapolated_n8192_calls_per_s": 1.0 / (cpu_s * (N_SAMPLES / n_s) ** 3),
      "gpu_same_sample_calls_per_s": 1.0 / gpu_s,
      "ari_gpu_vs_cpu_sample": ari(glab, labels),
  }


def concurrent_leg(sca, _lib, cfg, x, steps, streams=2):
  """Throughput of independent predict() calls issued from `streams` host threads on
  `streams` handles (HIP streams) of the same GPU: the single-workgroup phases of one
  call (Rayleigh-Ritz, k-means) overlap the GEMM of the other.  Extra information only;
  the headline `value` stays the single-stream number."""
  import threading
  pool = _lib.handle_pool(None, streams)
  n, d = x.shape
  per = max(1, steps // streams)

  def worker(h, out):
    lab = np.empty(n, dtype=np.int64)
    diag = _lib.ScDiag()
    h.check(h.lib.sc_set_embeddings(h.raw, _lib.as_double_p(x), n, d))
    h.check(h.lib.sc_run_resident(h.raw, cfg, _lib.as_int64_p(lab), diag))  # warm-up
    out.append((h, lab, diag))

  ready = []
  for h in pool:
    worker(h, ready)

  def run(h, lab, diag):
    for _ in range(per):
      h.check(h.lib.sc_run_resident(h.raw, cfg, _lib.as_int64_p(lab), diag))

  for h, _, _ in ready:
    h.check(h.lib.sc_synchronize(h.raw))
  t0 = time.perf_counter()
  threads = [threading.Thread(target=run, args=r) for r in ready]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  dt = time.perf_counter() - t0
  return {"streams": streams, "calls": per * streams, "value": per * streams / dt,
          "unit": "calls/s"}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-concurrent", action="store_true")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  dist = None
  if world > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

  import spectralcluster_amd as sca
  from spectralcluster_amd import _lib

  handle = _lib.default_handle(local_rank)
  lib = handle.lib
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=MAX_CLUSTERS,
      refinement_options=sca.configs.icassp2018_refinement_options,
      laplacian_type=sca.LaplacianType.GraphCut, device=local_rank)
  cfg = clusterer.build_config()

  # per-rank utterance (different seed per rank: independent replicas)
  x, truth = blobs(N_SAMPLES, N_FEATURES, N_SPEAKERS, SEED + rank)
  labels = np.empty(N_SAMPLES, dtype=np.int64)
  diag = _lib.ScDiag()
  handle.check(lib.sc_set_embeddings(handle.raw, _lib.as_double_p(x), N_SAMPLES,
                                     N_FEATURES))

  def step():
    handle.check(lib.sc_run_resident(handle.raw, cfg, _lib.as_int64_p(labels), diag))

  def fence():
    handle.check(lib.sc_synchronize(handle.raw))
    if dist is not None:
      import torch
      dist.barrier()
      torch.cuda.synchronize()

  for _ in range(args.warmup):
    step()
  fence()
  stage_sum = np.zeros(len(_lib.STAGE_NAMES))
  passes = 0
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
    stage_sum += [diag.stage_ms[i] for i in range(len(_lib.STAGE_NAMES))]
    passes += diag.eig_matvec_passes
  fence()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    k = args.steps
    stage_ms = {name: float(stage_sum[i] / k) for i, name in enumerate(_lib.STAGE_NAMES)}
    # dominant kernel: the Diffuse SYRK-style fp64 MFMA GEMM.  Algorithmic flops of
    # the symmetric product = (upper-triangle tile pairs) * 2 * 128 * 128 * n.
    nt = (N_SAMPLES + GEMM_TILE - 1) // GEMM_TILE
    flops = nt * (nt + 1) // 2 * 2.0 * GEMM_TILE * GEMM_TILE * N_SAMPLES
    diffuse_s = stage_ms["diffuse"] * 1e-3
    achieved = flops / diffuse_s / 1e12 if diffuse_s > 0 else 0.0
    out = {
        "metric": "predict() calls/sec, n=8192 d=256 ICASSP2018 (GraphCut, eigengap "
                  "k in [2,20], cosine k-means)",
        "value": world * k / elapsed, "unit": "calls/s", "n_gpus": world, "steps": k,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / k,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "icassp2018_graphcut_n8192_d256_k8_max20",
                   "n_samples": N_SAMPLES, "n_features": N_FEATURES,
                   "speakers": N_SPEAKERS, "parallelism": "replicas x%d" % world},
        "roofline": {"bound": "mfma", "kernel": "k_gemm_nt<EpiNone,SYM> (Diffuse)",
                     "achieved": achieved, "peak": PEAK_F64_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_F64_MFMA_TFLOPS,
                     "traffic": None, "flops_per_launch": flops,
                     "avg_launch_ms": stage_ms["diffuse"]},
        "stage_ms": stage_ms,
        "eig": {"matvec_passes_per_call": passes / k, "block": int(diag.eig_block),
                "basis": int(diag.eig_basis), "cycles": int(diag.eig_cycles)},
        "parity": {"n_clusters": int(diag.n_clusters),
                   "ari_vs_truth": ari(labels, truth)},
    }
    gpath = os.path.join(ROOT, "tests", "golden", "e2e_n8192_lap4_max20.npz")
    if os.path.exists(gpath):
      g = np.load(gpath)
      w = diag.eigenvalue_array()[g["consumed_index"]]
      out["parity"]["ari_vs_reference_labels"] = ari(labels, g["labels"])
      out["parity"]["max_rel_err_consumed_eigenvalues"] = float(np.max(
          np.abs(w - g["consumed_eigenvalues"]) /
          np.maximum(np.abs(g["consumed_eigenvalues"]), 1e-12)))
      out["parity"]["reference_seconds_per_call_8vcpu"] = float(g["ref_seconds"])
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_diffuse.json")
    if os.path.exists(tpath):  # HBM bytes per Diffuse launch from a separate --pmc run
      t = json.load(open(tpath))
      out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
      out["roofline"]["traffic_source"] = t["source"]
    if world == 1 and not args.no_concurrent:
      out["concurrent_streams"] = concurrent_leg(sca, _lib, cfg, x, args.steps)
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline_leg(clusterer)
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
