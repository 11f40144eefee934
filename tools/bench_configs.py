#!/usr/bin/env python
"""Timings of the other BASELINE.json configs on one GPU (not the driver's bench line):
   cfg2  n=2048 d=128 ICASSP2018, laplacian None
   cfg4  AutoTune 16-value p_percentile sweep, n=4096 d=256, GraphCut
   cfg5  512 utterances n in [300, 3000], d=256, ICASSP2018 preset (single-GPU share: all)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402  (input generator + ARI)
import spectralcluster_amd as sca  # noqa: E402

out = {}
opts = sca.configs.icassp2018_refinement_options

# cfg2
x = so.blobs(2048, 128, 4, 2048)
c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
c.predict(x)
t = time.perf_counter()
for _ in range(20):
  lab = c.predict(x)
out["cfg2_ms_per_call_incl_h2d"] = 1e3 * (time.perf_counter() - t) / 20
out["cfg2_stage_ms"] = c.last_diag.stage_times_ms()

# cfg4
x = so.blobs(4096, 256, 8, 4096)
def tuner():
  return sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95, init_search_step=0.025,
                      search_level=1)
c = sca.SpectralClusterer(min_clusters=2, max_clusters=20, refinement_options=sca.RefinementOptions(
    gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
    refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE), autotune=tuner(),
    laplacian_type=sca.LaplacianType.GraphCut)
c.predict(x)
c.autotune = tuner()
t = time.perf_counter()
lab = c.predict(x)
out["cfg4_autotune16_ms"] = 1e3 * (time.perf_counter() - t)
out["cfg4_best_p"] = float(c.refinement_options.p_percentile)

# cfg5
rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.configs.icassp2018_clusterer
for streams in (1, 2, 4, 8):
  c.predict_batch(utts[:16], streams=streams)
  t = time.perf_counter()
  labs = c.predict_batch(utts, streams=streams)
  dt = time.perf_counter() - t
  out["cfg5_batch512_s_streams%d" % streams] = dt
  out["cfg5_utterances_per_s_streams%d" % streams] = 512 / dt
for group in (4, 8, 16):  # the grouped batch: `group` utterances per launch, three lanes
  c.predict_batch(utts[:2 * group], group=group)
  t = time.perf_counter()
  labs = c.predict_batch(utts, group=group)
  dt = time.perf_counter() - t
  out["cfg5_batch512_s_group%d" % group] = dt
  out["cfg5_utterances_per_s_group%d" % group] = 512 / dt
truth_ok = 0
for i, (n, k) in enumerate(zip(ns, ks)):
  r = np.random.default_rng(i)
  r.standard_normal((int(k), 256))
  truth = np.sort(r.integers(0, int(k), int(n)))
  truth_ok += so.adjusted_rand_index(labs[i], truth) == 1.0
out["cfg5_ari1_vs_truth"] = int(truth_ok)
for n in (300, 1000, 3000):
  x = so.blobs(n, 256, 4, n)
  c.predict(x)
  t = time.perf_counter()
  for _ in range(10):
    c.predict(x)
  out["single_n%d_ms" % n] = 1e3 * (time.perf_counter() - t) / 10
  out["single_n%d_stage_ms" % n] = c.last_diag.stage_times_ms()
# E1 dense path: the reference's default max_clusters=None with a Laplacian
for n in (1000, 2048, 4096, 8192):
  x = so.blobs(n, 256 if n > 2048 else 128, 4, n)
  c = sca.SpectralClusterer(min_clusters=2, refinement_options=opts,
                            laplacian_type=sca.LaplacianType.GraphCut)
  c.predict(x)
  t = time.perf_counter()
  c.predict(x)
  out["dense_default_max_clusters_n%d_ms" % n] = 1e3 * (time.perf_counter() - t)
  out["dense_default_max_clusters_n%d_eig_ms" % n] = c.last_diag.stage_times_ms()["eig"]
print(json.dumps(out, indent=1))
