#!/usr/bin/env python
"""Randomised sweep over UNFRIENDLY inputs against the oracle (np.linalg.eig): odd sizes,
every Laplacian, several max_clusters, both eigengap rules.  A one-off stress of the round-3
solver paths (leading-vector Rayleigh-Ritz at bases 72..128, restarts, the dense landing pad
with a zero budget); not collected by pytest (the oracle's dgeev makes it minutes long).
   python tests/probes/hard_fuzz.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
LAP = {0: None, 2: sca.LaplacianType.Unnormalized, 3: sca.LaplacianType.RandomWalk,
       4: sca.LaplacianType.GraphCut}
bad = 0
t_all = time.perf_counter()
for case in range(cases):
  kind = so.HARD_KINDS[int(rng.integers(0, len(so.HARD_KINDS)))]
  n = int(rng.integers(300, 2600))
  d = int(rng.choice([32, 64, 256]))
  lap = int(rng.choice([0, 2, 3, 4]))
  maxc = int(rng.choice([7, 20, 40]))
  gap = int(rng.choice([so.EIGENGAP_RATIO, so.EIGENGAP_NORMALIZED_DIFF]))
  budget = int(rng.choice([0, 0, 0, -1]))  # sometimes: no restart budget -> dense landing pad
  p = float(rng.choice([0.95, 0.8, 0.6]))
  x = so.hard_inputs(kind, n, d, seed=1000 + case)
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=maxc, eigengap_type=gap,
                             p_percentile=p)
  dump = {}
  want = so.predict(x, cfg, dump)
  c = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc, laplacian_type=LAP[lap],
      eigengap_type=sca.EigenGapType.Ratio if gap == so.EIGENGAP_RATIO
      else sca.EigenGapType.NormalizedDiff,
      refinement_options=sca.RefinementOptions(
          gaussian_blur_sigma=1, p_percentile=p, thresholding_soft_multiplier=0.01,
          refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE))
  c.eig_max_cycles = budget
  try:
    got = c.predict(x)
  except Exception as e:  # pylint: disable=broad-except
    print("case %d RAISED %s: %s  (%s n=%d lap=%d maxc=%d gap=%d)" % (
        case, type(e).__name__, e, kind, n, lap, maxc, gap), flush=True)
    bad += 1
    continue
  dg = c.last_diag
  ref = np.real(dump["eigenvalues"])
  idx = so.consumed_eigen_indices(n, maxc, lap == 0, ref, 1e-2, gap)
  w = c.consumed_eigenvalues()
  err = np.max(np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max()))
  ari = so.adjusted_rand_index(got, want)
  k_ok = max(dg.n_clusters_raw, 2) == dump["n_clusters"]
  ok = err < 1e-5 and ari == 1.0 and k_ok
  bad += not ok
  print("case %2d %-7s n=%4d d=%3d lap=%d maxc=%2d gap=%d p=%.2f budget=%2d: path=%d passes=%3d "
        "cycles=%d err=%.1e k_ok=%d ari=%.3f %s" % (
            case, kind, n, d, lap, maxc, gap, p, budget, dg.eig_path, dg.eig_matvec_passes,
            dg.eig_cycles, err, k_ok, ari, "" if ok else "  <== MISMATCH"), flush=True)
print("%d cases, %d mismatches, %.0f s" % (cases, bad, time.perf_counter() - t_all))
