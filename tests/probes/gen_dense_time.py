"""Timing of the dense general route (eig_path 7) on the round-5 reference goldens:
python tests/probes/gen_dense_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

for name in ("general_dense_n300_lap4", "general_dense_n1000_lap4", "general_dense_n500_max80",
             "general_dense_n300_lap0"):
  g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  ttype = sca.ThresholdType.Percentile if int(g["percentile"]) else sca.ThresholdType.RowMax
  opts = sca.RefinementOptions(
      p_percentile=float(g["p_percentile"]), thresholding_soft_multiplier=0.01,
      thresholding_type=ttype, refinement_sequence=[sca.RefinementName.RowWiseThreshold])
  c = sca.SpectralClusterer(min_clusters=int(g["min_clusters"]),
                            max_clusters=None if maxc < 0 else maxc, refinement_options=opts,
                            laplacian_type={0: None, 4: sca.LaplacianType.GraphCut}[lap])
  c.predict(x)
  t0 = time.perf_counter()
  c.predict(x)
  dt = time.perf_counter() - t0
  dg = c.last_diag
  print("%s: predict %.1f ms (eig stage %.1f ms), eig_path %d, reference %.2f s, ARI %.1f" % (
      name, 1e3 * dt, dg.stage_ms[4], dg.eig_path, float(g["ref_seconds"]),
      so.adjusted_rand_index(c.predict(x), g["labels"])), flush=True)
# larger: a thresholded affinity at n = 2000 / 4000, all eigenvalues against numpy
for n in (2000, 3000):
  x = so.blobs(n, 64, 6, seed=n)
  # (Percentile cut: a RowMax cut of an affinity whose diagonal is 1 is the same for every row
  #  and leaves the matrix symmetric -- it would take the symmetric dense path)
  a = so.row_wise_threshold(so.affinity(x), p_percentile=0.9,
                            threshold_type=so.THRESHOLD_PERCENTILE)
  assert not np.allclose(a, a.T)
  t0 = time.perf_counter()
  w, _ = sca.utils.compute_sorted_eigenvectors(a, descend=True, count=80)
  dt = time.perf_counter() - t0
  t0 = time.perf_counter()
  ev = np.linalg.eigvals(a)
  dn = time.perf_counter() - t0
  ev = np.sort(ev.real)[::-1]
  print("n=%d: device+host %.2f s, numpy eigvals %.2f s, max rel err of the 80 leading values %.1e" % (
      n, dt, dn, np.max(np.abs(w - ev[:80]) / np.abs(ev[:80]).max())), flush=True)
