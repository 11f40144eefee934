#!/usr/bin/env python
"""Dense eigen path timings (the reference's default max_clusters=None with a Laplacian =
tridiagonalisation + bisection + Lanczos vectors; and the full landing pad, path 6, forced
with SC_EIG_FORCE_DENSE=1 in the environment):  python tests/probes/dense_time.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1000, 2048, 4096, 8192]
opts = sca.configs.icassp2018_refinement_options
for n in sizes:
  x = so.blobs(n, 256 if n > 2048 else 128, 4, n)
  c = sca.SpectralClusterer(min_clusters=2, refinement_options=opts,
                            laplacian_type=sca.LaplacianType.GraphCut)
  c.predict(x)
  t = time.perf_counter()
  labels = c.predict(x)
  ms = 1e3 * (time.perf_counter() - t)
  dg = c.last_diag
  w = c.consumed_eigenvalues()
  line = "n=%d: %.1f ms per call, eig %.1f ms, path %d, %d eigenvalues" % (
      n, ms, dg.stage_times_ms()["eig"], dg.eig_path, w.size)
  if n <= 2048:  # accuracy against the oracle's dense LAPACK spectrum
    cfg = so.icassp2018_config(laplacian_type=so.LAPLACIAN_GRAPH_CUT, max_clusters=None)
    dump = {}
    want = so.predict(x, cfg, dump)
    ref = np.real(dump["eigenvalues"])
    line += ", max |w - ref| = %.2e, ARI %.3f" % (np.abs(w - ref).max(),
                                                  so.adjusted_rand_index(labels, want))
  print(line, flush=True)
