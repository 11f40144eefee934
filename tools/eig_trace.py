#!/usr/bin/env python
"""Per-stage timings + eigen-solver bookkeeping on the BASELINE-shaped inputs (blobs):
   SC_EIG_TRACE=1 SC_KMEANS_TRACE=1 python tools/eig_trace.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

opts = sca.configs.icassp2018_refinement_options
cases = [(300, 256, 4, None, 7), (1000, 256, 5, None, 7), (3000, 256, 6, None, 7),
         (2048, 128, 4, None, 7), (4096, 256, 8, sca.LaplacianType.GraphCut, 20),
         (8192, 256, 8, sca.LaplacianType.GraphCut, 20)]
for n, d, k, lap, maxc in cases:
  x = so.blobs(n, d, k, n if n != 8192 else 0)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                            laplacian_type=lap)
  c.predict(x)
  t = time.perf_counter()
  for _ in range(5):
    c.predict(x)
  ms = 1e3 * (time.perf_counter() - t) / 5
  dg = c.last_diag
  st = {a: round(b, 3) for a, b in dg.stage_times_ms().items() if b}
  print("n=%d lap=%s: %.3f ms/call  passes=%d basis=%d cycles=%d host_chain=%d k=%d  %s"
        % (n, lap, ms, dg.eig_matvec_passes, dg.eig_basis, dg.eig_cycles, dg.eig_host_chain,
           dg.n_clusters, st), flush=True)
