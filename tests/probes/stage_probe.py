#!/usr/bin/env python
"""Run one stage in isolation (for rocprofv3 PMC passes):
   python tests/probes/stage_probe.py diffuse|affinity|predict [n] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralcluster_amd as sca  # noqa: E402
from spectralcluster_amd import refinement as rf  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "diffuse"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
if what == "diffuse":
  m = rng.random((n, n))
  for _ in range(reps):
    t = time.perf_counter()
    out = rf.Diffuse().refine(m)
    print("diffuse call (incl. H2D/D2H) %.1f ms" % (1e3 * (time.perf_counter() - t)))
elif what == "affinity":
  x = rng.standard_normal((n, 256))
  for _ in range(reps):
    sca.utils.compute_affinity_matrix(x)
elif what == "predict300":
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import _inputs as so
  x = so.blobs(n, 256, 4, n)
  c = sca.configs.icassp2018_clusterer
  for _ in range(reps):
    c.predict(x)
  print(c.last_diag.stage_times_ms())
else:
  x = rng.standard_normal((n, 256))
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                            refinement_options=sca.configs.icassp2018_refinement_options,
                            laplacian_type=sca.LaplacianType.GraphCut)
  for _ in range(reps):
    c.predict(x)
  print(c.last_diag.stage_times_ms())
