#!/usr/bin/env python
"""Single predict() calls over a range of utterance sizes (d=256, ICASSP2018 preset): ms per
call and the stage timers.   python tests/probes/single_sizes.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

c = sca.configs.icassp2018_clusterer
for n in [int(v) for v in (sys.argv[1:] or ["500", "1000", "1650", "2300", "3000"])]:
  x = so.blobs(n, 256, 4, seed=n)
  c.predict(x)
  t = time.perf_counter()
  for _ in range(20):
    c.predict(x)
  ms = 1e3 * (time.perf_counter() - t) / 20
  st = c.last_diag.stage_times_ms()
  print("n=%d: %.3f ms/call  affinity %.3f refine %.3f diffuse %.3f eig %.3f kmeans %.3f" % (
      n, ms, st["affinity"], st["refine"], st["diffuse"], st["eig"], st["kmeans"]), flush=True)
