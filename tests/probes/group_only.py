#!/usr/bin/env python
"""Two identical passes of BASELINE config 5 through the grouped batch (for rocprofv3 traces):
   python tests/probes/group_only.py [group] [streams]   (streams > 0: the multi-stream form instead)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

group = int(sys.argv[1]) if len(sys.argv) > 1 else 16
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.configs.icassp2018_clusterer
pre = os.environ.get("GROUP_ONLY_PRE", "")  # what bench.py does before its grouped leg
if "8" in pre:
  c.predict_batch(utts, streams=8)
if "1" in pre:
  c.predict_batch(utts, streams=1)
if "s" in pre:
  c.predict_batch(utts[:64], streams=8)
for _ in range(int(os.environ.get("GROUP_ONLY_PASSES", "2"))):
  t = time.perf_counter()
  if streams > 0:
    c.predict_batch(utts, streams=streams)
  else:
    c.predict_batch(utts, group=group)
  dt = time.perf_counter() - t
  print("%.3f s  %.0f utterances/s" % (dt, 512 / dt), flush=True)
