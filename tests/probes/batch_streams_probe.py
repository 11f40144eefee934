#!/usr/bin/env python
"""Config 5 (512 utterances) throughput vs the number of streams of sc_predict_batch_streams."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.configs.icassp2018_clusterer
for streams in [int(v) for v in (sys.argv[1:] or ["1", "4", "8", "12", "16", "24", "32"])]:
  c.predict_batch(utts[:2 * streams], streams=streams)
  best = 1e9
  for _ in range(2):
    t = time.perf_counter()
    c.predict_batch(utts, streams=streams)
    best = min(best, time.perf_counter() - t)
  print("streams %2d: %.3f s  %.0f utterances/s" % (streams, best, 512 / best), flush=True)
