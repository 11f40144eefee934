"""One rank's LPT share of config 5 (world 2 / 8) through predict_batch(group=16), five times in a
row after one full-batch warm-up: is a slow share a warm-up effect or the share's own time?
  python tests/probes/share_repeat_probe.py            (SC_GROUP_EQUAL_COUNT=1 for round 5's groups)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402
from spectralcluster_amd import multigpu  # noqa: E402

rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.configs.icassp2018_clusterer
c.predict_batch(utts, group=16)
t = time.perf_counter()
c.predict_batch(utts, group=16)
print("full batch: %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
for world in (2, 8):
  shares = multigpu.lpt_assignment([int(n) for n in ns], world)
  for r in (0, world - 1):
    share = [utts[i] for i in shares[r]]
    times = []
    for _ in range(5):
      t = time.perf_counter()
      c.predict_batch(share, group=16)
      times.append(1e3 * (time.perf_counter() - t))
    print("world %d rank %d (%d utterances): %s ms" % (
        world, r, len(share), " ".join("%.1f" % v for v in times)), flush=True)
