"""bench.py's projection of config 5 on 8 GPUs, replayed with the group trace on: every rank's
LPT share ONCE after the full-batch warm-up (what project_shares does), several rounds; the trace
of a share that took more than 1.12 x the round's median is printed.
  SC_GROUP_TRACE=1 python tests/probes/share_outlier_probe.py [rounds]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402
from spectralcluster_amd import multigpu  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.configs.icassp2018_clusterer
for _ in range(3):
  c.predict_batch(utts, group=16)
shares = multigpu.lpt_assignment([int(n) for n in ns], 8)
for rnd in range(rounds):
  times, traces = [], []
  for r in range(8):
    share = [utts[i] for i in shares[r]]
    tmp = tempfile.TemporaryFile()
    sys.stderr.flush()
    saved = os.dup(2)
    os.dup2(tmp.fileno(), 2)
    t = time.perf_counter()
    c.predict_batch(share, group=16)
    dt = 1e3 * (time.perf_counter() - t)
    os.dup2(saved, 2)
    os.close(saved)
    tmp.seek(0)
    traces.append(tmp.read().decode(errors="replace"))
    times.append(dt)
  med = float(np.median(times))
  print("round %d: %s  (median %.2f)" % (rnd, " ".join("%.2f" % v for v in times), med), flush=True)
  for r, dt in enumerate(times):
    if dt > 1.12 * med:
      print("---- rank %d took %.2f ms:\n%s" % (r, dt, traces[r]), flush=True)
