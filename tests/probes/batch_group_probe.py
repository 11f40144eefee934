#!/usr/bin/env python
"""Config 5 (512 utterances) throughput of the grouped batch (one stream, one host thread) vs
the group width, beside the multi-stream form; labels of both must agree.
   python tests/probes/batch_group_probe.py [group widths ...]"""
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
ref = None
for streams in (1, 8):
  c.predict_batch(utts[:16], streams=streams)
  best = 1e9
  for _ in range(2):
    t = time.perf_counter()
    ref = c.predict_batch(utts, streams=streams)
    best = min(best, time.perf_counter() - t)
  print("streams %2d: %.3f s  %.0f utterances/s" % (streams, best, 512 / best), flush=True)
for group in [int(v) for v in (sys.argv[1:] or ["4", "8", "16"])]:
  c.predict_batch(utts[:2 * group], group=group)
  best = 1e9
  for _ in range(3):
    t = time.perf_counter()
    got = c.predict_batch(utts, group=group)
    best = min(best, time.perf_counter() - t)
  same = sum(int(np.array_equal(a, b)) for a, b in zip(got, ref))
  basis = np.bincount([d.eig_basis for d in c.last_batch_diags], minlength=9)
  print("group %2d: %.3f s  %.0f utterances/s   labels equal to the stream batch: %d/512   "
        "basis histogram %s" % (group, best, 512 / best, same,
                                {int(m): int(v) for m, v in enumerate(basis) if v}), flush=True)
