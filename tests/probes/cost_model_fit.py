#!/usr/bin/env python
"""Calibration of multigpu.cost_model: seconds per utterance of the GROUPED batch
(predict_batch(group=16), the execution the LPT partition schedules) as a function of n,
fitted as a + b n^2 + c n^3 (least squares on relative error).
   python tests/probes/cost_model_fit.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralcluster_amd as sca  # noqa: E402
from bench import blobs  # noqa: E402

c = sca.configs.icassp2018_clusterer
sizes = [300, 450, 650, 900, 1200, 1400, 1535, 1536, 1700, 2000, 2250, 2500, 2750, 3000]
per = []
for n in sizes:
  xs = [blobs(n, 256, 2 + i % 6, seed=1000 + i)[0] for i in range(64)]
  c.predict_batch(xs, group=16)
  best = 1e9
  for _ in range(3):
    t0 = time.perf_counter()
    c.predict_batch(xs, group=16)
    best = min(best, time.perf_counter() - t0)
  per.append(best / 64)
  print("n=%d: %.1f us per utterance (64 utterances, 4 groups)" % (n, 1e6 * per[-1]), flush=True)


def fit(ns, ts, what):
  n = np.array(ns, dtype=np.float64)
  t = np.array(ts) * 1e6
  A = np.stack([np.ones_like(n), n ** 2, n ** 3], axis=1) / t[:, None]  # relative error
  import scipy.optimize
  coef, _ = scipy.optimize.nnls(A, np.ones_like(t))
  f = coef[0] + coef[1] * n ** 2 + coef[2] * n ** 3
  print("%s: fit us = %.4g + %.4g n^2 + %.4g n^3, max rel err %.3f" % ((what,) + tuple(coef) + (np.max(np.abs(f - t) / t),)))
  print("COEF", what, repr(list(coef)))


# (members of n >= 1536 take the matrix-free Diffuse inside the grouped front: two branches)
lo = [(n, t) for n, t in zip(sizes, per) if 512 <= n < 1536]
hi = [(n, t) for n, t in zip(sizes, per) if n >= 1536]
fit([n for n, _ in lo], [t for _, t in lo], "512 <= n < 1536")
fit([n for n, _ in hi], [t for _, t in hi], "n >= 1536")
