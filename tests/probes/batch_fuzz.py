"""Randomised batches through the grouped batch (three lanes) against single predict() calls:
   python tests/probes/batch_fuzz.py [batches] [seed]
Sizes from 20 to 4500 (members below 129 and from 4096 take the single-call path inside the
batch), d in {16, 64, 256}, 2-9 speakers, noise levels up to heavily overlapping clusters,
both Laplacian settings, group widths 3..16; every utterance must get the labels and cluster
count of its own predict() call, and a second pass must repeat the first bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _inputs as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
rng = np.random.default_rng(seed)
bad = total = 0
t0 = time.perf_counter()
for b in range(batches):
  count = int(rng.integers(20, 140))
  d = int(rng.choice([16, 64, 256]))
  lap = [None, sca.LaplacianType.GraphCut][int(rng.integers(0, 2))]
  group = int(rng.integers(3, 17))
  ns = np.where(rng.random(count) < 0.08, rng.integers(20, 129, count),
                rng.integers(129, 3200, count))
  if rng.random() < 0.5:
    ns[int(rng.integers(0, count))] = int(rng.integers(4096, 4500))
  utts = []
  for i, n in enumerate(ns):
    k = int(rng.integers(2, 10))
    x = so.blobs(int(n), d, k, seed=int(rng.integers(1 << 30)))
    noise = float(rng.choice([0.0, 0.0, 0.5, 1.5]))
    if noise > 0:
      x = x + noise * np.random.default_rng(i).standard_normal(x.shape) / np.sqrt(d)
    utts.append(x)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=int(rng.integers(6, 21)),
                            refinement_options=sca.configs.icassp2018_refinement_options,
                            laplacian_type=lap)
  got = c.predict_batch(utts, group=group)
  diags = [(dg.n_clusters, dg.eigenvalue_array().copy()) for dg in c.last_batch_diags]
  again = c.predict_batch(utts, group=group)
  wrong = []
  for i, u in enumerate(utts):
    want = c.predict(u)
    ok = (np.array_equal(got[i], want) and np.array_equal(got[i], again[i]) and
          diags[i][0] == c.last_diag.n_clusters and
          np.array_equal(diags[i][1], c.last_batch_diags[i].eigenvalue_array()))
    if not ok:
      wrong.append((i, int(ns[i])))
  total += count
  bad += len(wrong)
  print("batch %d: %3d utterances, n %d..%d, d %d, group %d, laplacian %s: %d mismatches %s" %
        (b, count, ns.min(), ns.max(), d, group, "GraphCut" if lap else "None", len(wrong),
         wrong[:5]), flush=True)
print("%d utterances in %d batches, %d mismatches, %.1f s" %
      (total, batches, bad, time.perf_counter() - t0))
