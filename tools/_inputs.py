"""Synthetic inputs and the label score the probes print (SURVEY.md section 8d generator).
Kept apart from oracle/: the probes time the product path and only need inputs."""
import numpy as np


def blobs(n, d, k, seed, noise=0.3, with_labels=False):
  """k Gaussian blobs in d dimensions, samples sorted by blob."""
  rng = np.random.default_rng(seed)
  centers = rng.standard_normal((k, d))
  lab = np.sort(rng.integers(0, k, n))
  x = np.ascontiguousarray(centers[lab] + noise * rng.standard_normal((n, d)))
  return (x, lab) if with_labels else x


def adjusted_rand_index(a, b):
  """ARI (Hubert & Arabie 1985) from the contingency table."""
  a = np.asarray(a).ravel()
  b = np.asarray(b).ravel()
  _, ai = np.unique(a, return_inverse=True)
  _, bi = np.unique(b, return_inverse=True)
  table = np.zeros((ai.max() + 1, bi.max() + 1), dtype=np.int64)
  np.add.at(table, (ai, bi), 1)
  pairs = lambda t: int((t * (t - 1) // 2).sum())
  both, rows, cols = pairs(table), pairs(table.sum(axis=1)), pairs(table.sum(axis=0))
  total = a.size * (a.size - 1) // 2
  expected = rows * cols / total if total else 0.0
  top = 0.5 * (rows + cols)
  return 1.0 if top == expected else float((both - expected) / (top - expected))
