"""K-means on the spectral embedding (mirror of reference
`spectralcluster/custom_distance_kmeans.py`): sklearn's k-means++ seeds + the custom-distance
loop (cosine by default; euclidean, sqeuclidean, cityblock, chebyshev) inside one HIP kernel."""

from __future__ import annotations

import ctypes
import typing

import numpy as np

from spectralcluster_amd import _lib


def run_kmeans(spectral_embeddings: np.ndarray, n_clusters: int,
               custom_dist: typing.Union[str, typing.Callable],
               max_iter: int) -> np.ndarray:
  """k-means++ (sklearn `RandomState(0)` stream) + one Lloyd step for the seeds, then the
  reference's custom-distance loop (custom_distance_kmeans.py:39-51, 85-141).  A falsy
  `custom_dist` raises NotFittedError, as the reference does (it predicts with an unfitted
  sklearn KMeans, :33-36, :51)."""
  metric = _lib.kmeans_metric_code(custom_dist)
  e = np.ascontiguousarray(spectral_embeddings, dtype=np.float64)
  if e.ndim != 2:
    raise ValueError("spectral_embeddings must be 2-dimensional")
  n, k = e.shape
  if k != n_clusters:
    raise ValueError("spectral_embeddings must have n_clusters columns")
  labels = np.empty(n, dtype=np.int64)
  iters = ctypes.c_int(0)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_stage_kmeans_metric(
      handle.raw, _lib.as_double_p(e), n, int(n_clusters), int(max_iter), metric,
      _lib.as_int64_p(labels), None, ctypes.byref(iters)))
  return labels
