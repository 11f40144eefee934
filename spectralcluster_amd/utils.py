"""Utility functions (mirror of reference `spectralcluster/utils.py`)."""

from __future__ import annotations

import ctypes
import enum
import typing

import numpy as np

from spectralcluster_amd import _lib

EPS = 1e-10


class EigenGapType(enum.Enum):
  """Ratio of, or max-normalised difference between, consecutive eigenvalues
  (reference utils.py:10-17)."""
  Ratio = 1
  NormalizedDiff = 2


def compute_affinity_matrix(embeddings: np.ndarray) -> np.ndarray:
  """(cos(x_i, x_j) + 1) / 2 on the fp64 MFMA GEMM (reference utils.py:20-41)."""
  x = np.ascontiguousarray(embeddings, dtype=np.float64)
  if x.ndim != 2:
    raise ValueError("embeddings must be 2-dimensional")
  n = x.shape[0]
  out = np.empty((n, n), dtype=np.float64)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_stage_affinity(
      handle.raw, _lib.as_double_p(x), n, x.shape[1], _lib.as_double_p(out)))
  return out


def compute_sorted_eigenvectors(
    input_matrix: np.ndarray, descend: bool = True,
    count: typing.Optional[int] = None) -> typing.Tuple[np.ndarray, np.ndarray]:
  """Sorted eigenpairs (reference utils.py:44-71): `np.linalg.eig`, real parts,
  argsort by eigenvalue.

  A symmetric input runs on the symmetric solver (dense Jacobi for n <= 128 -- every
  eigenpair, as the reference returns -- else block Lanczos for the `count` (default
  extreme ones; more than 64 come from the dense tridiagonal path).  Anything else runs on
  the general solver (Hessenberg + complex QR for n <= 64, else block Arnoldi for `count` <= 64
  extreme ones, default 32; more than 64 -- up to all n -- take the dense Hessenberg route:
  reduction on the device, QR iteration + inverse iteration on the host); eigenvectors of
  complex pairs carry LAPACK's normalisation before `.real`.
  """
  m = np.ascontiguousarray(input_matrix, dtype=np.float64)
  if m.ndim != 2 or m.shape[0] != m.shape[1]:
    raise ValueError("input_matrix must be square")
  n = m.shape[0]
  scale = float(np.max(np.abs(m))) if m.size else 0.0
  symmetric = np.allclose(m, m.T, rtol=0.0, atol=1e-12 * max(scale, 1e-300))
  if count is None:
    if symmetric:
      count = n if n <= 128 else 64
    else:
      count = n if n <= 64 else 32
  values = np.empty(count, dtype=np.float64)
  vectors = np.empty((n, count), dtype=np.float64)
  handle = _lib.default_handle()
  entry = handle.lib.sc_stage_sym_eig if symmetric else handle.lib.sc_stage_eig
  handle.check(entry(
      handle.raw, _lib.as_double_p(m), n, count, int(bool(descend)),
      _lib.as_double_p(values), _lib.as_double_p(vectors), None))
  return values, vectors


def compute_number_of_clusters(eigenvalues: np.ndarray,
                               max_clusters: typing.Optional[int] = None,
                               stop_eigenvalue: float = 1e-2,
                               eigengap_type: EigenGapType = EigenGapType.Ratio,
                               descend: bool = True,
                               eps: float = EPS) -> typing.Tuple[int, float]:
  """Maximum-eigengap cluster count (reference utils.py:74-130); the scalar loop
  runs in the native library (`sc_eigengap`)."""
  if not isinstance(eigengap_type, EigenGapType):
    raise TypeError("eigengap_type must be a EigenGapType")
  if eps != EPS:
    raise _lib.UnsupportedOnDeviceError("eps is fixed at 1e-10 in the native library")
  w = np.ascontiguousarray(eigenvalues, dtype=np.float64)
  k = ctypes.c_int(0)
  delta = ctypes.c_double(0.0)
  rc = _lib.load().sc_eigengap(_lib.as_double_p(w), w.size, int(max_clusters or 0),
                               float(stop_eigenvalue), eigengap_type.value,
                               int(bool(descend)), ctypes.byref(k),
                               ctypes.byref(delta))
  if rc != _lib.SC_OK:
    raise ValueError("Unsupported eigengap_type")
  return k.value, delta.value


def enforce_ordered_labels(labels: np.ndarray) -> np.ndarray:
  """Relabel so that labels appear in increasing order of first occurrence
  (reference utils.py:133-156)."""
  out = np.empty_like(labels)
  order = {}
  for pos, value in enumerate(labels.tolist()):
    out[pos] = order.setdefault(value, len(order))
  return out


def get_cluster_centroids(embeddings: np.ndarray, labels: np.ndarray) -> np.ndarray:
  """Mean embedding of every cluster, (max(labels) + 1, n_features) (reference
  utils.py:159-176); the segmented mean runs on the device and adds the members in index
  order, as `np.mean(axis=0)` does."""
  x = np.ascontiguousarray(embeddings, dtype=np.float64)
  lab = np.ascontiguousarray(labels, dtype=np.int64)
  if x.ndim != 2 or lab.shape != (x.shape[0],):
    raise ValueError("embeddings must be (n_samples, n_features), labels (n_samples,)")
  k = int(lab.max()) + 1
  out = np.empty((k, x.shape[1]), dtype=np.float64)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_cluster_centroids(
      handle.raw, _lib.as_double_p(x), x.shape[0], x.shape[1], _lib.as_int64_p(lab), k,
      _lib.as_double_p(out)))
  return out


def chain_labels(pre_labels: typing.Optional[np.ndarray],
                 main_labels: np.ndarray) -> np.ndarray:
  """Labels of the samples given the labels of their pre-clusters (reference
  utils.py:179-206).  Like the reference the result is float64 (it fills `np.zeros`)."""
  if pre_labels is None:
    return main_labels
  count = int(max(pre_labels) + 1)
  if count != main_labels.shape[0]:
    raise ValueError("pre_labels has {} values while main_labels has {} rows.".format(
        count, main_labels.shape[0]))
  return np.asarray(main_labels, dtype=np.float64)[np.asarray(pre_labels, dtype=np.int64)]


LINKAGE_CODES = {"complete": 1, "average": 2}


def cosine_agglomerative_clustering(embeddings: np.ndarray,
                                    n_clusters: typing.Optional[int] = None,
                                    linkage: str = "complete",
                                    distance_threshold: typing.Optional[float] = None
                                    ) -> np.ndarray:
  """`sklearn.cluster.AgglomerativeClustering(n_clusters=..., metric="cosine",
  linkage=..., distance_threshold=...).fit_predict(embeddings)` on the device -- the
  pre-clusterer of the reference (spectral_clusterer.py:170-199,
  multi_stage_clusterer.py:109-112) and its agglomerative fallback
  (fallback_clusterer.py:108-113).  Labels are numbered exactly as sklearn numbers them
  (its heap-ordered tree cut); the downstream GaussianBlur depends on that order."""
  if linkage not in LINKAGE_CODES:
    raise _lib.UnsupportedOnDeviceError("linkage must be 'complete' or 'average'")
  if (n_clusters is None) == (distance_threshold is None):
    raise ValueError("Exactly one of n_clusters and distance_threshold has to be set, "
                     "and the other needs to be None.")
  x = np.ascontiguousarray(embeddings, dtype=np.float64)
  if x.ndim != 2:
    raise ValueError("embeddings must be 2-dimensional")
  labels = np.empty(x.shape[0], dtype=np.int64)
  found = ctypes.c_int(0)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_ahc(
      handle.raw, _lib.as_double_p(x), x.shape[0], x.shape[1], LINKAGE_CODES[linkage],
      int(n_clusters or 0), float(distance_threshold or 0.0), _lib.as_int64_p(labels),
      ctypes.byref(found)))
  return labels
