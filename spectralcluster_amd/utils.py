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
  and max 64) extreme ones).  Anything else runs on the general solver (Hessenberg +
  complex QR for n <= 64, else block Arnoldi for `count` <= 32 extreme ones, default
  32); eigenvectors of complex pairs carry LAPACK's normalisation before `.real`.
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
