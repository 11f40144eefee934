"""Laplacian matrices (mirror of reference `spectralcluster/laplacian.py`)."""

from __future__ import annotations

import enum

import numpy as np

from spectralcluster_amd import _lib

EPS = 1e-10


class LaplacianType(enum.Enum):
  """Affinity: W itself.  Unnormalized: D - W.  RandomWalk: D^-1 (D - W).
  GraphCut: D^-1/2 (D - W) D^-1/2 (reference laplacian.py:9-21)."""
  Affinity = 1
  Unnormalized = 2
  RandomWalk = 3
  GraphCut = 4


def compute_laplacian(affinity: np.ndarray,
                      laplacian_type: LaplacianType = LaplacianType.GraphCut,
                      eps: float = EPS) -> np.ndarray:
  """Materialised Laplacian on the device (reference laplacian.py:24-60).

  The predict() pipeline never materialises it: there the D^-1/2 scaling is
  folded into the eigen-operator.  This entry point exists for parity tests and
  callers of the reference function.  `eps` is fixed at 1e-10 on the device.
  """
  if not isinstance(laplacian_type, LaplacianType):
    raise TypeError("laplacian_type must be a LaplacianType")
  if eps != EPS:
    raise _lib.UnsupportedOnDeviceError("eps is fixed at 1e-10 on the device path")
  src = np.ascontiguousarray(affinity, dtype=np.float64)
  if src.ndim != 2 or src.shape[0] != src.shape[1]:
    raise ValueError("affinity must be a square matrix")
  out = np.empty_like(src)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_stage_laplacian(
      handle.raw, laplacian_type.value, _lib.as_double_p(src), src.shape[0],
      _lib.as_double_p(out)), TypeError)
  return out
