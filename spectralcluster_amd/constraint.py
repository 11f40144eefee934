"""Pairwise-constraint operators (mirror of reference `spectralcluster/constraint.py`).

`AffinityIntegration` and `ConstraintPropagation` run on the device through
`sc_stage_constraint`; inside `SpectralClusterer.predict` the constraint matrix stays
resident and the same kernels run in the pipeline (`sc_set_constraint`).  The E2CP
inverse `(I - alpha * A_norm)^-1` is evaluated as a Neumann product of fp64 MFMA GEMMs
(see `csrc/constraint_api.hip: constraint_propagation`), which needs |alpha| < 1 and a
non-negative affinity; anything else raises `UnsupportedOnDeviceError`.
"""

from __future__ import annotations

import abc
import dataclasses
import enum
import typing

import numpy as np

from spectralcluster_amd import _lib

EPS = 1e-10


class ConstraintName(enum.Enum):
  """Constrained-clustering methods (reference constraint.py:10-16)."""
  AffinityIntegration = 1
  ConstraintPropagation = 2


class IntegrationType(enum.Enum):
  """How AffinityIntegration merges the two matrices (reference constraint.py:19-22)."""
  Max = 1
  Average = 2


def _device_adjust(affinity: np.ndarray, constraint_matrix: np.ndarray, name: ConstraintName,
                   integration_type: typing.Optional[IntegrationType],
                   alpha: float) -> np.ndarray:
  src = np.ascontiguousarray(affinity, dtype=np.float64)
  con = np.ascontiguousarray(constraint_matrix, dtype=np.float64)
  cfg = _lib.ScConfig()
  _lib.load().sc_config_default(cfg)
  cfg.constraint_name = name.value
  cfg.constraint_before_refinement = 1
  cfg.integration_type = integration_type.value if integration_type is not None else 0
  cfg.constraint_alpha = float(alpha)
  out = np.empty_like(src)
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_stage_constraint(
      handle.raw, cfg, _lib.as_double_p(src), _lib.as_double_p(con), src.shape[0],
      _lib.as_double_p(out)))
  return out


class ConstraintOperation(metaclass=abc.ABCMeta):
  """Base class of the two operators (reference constraint.py:51-92)."""

  def check_input(self, affinity: np.ndarray, constraint_matrix: np.ndarray):
    """Same checks and messages as reference constraint.py:54-76."""
    for what, m in (("affinity", affinity), ("constraint matrix", constraint_matrix)):
      if len(m.shape) != 2:
        raise ValueError("%s must be 2-dimensional" % what)
      if m.shape[0] != m.shape[1]:
        raise ValueError("%s must be a square matrix" % what)
    if affinity.shape != constraint_matrix.shape:
      raise ValueError("affinity and constraint matrix must have the same shape")

  @abc.abstractmethod
  def adjust_affinity(self, affinity: np.ndarray,
                      constraint_matrix: np.ndarray) -> np.ndarray:
    """Returns the adjusted (n, n) affinity."""


class AffinityIntegration(ConstraintOperation):
  """max(A, Q) or (A + Q) / 2 (reference constraint.py:95-118)."""

  def __init__(self, integration_type: IntegrationType = IntegrationType.Max):
    self.integration_type = integration_type

  def adjust_affinity(self, affinity, constraint_matrix):
    self.check_input(affinity, constraint_matrix)
    if not isinstance(self.integration_type, IntegrationType):
      raise ValueError("Unsupported integration type: {}".format(self.integration_type))
    return _device_adjust(affinity, constraint_matrix, ConstraintName.AffinityIntegration,
                          self.integration_type, 0.0)


class ConstraintPropagation(ConstraintOperation):
  """Exhaustive and efficient constraint propagation, E2CP (Lu & Ip, ECCV 2010;
  reference constraint.py:121-164): F = (1-a)^2 (I - a A_norm)^-1 Q (I - a A_norm)^-1,
  then A <- 1 - (1 - F)(1 - A) where F > 0 and (1 + F) A elsewhere."""

  def __init__(self, alpha: float = 0.6):
    self.alpha = alpha

  def adjust_affinity(self, affinity, constraint_matrix):
    self.check_input(affinity, constraint_matrix)
    return _device_adjust(affinity, constraint_matrix, ConstraintName.ConstraintPropagation,
                          None, self.alpha)


@dataclasses.dataclass
class ConstraintOptions:
  """Option bag handed to SpectralClusterer (reference constraint.py:25-48)."""
  constraint_name: ConstraintName
  # True: adjust the raw affinity (suggested for ConstraintPropagation);
  # False: adjust the refined matrix (suggested for AffinityIntegration).
  apply_before_refinement: bool
  integration_type: typing.Optional[IntegrationType] = None
  constraint_propagation_alpha: float = 0.6

  def __post_init__(self):
    if self.constraint_name == ConstraintName.AffinityIntegration:
      self.constraint_operator = AffinityIntegration(self.integration_type)
    elif self.constraint_name == ConstraintName.ConstraintPropagation:
      self.constraint_operator = ConstraintPropagation(self.constraint_propagation_alpha)

  def to_config(self, cfg: _lib.ScConfig) -> None:
    """Fill the constraint fields of an `sc_config`."""
    if not isinstance(self.constraint_name, ConstraintName):
      raise TypeError("constraint_name must be a ConstraintName")
    cfg.constraint_name = self.constraint_name.value
    cfg.constraint_before_refinement = int(bool(self.apply_before_refinement))
    if self.constraint_name == ConstraintName.AffinityIntegration:
      # read the live operator, like the reference does at call time
      kind = self.constraint_operator.integration_type
      if not isinstance(kind, IntegrationType):
        raise ValueError("Unsupported integration type: {}".format(kind))
      cfg.integration_type = kind.value
    else:
      cfg.constraint_alpha = float(self.constraint_operator.alpha)


class ConstraintMatrix:
  """Constraint matrix from speaker-turn confidences (reference constraint.py:167-207):
  adjacent segments without a turn must link (+1); a turn whose score exceeds
  `threshold` makes them cannot-link (-1).  Host-side: O(n) scalar work."""

  def __init__(self, speaker_turn_scores: typing.Sequence[float], threshold: float = 1):
    if any(score < 0 for score in speaker_turn_scores):
      raise ValueError("Speaker turn score must be larger or equal to 0.")
    self.speaker_turn_scores = speaker_turn_scores
    self.threshold = threshold

  def compute_diagonals(self) -> np.ndarray:
    scores = np.asarray(self.speaker_turn_scores, dtype=np.float64)
    n = len(scores)
    out = np.zeros((n, n))
    if n < 2:
      return out
    nxt = scores[1:]  # nxt[i]: turn score between segment i and i + 1
    band = np.where(nxt == 0, 1.0, np.where(nxt > self.threshold, -1.0, 0.0))
    idx = np.arange(n - 1)
    out[idx, idx + 1] = band
    out[idx + 1, idx] = band
    return out
