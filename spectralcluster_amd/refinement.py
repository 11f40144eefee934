"""Affinity refinement operations, executed on the MI355X.

Mirror of the reference interface `spectralcluster/refinement.py` (enums :11-36,
`AffinityRefinementOperation` :39-68, `RefinementOptions` :71-133, the six
operators :136-245).  Every `refine()` call runs the corresponding HIP kernel
through `sc_stage_refine` (ndarray in, ndarray out); `SpectralClusterer` does
not call these one by one but hands the whole sequence to the device pipeline.
"""

from __future__ import annotations

import abc
import dataclasses
import enum
import typing

import numpy as np

from spectralcluster_amd import _lib


class RefinementName(enum.Enum):
  """Names of the refinement operations (reference refinement.py:11-18)."""
  CropDiagonal = 1
  GaussianBlur = 2
  RowWiseThreshold = 3
  Symmetrize = 4
  Diffuse = 5
  RowWiseNormalize = 6


class ThresholdType(enum.Enum):
  """RowMax: clear values below row_max * p.  Percentile: clear the p*100 %
  smallest values of the row (reference refinement.py:21-27)."""
  RowMax = 1
  Percentile = 2


class SymmetrizeType(enum.Enum):
  """Max: max(A, A^T).  Average: (A + A^T) / 2 (reference refinement.py:30-36)."""
  Max = 1
  Average = 2


def gaussian_weights(sigma: float) -> np.ndarray:
  """The 1-D kernel scipy.ndimage.gaussian_filter builds for `sigma` (order 0,
  truncate 4.0): same NumPy expression as scipy 1.15.3 `_gaussian_kernel1d`, so
  the weights handed to the kernel are bit-identical to the reference's."""
  sd = float(sigma)
  radius = int(4.0 * sd + 0.5)
  grid = np.arange(-radius, radius + 1)
  phi = np.exp(-0.5 / (sd * sd) * grid ** 2)
  return phi / phi.sum()


def fill_config(cfg: _lib.ScConfig, *, sigma=1, p_percentile=0.95, multiplier=0.01,
                threshold_type=ThresholdType.RowMax, binarize=False,
                preserve_diagonal=False, symmetrize_type=SymmetrizeType.Max):
  """Write the refinement fields of an `sc_config`."""
  if not isinstance(threshold_type, ThresholdType):
    raise TypeError("thresholding_type must be a ThresholdType")
  if not isinstance(symmetrize_type, SymmetrizeType):
    raise ValueError("Unsupported symmetrize_type.")
  if float(sigma) > 1e-15:
    w = gaussian_weights(sigma)
    radius = (w.size - 1) // 2
    cfg.blur_radius = radius
    if radius > _lib.SC_MAX_BLUR_RADIUS:
      # sigma > 8: the weights do not fit `sc_config`; they travel to the handle on their own
      # (`_lib.sync_blur_weights`, called wherever this config meets a handle)
      cfg._blur_ext = np.ascontiguousarray(w, dtype=np.float64)
      # ... and the config names them by their central 65 entries (checked on every call)
      for i in range(2 * _lib.SC_MAX_BLUR_RADIUS + 1):
        cfg.blur_weights[i] = float(w[radius - _lib.SC_MAX_BLUR_RADIUS + i])
    else:
      for i, v in enumerate(w):
        cfg.blur_weights[i] = float(v)
  else:
    cfg.blur_radius = 0
    cfg.blur_weights[0] = 1.0
  cfg.p_percentile = float(p_percentile)
  cfg.soft_multiplier = float(multiplier)
  cfg.threshold_type = threshold_type.value
  cfg.binarize = int(bool(binarize))
  cfg.preserve_diagonal = int(bool(preserve_diagonal))
  cfg.symmetrize_type = symmetrize_type.value


class AffinityRefinementOperation(metaclass=abc.ABCMeta):
  """Base class of the operators (reference refinement.py:39-68)."""

  def check_input(self, affinity: np.ndarray):
    """ValueError unless `affinity` is a square 2-D matrix (:42-56)."""
    shape = affinity.shape
    if len(shape) != 2:
      raise ValueError("affinity must be 2-dimensional")
    if shape[0] != shape[1]:
      raise ValueError("affinity must be a square matrix")

  def _config(self) -> _lib.ScConfig:
    cfg = _lib.ScConfig()
    _lib.load().sc_config_default(cfg)
    return cfg

  _OP: typing.ClassVar[RefinementName]

  def refine(self, affinity: np.ndarray) -> np.ndarray:
    """Run this operator's kernel on the device; returns a new (n, n) array."""
    self.check_input(affinity)
    src = np.ascontiguousarray(affinity, dtype=np.float64)
    out = np.empty_like(src)
    handle = _lib.default_handle()
    cfg = self._config()
    _lib.sync_blur_weights(handle, cfg)
    handle.check(handle.lib.sc_stage_refine(
        handle.raw, self._OP.value, cfg, _lib.as_double_p(src), src.shape[0],
        _lib.as_double_p(out)))
    return out


class CropDiagonal(AffinityRefinementOperation):
  """diag_i <- max(0, max_{j != i} a_ij) (reference refinement.py:136-151)."""
  _OP = RefinementName.CropDiagonal


class GaussianBlur(AffinityRefinementOperation):
  """scipy.ndimage.gaussian_filter(a, sigma) (reference refinement.py:154-162)."""
  _OP = RefinementName.GaussianBlur

  def __init__(self, sigma: float = 1):
    self.sigma = sigma

  def _config(self):
    cfg = super()._config()
    fill_config(cfg, sigma=self.sigma)
    return cfg


class RowWiseThreshold(AffinityRefinementOperation):
  """Row-wise soft/hard thresholding (reference refinement.py:165-210)."""
  _OP = RefinementName.RowWiseThreshold

  def __init__(self, p_percentile: float = 0.95,
               thresholding_soft_multiplier: float = 0.01,
               thresholding_type: ThresholdType = ThresholdType.RowMax,
               thresholding_with_binarization: bool = False,
               thresholding_preserve_diagonal: bool = False):
    if not isinstance(thresholding_type, ThresholdType):
      raise TypeError("thresholding_type must be a ThresholdType")
    self.p_percentile = p_percentile
    self.multiplier = thresholding_soft_multiplier
    self.thresholding_type = thresholding_type
    self.thresholding_with_binarization = thresholding_with_binarization
    self.thresholding_preserve_diagonal = thresholding_preserve_diagonal

  def _config(self):
    cfg = super()._config()
    fill_config(cfg, p_percentile=self.p_percentile, multiplier=self.multiplier,
                threshold_type=self.thresholding_type,
                binarize=self.thresholding_with_binarization,
                preserve_diagonal=self.thresholding_preserve_diagonal)
    return cfg


class Symmetrize(AffinityRefinementOperation):
  """max(A, A^T) or (A + A^T) / 2 (reference refinement.py:213-226)."""
  _OP = RefinementName.Symmetrize

  def __init__(self, symmetrize_type: SymmetrizeType = SymmetrizeType.Max):
    self.symmetrize_type = symmetrize_type

  def _config(self):
    cfg = super()._config()
    fill_config(cfg, symmetrize_type=self.symmetrize_type)
    return cfg


class Diffuse(AffinityRefinementOperation):
  """A A^T on the fp64 MFMA GEMM (reference refinement.py:229-234)."""
  _OP = RefinementName.Diffuse


class RowWiseNormalize(AffinityRefinementOperation):
  """Each row divided by its max (reference refinement.py:237-245)."""
  _OP = RefinementName.RowWiseNormalize


@dataclasses.dataclass
class RefinementOptions:
  """Option bag for the refinement chain; same fields and defaults as the
  reference dataclass (refinement.py:71-100)."""
  gaussian_blur_sigma: int = 1
  p_percentile: float = 0.95
  thresholding_soft_multiplier: float = 0.01
  thresholding_type: ThresholdType = ThresholdType.RowMax
  thresholding_with_binarization: bool = False
  thresholding_preserve_diagonal: bool = False
  symmetrize_type: SymmetrizeType = SymmetrizeType.Max
  refinement_sequence: typing.Optional[typing.Sequence[RefinementName]] = None

  def get_refinement_operator(self, name: RefinementName) -> (
      AffinityRefinementOperation):
    """Operator object for `name` (reference refinement.py:102-133)."""
    builders = {
        RefinementName.CropDiagonal: CropDiagonal,
        RefinementName.GaussianBlur: lambda: GaussianBlur(self.gaussian_blur_sigma),
        RefinementName.RowWiseThreshold: lambda: RowWiseThreshold(
            self.p_percentile, self.thresholding_soft_multiplier,
            self.thresholding_type, self.thresholding_with_binarization,
            self.thresholding_preserve_diagonal),
        RefinementName.Symmetrize: lambda: Symmetrize(self.symmetrize_type),
        RefinementName.Diffuse: Diffuse,
        RefinementName.RowWiseNormalize: RowWiseNormalize,
    }
    if name not in builders:
      raise ValueError("Unknown refinement operation: {}".format(name))
    return builders[name]()

  def to_config(self, cfg: _lib.ScConfig):
    """Flatten into the refinement fields of an `sc_config`."""
    sequence = list(self.refinement_sequence or [])
    if len(sequence) > _lib.SC_MAX_OPS:
      raise _lib.UnsupportedOnDeviceError(
          "refinement_sequence longer than %d" % _lib.SC_MAX_OPS)
    for name in sequence:
      if not isinstance(name, RefinementName):
        raise ValueError("Unknown refinement operation: {}".format(name))
    cfg.n_ops = len(sequence)
    for i, name in enumerate(sequence):
      cfg.ops[i] = name.value
    fill_config(cfg, sigma=self.gaussian_blur_sigma, p_percentile=self.p_percentile,
                multiplier=self.thresholding_soft_multiplier,
                threshold_type=self.thresholding_type,
                binarize=self.thresholding_with_binarization,
                preserve_diagonal=self.thresholding_preserve_diagonal,
                symmetrize_type=self.symmetrize_type)
