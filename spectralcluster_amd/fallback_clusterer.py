"""When not to run spectral clustering (mirror of reference
`spectralcluster/fallback_clusterer.py`; Wang et al., arXiv:2210.13690): too few embeddings
-> a fallback clusterer; `min_clusters == 1` -> a single-vs-multiple-clusters test first.

Every numeric piece runs on the device: the agglomerative fallback is the cosine
average-linkage AHC of `ahc.hip`, the naive fallback one persistent workgroup, the affinity
conditions are reductions over the resident affinity, and the GMM/BIC condition is an EM
over its upper triangle (`fallback.hip`).
"""

from __future__ import annotations

import dataclasses
import enum
import typing

import numpy as np

from spectralcluster_amd import _lib
from spectralcluster_amd import naive_clusterer
from spectralcluster_amd import utils


class SingleClusterCondition(enum.Enum):
  """How `min_clusters == 1` decides between one and several clusters
  (reference fallback_clusterer.py:24-44)."""
  AffinityGmmBic = 1       # BIC of a 1- vs 2-component GMM on the affinity values
  AllAffinity = 2          # every affinity above the threshold
  NeighborAffinity = 3     # every neighbouring affinity above the threshold
  AffinityStd = 4          # standard deviation of the affinities below the threshold
  FallbackClusterer = 5    # the fallback clusterer finds a single cluster


class FallbackClustererType(enum.Enum):
  """reference fallback_clusterer.py:47-54"""
  Agglomerative = 1
  Naive = 2


@dataclasses.dataclass
class FallbackOptions:
  """reference fallback_clusterer.py:58-92 (same fields and defaults)."""
  spectral_min_embeddings: int = 1
  single_cluster_condition: SingleClusterCondition = SingleClusterCondition.AffinityGmmBic
  single_cluster_affinity_threshold: float = 0.75
  single_cluster_affinity_diagonal_offset: int = 1
  fallback_clusterer_type: FallbackClustererType = FallbackClustererType.Naive
  agglomerative_threshold: float = 0.5
  naive_threshold: float = 0.5
  naive_adaptation_threshold: typing.Optional[float] = None


class _CosineAverageLinkage:
  """AgglomerativeClustering(n_clusters=None, metric="cosine", linkage="average",
  distance_threshold=t) of reference fallback_clusterer.py:108-113, on the device."""

  def __init__(self, distance_threshold: float):
    self.distance_threshold = distance_threshold

  def fit_predict(self, embeddings: np.ndarray) -> np.ndarray:
    return utils.cosine_agglomerative_clustering(
        embeddings, linkage="average", distance_threshold=self.distance_threshold)


class FallbackClusterer:
  """reference fallback_clusterer.py:95-124"""

  def __init__(self, options: FallbackOptions):
    self.options = options
    if options.fallback_clusterer_type == FallbackClustererType.Agglomerative:
      self.clusterer = _CosineAverageLinkage(options.agglomerative_threshold)
    elif options.fallback_clusterer_type == FallbackClustererType.Naive:
      self.clusterer = naive_clusterer.NaiveClusterer(
          threshold=options.naive_threshold,
          adaptation_threshold=options.naive_adaptation_threshold)
    # any other value: the reference builds the ValueError without raising it, and fails
    # later on the missing attribute; same here

  def predict(self, embeddings: np.ndarray) -> np.ndarray:
    return self.clusterer.fit_predict(embeddings)


def _resident_affinity_handle(affinity: np.ndarray, handle=None):
  """Upload `affinity` unless the caller says it is already resident on `handle`."""
  if handle is not None:
    return handle
  handle = _lib.default_handle()
  a = np.ascontiguousarray(affinity, dtype=np.float64)
  if a.ndim != 2 or a.shape[0] != a.shape[1]:
    raise ValueError("affinity must be a square matrix")
  handle.check(handle.lib.sc_set_affinity(handle.raw, _lib.as_double_p(a), a.shape[0]))
  return handle


def check_single_cluster(fallback_options: FallbackOptions,
                         embeddings: typing.Optional[np.ndarray],
                         affinity: typing.Optional[np.ndarray],
                         _resident_on=None) -> bool:
  """True when there is only a single cluster (reference fallback_clusterer.py:127-187;
  only called when min_clusters == 1).  `_resident_on` (internal) names the handle that
  already holds the affinity, so predict() does not ship it back and forth."""
  condition = fallback_options.single_cluster_condition
  threshold = fallback_options.single_cluster_affinity_threshold
  if condition in (SingleClusterCondition.AllAffinity, SingleClusterCondition.NeighborAffinity,
                   SingleClusterCondition.AffinityStd):
    handle = _resident_affinity_handle(affinity, _resident_on)
    stats = np.empty(4, dtype=np.float64)
    handle.check(handle.lib.sc_affinity_stats(handle.raw, _lib.as_double_p(stats)))
    if condition == SingleClusterCondition.AllAffinity:
      return bool(stats[0] > threshold)
    if condition == SingleClusterCondition.NeighborAffinity:
      return bool(stats[1] > threshold)
    return bool(stats[3] < threshold)
  if condition == SingleClusterCondition.AffinityGmmBic:
    handle = _resident_affinity_handle(affinity, _resident_on)
    bic = np.empty(2, dtype=np.float64)
    handle.check(handle.lib.sc_affinity_gmm_bic(
        handle.raw, int(fallback_options.single_cluster_affinity_diagonal_offset),
        _lib.as_double_p(bic[0:1]), _lib.as_double_p(bic[1:2])))
    return bool(bic[0] < bic[1])
  if condition == SingleClusterCondition.FallbackClusterer:
    labels = FallbackClusterer(fallback_options).predict(embeddings)
    return bool(np.unique(labels).size == 1)
  raise TypeError("Unsupported single_cluster_condition")
