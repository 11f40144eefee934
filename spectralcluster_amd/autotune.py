"""p_percentile auto-tuning (mirror of reference `spectralcluster/autotune.py`).

Host-side search logic only; each evaluated p_percentile is one device
`sc_eig_ncluster` call on the resident affinity.
"""

from __future__ import annotations

import enum
import math
import typing

import numpy as np

MIN_SEARCH_STEP = 1e-04


class AutoTuneProxy(enum.Enum):
  """Proxy of the DER that is minimised (reference autotune.py:10-23):
  (1 - p) / g (Park et al. 2019) or sqrt(1 - p) / g (Xia et al. 2022), g being
  the maximum eigengap."""
  PercentileOverNME = 1
  PercentileSqrtOverNME = 2


class AutoTune:
  """Grid search over p_percentile (reference autotune.py:26-132)."""

  def __init__(self, p_percentile_min: float = 0.60, p_percentile_max: float = 0.95,
               init_search_step: float = 0.01, search_level: int = 1,
               proxy: AutoTuneProxy = AutoTuneProxy.PercentileSqrtOverNME):
    if not isinstance(proxy, AutoTuneProxy):
      raise TypeError("proxy must be an instance of AutoTuneProxy")
    self.p_percentile_min = p_percentile_min
    self.p_percentile_max = p_percentile_max
    self.search_step = init_search_step
    self.search_level = search_level
    self.proxy = proxy

  def get_percentile_range(self) -> typing.Sequence[float]:
    """linspace(min, max, ceil((max - min) / step)) (reference :58-64)."""
    count = int(math.ceil(
        (self.p_percentile_max - self.p_percentile_min) / self.search_step))
    return list(np.linspace(self.p_percentile_min, self.p_percentile_max, count))

  def update_percentile_range(self, p_percentile_min: float, p_percentile_max: float,
                              search_step: float) -> typing.Sequence[float]:
    self.p_percentile_min = p_percentile_min
    self.p_percentile_max = p_percentile_max
    self.search_step = search_step
    return self.get_percentile_range()

  def ratio(self, p_percentile: float, max_delta_norm: float) -> float:
    """The proxy value for one evaluation (reference
    spectral_clusterer.py:281-286)."""
    if self.proxy == AutoTuneProxy.PercentileSqrtOverNME:
      return np.sqrt(1 - p_percentile) / max_delta_norm
    if self.proxy == AutoTuneProxy.PercentileOverNME:
      return (1 - p_percentile) / max_delta_norm
    raise ValueError("Unsupported value of AutoTuneProxy")

  def tune(self, p_percentile_to_ratio: typing.Callable,
           evaluate_many: typing.Optional[typing.Callable] = None):
    """Minimise the proxy (reference :76-132).

    `p_percentile_to_ratio(p) -> (ratio, eigenvectors, n_clusters)`.  The
    optional `evaluate_many(list_of_p) -> list of those triples` lets a caller
    evaluate one search level's grid in parallel (the entries of a level are
    independent; multi-GPU sharding plugs in here).  First strict minimum wins.
    Like the reference, the object's range/step are left narrowed afterwards.
    """
    grid = self.get_percentile_range()
    done = {}
    best = None
    for _ in range(self.search_level):
      lowest = np.inf
      todo = [(i, p) for i, p in enumerate(grid) if p not in done]
      if evaluate_many is not None:
        results = evaluate_many([p for _, p in todo])
      else:
        results = [p_percentile_to_ratio(p) for _, p in todo]
      for (index, p), (ratio, vectors, n_clusters) in zip(todo, results):
        done[p] = ratio
        if ratio < lowest:
          lowest = ratio
          best = (vectors, n_clusters, p, index)
      if not grid or len(grid) == 1 or self.search_step < MIN_SEARCH_STEP:
        break
      reach = max(2, len(grid) // 8)
      low = max(0, best[3] - reach)
      high = min(len(grid) - 1, best[3] + reach)
      grid = self.update_percentile_range(grid[low], grid[high],
                                          self.search_step / 2)
    return best[0], best[1], best[2]
