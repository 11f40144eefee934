"""Multi-stage streaming clustering (mirror of reference
`spectralcluster/multi_stage_clusterer.py`; Wang et al., "Highly efficient real-time
streaming and fully on-device speaker diarization with multi-stage clustering",
arXiv:2210.13690).

Stage by size of the stream: fewer than L embeddings -> the agglomerative fallback; up to U1
-> the main spectral clusterer; beyond U1 -> a complete-linkage cosine AHC pre-clusterer
compresses the cache to U1 centroids which the main clusterer then clusters; at U2 cached
rows the cache is replaced by those centroids ("dynamic compression").  Every clustering
step runs on the device (AHC: `ahc.hip`; spectral: the hot path); this class is the host
bookkeeping around them, plus the label de-flickering.
"""

from __future__ import annotations

import enum

import numpy as np

from spectralcluster_amd import fallback_clusterer
from spectralcluster_amd import spectral_clusterer
from spectralcluster_amd import utils


class Deflicker(enum.Enum):
  """How the streaming output labels are kept stable (reference :19-28)."""
  NoDeflicker = 1
  OrderBased = 2    # enforce order-based labels
  Hungarian = 3     # match the previous output by an assignment problem


def linear_sum_assignment(cost: np.ndarray, maximize: bool = False):
  """Rectangular linear sum assignment, shortest augmenting paths (D. F. Crouse, "On
  implementing 2D rectangular assignment algorithms", IEEE T-AES 2016) -- the algorithm
  behind `scipy.optimize.linear_sum_assignment`, which the reference calls (:51).  Host
  code: the matrices here are (speakers x speakers).  Returns (row_ind, col_ind), rows
  ascending."""
  c = np.array(cost, dtype=np.float64)
  if c.ndim != 2:
    raise ValueError("expected a matrix (2-D array)")
  if maximize:
    c = -c
  transposed = c.shape[1] < c.shape[0]
  if transposed:
    c = c.T.copy()
  nr, nc = c.shape
  if nr == 0:
    return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
  c = c - c.min()                       # non-negative costs for the dual updates
  u = np.zeros(nr)
  v = np.zeros(nc)
  col4row = -np.ones(nr, dtype=np.int64)
  row4col = -np.ones(nc, dtype=np.int64)
  for cur_row in range(nr):
    shortest = np.full(nc, np.inf)
    path = -np.ones(nc, dtype=np.int64)
    in_rows = np.zeros(nr, dtype=bool)
    in_cols = np.zeros(nc, dtype=bool)
    remaining = list(range(nc - 1, -1, -1))   # columns still open, scanned in this order
    min_val = 0.0
    i = cur_row
    sink = -1
    while sink == -1:
      index = -1
      lowest = np.inf
      in_rows[i] = True
      for it, j in enumerate(remaining):
        reduced = min_val + c[i, j] - u[i] - v[j]
        if reduced < shortest[j]:
          path[j] = i
          shortest[j] = reduced
        # among equally short paths prefer an unassigned column: it ends the search
        if shortest[j] < lowest or (shortest[j] == lowest and row4col[j] == -1):
          lowest = shortest[j]
          index = it
      min_val = lowest
      if not np.isfinite(min_val):
        raise ValueError("cost matrix is infeasible")
      j = remaining[index]
      if row4col[j] == -1:
        sink = j
      else:
        i = row4col[j]
      in_cols[j] = True
      remaining[index] = remaining[-1]
      remaining.pop()
    u[cur_row] += min_val
    for r in range(nr):
      if in_rows[r] and r != cur_row:
        u[r] += min_val - shortest[col4row[r]]
    for col in range(nc):
      if in_cols[col]:
        v[col] -= min_val - shortest[col]
    j = sink
    while True:                          # augment along the path back to cur_row
      r = path[j]
      row4col[j] = r
      col4row[r], j = j, col4row[r]
      if r == cur_row:
        break
  if transposed:
    order = np.argsort(col4row)
    return col4row[order], order.astype(np.int64)
  return np.arange(nr, dtype=np.int64), col4row


def match_labels(current: np.ndarray, previous: np.ndarray) -> np.ndarray:
  """Rename the labels of `current` (one element longer) so that they agree with
  `previous` as much as possible (reference :31-63)."""
  current = utils.enforce_ordered_labels(np.asarray(current)).astype(np.int32)
  previous = np.asarray(previous).astype(np.int32)
  cropped = current[:-1]
  if cropped.shape != previous.shape:
    raise ValueError("current must have one more element than previous .")
  num_current = max(cropped) + 1
  num_previous = max(max(previous) + 1, num_current)
  # overlap[i, j]: how often label i now coincides with label j before
  overlap = np.zeros((num_current, num_previous), dtype=np.int32)
  np.add.at(overlap, (cropped, previous), 1)
  rows, cols = linear_sum_assignment(overlap, maximize=True)
  renamed = dict(zip(rows.tolist(), cols.tolist()))
  out = current.copy()
  for label in range(max(current) + 1):
    if label in renamed:
      out[current == label] = renamed[label]
  return out


class MultiStageClusterer:
  """reference multi_stage_clusterer.py:66-180"""

  def __init__(self,
               main_clusterer: spectral_clusterer.SpectralClusterer,
               fallback_threshold: float = 0.5,
               L: int = 50,
               U1: int = 100,
               U2: int = 600,
               deflicker: Deflicker = Deflicker.NoDeflicker):
    self.deflicker = deflicker
    self.main = main_clusterer
    if self.main.max_spectral_size:
      raise ValueError(
          "Do not set max_spectral_size for SpectralClusterer when"
          "using MultiStageClusterer.")
    options = self.main.fallback_options
    options.spectral_min_embeddings = L            # lower bound of the main clusterer
    self.U1 = U1                                   # upper bound of the main clusterer
    self.U2 = U2                                   # upper bound of the pre-clusterer
    options.agglomerative_threshold = fallback_threshold
    options.single_cluster_condition = (
        fallback_clusterer.SingleClusterCondition.FallbackClusterer)
    options.fallback_clusterer_type = (
        fallback_clusterer.FallbackClustererType.Agglomerative)
    self.cache = None                  # all cached embeddings / centroids
    self.num_embeddings = 0
    self.compression_labels = None     # original embedding -> compressed centroid
    self.previous_output = None

  def _pre_cluster(self, rows: np.ndarray) -> np.ndarray:
    """AgglomerativeClustering(n_clusters=U1, metric="cosine", linkage="complete")."""
    return utils.cosine_agglomerative_clustering(rows, n_clusters=self.U1,
                                                 linkage="complete")

  def streaming_predict(self, embedding: np.ndarray) -> np.ndarray:
    """Label the next embedding and re-label all earlier ones (reference :128-180)."""
    self.num_embeddings += 1
    if self.num_embeddings == 1:
      self.cache = embedding
      labels = np.array([0])
      self.previous_output = labels
      return labels
    self.cache = np.vstack([self.cache, embedding])

    if self.num_embeddings <= self.U1:             # fallback or main clusterer only
      labels = self.main.predict(self.cache)
      self.previous_output = labels
      return labels

    if self.compression_labels is not None:
      self.compression_labels = np.append(self.compression_labels,
                                          max(self.compression_labels) + 1)
    pre_labels = self._pre_cluster(self.cache)
    pre_centroids = utils.get_cluster_centroids(self.cache, pre_labels)
    main_labels = self.main.predict(pre_centroids)
    labels = utils.chain_labels(self.compression_labels,
                                utils.chain_labels(pre_labels, main_labels))

    if self.cache.shape[0] == self.U2:             # dynamic compression
      self.cache = pre_centroids
      self.compression_labels = utils.chain_labels(self.compression_labels, pre_labels)

    if self.deflicker == Deflicker.OrderBased:
      labels = utils.enforce_ordered_labels(labels)
    elif self.deflicker == Deflicker.Hungarian:
      labels = match_labels(labels, self.previous_output)
    self.previous_output = labels
    return labels
