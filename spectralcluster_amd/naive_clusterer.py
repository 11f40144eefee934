"""Naive online clusterer (mirror of reference `spectralcluster/naive_clusterer.py`, the
algorithm of "Speaker Diarization with LSTM"): an embedding joins the centroid with the
largest cosine similarity if that similarity reaches `threshold` (and moves the centroid if it
also exceeds `adaptation_threshold`), otherwise it starts a new centroid.

The walk over the embeddings is sequential by definition; it runs as one persistent
workgroup on the device (`sc_naive_cluster`), the state (centroids, counts) travelling with
this object so that `predict_next` continues an earlier `predict`.
"""

from __future__ import annotations

import ctypes
import typing

import numpy as np

from spectralcluster_amd import _lib


class NaiveCentroid:
  """One centroid: running mean of its members (reference naive_clusterer.py:5-22).
  Host-side value object; the clusterer itself keeps its state in arrays."""

  def __init__(self, embedding: np.ndarray):
    self.embedding = embedding
    self.count = 1

  def merge(self, embedding: np.ndarray):
    self.embedding = (self.embedding * self.count + embedding) / (self.count + 1)
    self.count += 1

  def cosine(self, embedding: np.ndarray) -> float:
    return np.dot(self.embedding, embedding) / (
        np.linalg.norm(self.embedding) * np.linalg.norm(embedding))


class NaiveClusterer:
  """Online clustering by cosine similarity to running centroids
  (reference naive_clusterer.py:24-105)."""

  def __init__(self, threshold: float,
               adaptation_threshold: typing.Optional[float] = None):
    self.threshold = threshold
    if adaptation_threshold is None:
      self.adaptation_threshold = threshold
    elif adaptation_threshold < threshold:
      raise ValueError("adaptation_threshold cannot be smaller than threshold")
    else:
      self.adaptation_threshold = adaptation_threshold
    self.reset()

  def reset(self):
    self._centroids = np.zeros((0, 0), dtype=np.float64)
    self._counts = np.zeros(0, dtype=np.int32)

  @property
  def centroids(self) -> typing.List[NaiveCentroid]:
    """The state as the reference exposes it: a list of NaiveCentroid."""
    out = []
    for row, count in zip(self._centroids, self._counts):
      c = NaiveCentroid(row.copy())
      c.count = int(count)
      out.append(c)
    return out

  def predict(self, embeddings: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(embeddings, dtype=np.float64)
    if x.ndim != 2:
      raise ValueError("embeddings must be 2-dimensional")
    n, d = x.shape
    if n == 0:
      return np.zeros(0, dtype=np.int64)
    k0 = self._counts.shape[0]
    if k0 and self._centroids.shape[1] != d:
      raise ValueError("embedding dimension changed")
    capacity = k0 + n
    centroids = np.zeros((capacity, d), dtype=np.float64)
    counts = np.zeros(capacity, dtype=np.int32)
    if k0:
      centroids[:k0] = self._centroids
      counts[:k0] = self._counts
    found = ctypes.c_int32(k0)
    labels = np.empty(n, dtype=np.int64)
    handle = _lib.default_handle()
    handle.check(handle.lib.sc_naive_cluster(
        handle.raw, _lib.as_double_p(x), n, d, float(self.threshold),
        float(self.adaptation_threshold), _lib.as_double_p(centroids),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(found), capacity,
        _lib.as_int64_p(labels)))
    self._centroids = centroids[:found.value].copy()
    self._counts = counts[:found.value].copy()
    return labels

  def predict_next(self, embedding: np.ndarray) -> int:
    return int(self.predict(np.asarray(embedding, dtype=np.float64)[None, :])[0])

  def fit_predict(self, embeddings: np.ndarray) -> np.ndarray:
    """Same as predict(): this is an online clusterer."""
    return self.predict(embeddings)
