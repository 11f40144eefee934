"""Partitioning of independent predict() calls over the GPUs of one node.

The hot path has no exchange step inside a predict(): a
problem of <= 537 MB per matrix fits one MI355X and its stages are sequentially
dependent.  What shards are the *units* around it:

  * batched utterances  -> `predict_batch_sharded`  (LPT by an n^3 cost model)
  * AutoTune p sweeps   -> `autotune_sharded`       (p-grid round-robin)

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Collectives are the bare minimum: broadcast of
inputs from rank 0, all_gather of results (labels / three scalars per p).  The
compute itself is a callable, so the plumbing is testable without a GPU.
"""

from __future__ import annotations

import typing

import numpy as np


def cost_model(n: int) -> float:
  """Relative cost of one predict(): the n^3 Diffuse GEMM dominates, the O(n^2)
  row ops / eigen passes matter for small n."""
  return float(n) ** 3 + 64.0 * float(n) ** 2


def lpt_assignment(sizes: typing.Sequence[int], world: int) -> typing.List[typing.List[int]]:
  """Longest-processing-time-first: sort by cost descending, give each item to
  the least-loaded rank.  Returns, per rank, the item indices it owns (in the
  order it will run them).  Deterministic: ties go to the lowest rank."""
  order = sorted(range(len(sizes)), key=lambda i: (-cost_model(sizes[i]), i))
  load = [0.0] * world
  owned = [[] for _ in range(world)]
  for i in order:
    r = min(range(world), key=lambda q: (load[q], q))
    owned[r].append(i)
    load[r] += cost_model(sizes[i])
  return owned


def _dist():
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    raise RuntimeError("torch.distributed is not initialised")
  return dist


def _device_for_backend(dist):
  import torch
  if dist.get_backend() == "nccl":
    return torch.device("cuda", torch.cuda.current_device())
  return torch.device("cpu")


def broadcast_array(arr: typing.Optional[np.ndarray], src: int = 0) -> np.ndarray:
  """Broadcast a float64/int64 ndarray from `src` (shape first, then data)."""
  import torch
  dist = _dist()
  dev = _device_for_backend(dist)
  rank = dist.get_rank()
  meta = torch.zeros(9, dtype=torch.int64, device=dev)
  if rank == src:
    a = np.ascontiguousarray(arr)
    meta[0] = a.ndim
    meta[1] = 0 if a.dtype == np.float64 else 1
    for i, s in enumerate(a.shape):
      meta[2 + i] = s
  dist.broadcast(meta, src)
  m = meta.cpu().tolist()
  shape = tuple(int(v) for v in m[2:2 + int(m[0])])
  dtype = torch.float64 if m[1] == 0 else torch.int64
  if rank == src:
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
  else:
    t = torch.empty(shape, dtype=dtype, device=dev)
  dist.broadcast(t, src)
  return t.cpu().numpy()


def predict_batch_sharded(
    predict_fn: typing.Optional[typing.Callable[[np.ndarray], np.ndarray]],
    utterances: typing.Sequence[np.ndarray],
    predict_many_fn: typing.Optional[typing.Callable] = None) -> typing.List[np.ndarray]:
  """Every rank holds (or has been broadcast) the same utterance list; rank r runs
  `predict_fn` on its LPT share (or `predict_many_fn(list_of_arrays) -> list_of_labels`
  on the whole share at once, e.g. a multi-stream batch) and the int64 labels are
  all-gathered (padded to the longest utterance).  Every rank returns the complete,
  input-ordered result."""
  import torch
  dist = _dist()
  world, rank = dist.get_world_size(), dist.get_rank()
  dev = _device_for_backend(dist)
  sizes = [int(u.shape[0]) for u in utterances]
  owned = lpt_assignment(sizes, world)
  slots = max((len(o) for o in owned), default=0)
  longest = max(sizes, default=0)
  mine = torch.full((slots, longest), -1, dtype=torch.int64, device=dev)
  if predict_many_fn is not None:
    local = predict_many_fn([utterances[idx] for idx in owned[rank]])
  else:
    local = [predict_fn(utterances[idx]) for idx in owned[rank]]
  for s, lab in enumerate(local):
    lab = np.asarray(lab, dtype=np.int64)
    mine[s, :lab.shape[0]] = torch.from_numpy(lab).to(dev)
  gathered = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(gathered, mine)
  out = [None] * len(utterances)
  for r in range(world):
    block = gathered[r].cpu().numpy()
    for s, idx in enumerate(owned[r]):
      out[idx] = block[s, :sizes[idx]].copy()
  return out


def autotune_sharded(
    evaluate_fn: typing.Callable[[float], typing.Tuple[float, int]],
    grid: typing.Sequence[float]) -> typing.Tuple[np.ndarray, np.ndarray]:
  """One AutoTune search level (reference autotune.py:98-111): rank r evaluates
  grid[r::world]; `evaluate_fn(p) -> (ratio, n_clusters)`.  Returns the full
  (ratios, n_clusters) arrays on every rank (all_gather of 2 scalars per p)."""
  import torch
  dist = _dist()
  world, rank = dist.get_world_size(), dist.get_rank()
  dev = _device_for_backend(dist)
  per = (len(grid) + world - 1) // world
  mine = torch.full((per, 2), float("nan"), dtype=torch.float64, device=dev)
  for s, i in enumerate(range(rank, len(grid), world)):
    ratio, k = evaluate_fn(float(grid[i]))
    mine[s, 0] = float(ratio)
    mine[s, 1] = float(k)
  gathered = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(gathered, mine)
  ratios = np.full(len(grid), np.nan)
  ks = np.zeros(len(grid), dtype=np.int64)
  for r in range(world):
    block = gathered[r].cpu().numpy()
    for s, i in enumerate(range(r, len(grid), world)):
      ratios[i] = block[s, 0]
      ks[i] = int(block[s, 1])
  return ratios, ks


def first_strict_minimum(ratios: np.ndarray) -> int:
  """Index AutoTune.tune keeps: strict `<` scan, first index wins ties
  (reference autotune.py:106-111)."""
  best, best_i = np.inf, -1
  for i, r in enumerate(ratios):
    if r < best:
      best, best_i = r, i
  return best_i


# --------------------------------------------------------------------------------
# thin wrappers around a SpectralClusterer (device compute on this rank's GPU)
# --------------------------------------------------------------------------------
def predict_batch_distributed(clusterer, utterances: typing.Sequence[np.ndarray],
                              streams: int = 4) -> typing.List[np.ndarray]:
  """BASELINE config 5: utterances partitioned over the ranks (LPT), each rank runs its
  share as a multi-stream batch on its own GPU, labels all-gathered."""
  return predict_batch_sharded(
      None, utterances,
      predict_many_fn=lambda share: clusterer.predict_batch(share, streams=streams))


def predict_autotune_distributed(clusterer, embeddings: np.ndarray) -> np.ndarray:
  """BASELINE config 4: every rank holds the embeddings (broadcast them first if only
  rank 0 has them), recomputes the affinity locally (cheaper than shipping n^2 doubles),
  evaluates its share of each AutoTune search level, all-gathers (ratio, n_clusters),
  and every rank finishes with the winner's eigenvectors + k-means: identical labels
  everywhere, no n x n or n x k matrix ever crosses xGMI."""
  import ctypes
  from spectralcluster_amd import _lib
  tuner = clusterer.autotune
  if tuner is None:
    raise ValueError("clusterer.autotune is not set")
  handle = clusterer._handle()
  clusterer._scope_check()
  clusterer._upload(handle, embeddings)

  def evaluate(p):
    diag = clusterer._eig_resident(handle, p)
    return tuner.ratio(p, diag.max_delta), int(diag.n_clusters_raw)

  def evaluate_many(ps):
    ratios, ks = autotune_sharded(evaluate, ps)
    return [(float(r), p, int(k)) for r, p, k in zip(ratios, ps, ks)]

  _, n_clusters, best_p = tuner.tune(None, evaluate_many=evaluate_many)
  diag = clusterer._eig_resident(handle, best_p)
  if clusterer.min_clusters is not None:
    n_clusters = max(n_clusters, clusterer.min_clusters)
  n = embeddings.shape[0]
  labels = np.empty(n, dtype=np.int64)
  handle.check(handle.lib.sc_cluster(handle.raw, clusterer.build_config(best_p), n_clusters,
                                     _lib.as_int64_p(labels), diag))
  return labels
