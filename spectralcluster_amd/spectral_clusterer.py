"""SpectralClusterer facade: same constructor and predict() surface as the
reference `spectralcluster/spectral_clusterer.py`, with the dense hot path
(affinity -> refinement -> Laplacian -> top-k eigen + eigengap -> cosine k-means)
executed on one MI355X through the C ABI in `include/spectralcluster_amd.h`.

Constraints (`constraint_options` + `predict(embeddings, constraint_matrix)`) run on the
device too, and refinement sequences whose result is not
diagonally similar to a symmetric matrix take the general eigen path (8f-N2);
`max_spectral_size` pre-clusters on the device (cosine complete-linkage AHC) and
`fallback_options` (too-few-embeddings fallback, single-cluster test for min_clusters=1) run
there too (8f-N4).  `custom_dist`: cosine, euclidean (minkowski), sqeuclidean, cityblock,
chebyshev, correlation, braycurtis, canberra and scipy's aliases of them; metrics that need more
than the two vectors and callables are out of the device scope.  Those raise
`UnsupportedOnDeviceError`; nothing silently falls back to the CPU.
"""

from __future__ import annotations

import ctypes
import typing

import numpy as np

from spectralcluster_amd import _lib
from spectralcluster_amd import autotune as autotune_lib
from spectralcluster_amd import constraint as constraint_lib
from spectralcluster_amd import custom_distance_kmeans
from spectralcluster_amd import fallback_clusterer
from spectralcluster_amd import laplacian
from spectralcluster_amd import refinement
from spectralcluster_amd import utils

AutoTune = autotune_lib.AutoTune
AutoTuneProxy = autotune_lib.AutoTuneProxy
LaplacianType = laplacian.LaplacianType
RefinementName = refinement.RefinementName
RefinementOptions = refinement.RefinementOptions
EigenGapType = utils.EigenGapType


class SpectralClusterer:
  """Spectral clustering of speaker embeddings on the GPU.

  Constructor arguments are those of the reference (spectral_clusterer.py:29-46),
  plus `device` (HIP device ordinal; default: LOCAL_RANK or 0).
  """

  def __init__(self,
               min_clusters: typing.Optional[int] = None,
               max_clusters: typing.Optional[int] = None,
               refinement_options: typing.Optional[RefinementOptions] = None,
               autotune: typing.Optional[AutoTune] = None,
               fallback_options=None,
               laplacian_type: typing.Optional[LaplacianType] = None,
               stop_eigenvalue: float = 1e-2,
               row_wise_renorm: bool = False,
               custom_dist: typing.Union[str, typing.Callable] = "cosine",
               max_iter: int = 300,
               constraint_options=None,
               eigengap_type: EigenGapType = EigenGapType.Ratio,
               max_spectral_size: typing.Optional[int] = None,
               affinity_function: typing.Callable = utils.compute_affinity_matrix,
               post_eigen_cluster_function: typing.Callable = (
                   custom_distance_kmeans.run_kmeans),
               device: typing.Optional[int] = None):
    self.min_clusters = min_clusters
    self.max_clusters = max_clusters
    self.refinement_options = refinement_options or RefinementOptions()
    self.autotune = autotune
    self.fallback_options = fallback_options or fallback_clusterer.FallbackOptions()
    self.laplacian_type = laplacian_type
    self.row_wise_renorm = row_wise_renorm
    self.stop_eigenvalue = stop_eigenvalue
    self.custom_dist = custom_dist
    self.max_iter = max_iter
    self.constraint_options = constraint_options
    self.eigengap_type = eigengap_type
    self.max_spectral_size = max_spectral_size
    self.affinity_function = affinity_function
    self.post_eigen_cluster_function = post_eigen_cluster_function
    self.device = device
    self.last_diag: typing.Optional[_lib.ScDiag] = None
    # not in the reference: restart cycles block Lanczos may spend before the dense
    # eigensolver takes over (0 = the library default, 40); predict() returns either way
    self.eig_max_cycles = 0
    # not in the reference: bound on |eigenvalue error| / |eigenvalue| for the values the
    # eigengap rule reads (0 = the library default, 1e-6; the parity bar is 1e-5)
    self.eig_value_tol = 0.0
    # not in the reference: route of a Diffuse that only feeds RowWiseNormalize / the Laplacian
    # (sc_config.diffuse_mode): 0 default (matrix-free from n = 2048 on), 1 explicit fp64
    # product, 2 matrix-free wherever the sequence allows it.  Same results to summation order.
    self.diffuse_mode = 0

  # ----------------------------------------------------------------- plumbing
  def _handle(self) -> _lib.Handle:
    return _lib.default_handle(self.device)

  def build_config(self, p_percentile: typing.Optional[float] = None) -> _lib.ScConfig:
    """Flatten the constructor arguments into an `sc_config`."""
    cfg = _lib.ScConfig()
    _lib.load().sc_config_default(cfg)
    self.refinement_options.to_config(cfg)
    if p_percentile is not None:
      cfg.p_percentile = float(p_percentile)
    if self.laplacian_type is None:
      cfg.laplacian_type = 0
    elif isinstance(self.laplacian_type, LaplacianType):
      cfg.laplacian_type = self.laplacian_type.value
    else:
      raise TypeError("laplacian_type must be a LaplacianType")
    if not isinstance(self.eigengap_type, EigenGapType):
      raise TypeError("eigengap_type must be a EigenGapType")
    cfg.eigengap_type = self.eigengap_type.value
    cfg.min_clusters = int(self.min_clusters or 0)
    cfg.max_clusters = int(self.max_clusters or 0)
    cfg.stop_eigenvalue = float(self.stop_eigenvalue)
    cfg.row_wise_renorm = int(bool(self.row_wise_renorm))
    cfg.max_iter = int(self.max_iter)
    cfg.eig_max_cycles = int(getattr(self, "eig_max_cycles", 0) or 0)
    if getattr(self, "eig_value_tol", 0.0):
      cfg.eig_value_tol = float(self.eig_value_tol)
    cfg.diffuse_mode = int(getattr(self, "diffuse_mode", 0) or 0)
    if self.post_eigen_cluster_function is custom_distance_kmeans.run_kmeans:
      cfg.kmeans_metric = _lib.kmeans_metric_code(self.custom_dist)
    if self.constraint_options is not None:
      self.constraint_options.to_config(cfg)
    if getattr(cfg, "_blur_ext", None) is not None:  # sigma > 8: weights go to the handle
      _lib.sync_blur_weights(self._handle(), cfg)
    return cfg

  def _set_constraint(self, handle: _lib.Handle, n: int, constraint_matrix) -> bool:
    """Make `constraint_matrix` resident (or clear a stale one).  Like the reference
    (spectral_clusterer.py:137-142, 259-264) it is used only when both
    `constraint_options` and the matrix are given."""
    if self.constraint_options is None or constraint_matrix is None:
      handle.check(handle.lib.sc_clear_constraint(handle.raw))
      return False
    con = np.asarray(constraint_matrix)
    # ConstraintOperation.check_input against the (n, n) affinity: only its shape is read
    self.constraint_options.constraint_operator.check_input(
        np.lib.stride_tricks.as_strided(np.zeros(1, dtype=bool), (n, n), (0, 0)), con)
    con = np.ascontiguousarray(con, dtype=np.float64)
    handle.check(handle.lib.sc_set_constraint(handle.raw, _lib.as_double_p(con), n))
    return True

  def _upload(self, handle: _lib.Handle, embeddings: np.ndarray):
    """Embeddings -> resident affinity (device GEMM, or the user's function)."""
    if self.affinity_function is utils.compute_affinity_matrix:
      x = np.ascontiguousarray(embeddings, dtype=np.float64)
      handle.check(handle.lib.sc_set_embeddings(
          handle.raw, _lib.as_double_p(x), x.shape[0], x.shape[1]))
      handle.check(handle.lib.sc_compute_affinity(handle.raw))
    else:
      a = np.ascontiguousarray(self.affinity_function(embeddings), dtype=np.float64)
      handle.check(handle.lib.sc_set_affinity(handle.raw, _lib.as_double_p(a),
                                              a.shape[0]))

  def _eig_resident(self, handle: _lib.Handle, p_percentile=None) -> _lib.ScDiag:
    diag = _lib.ScDiag()
    cfg = self.build_config(p_percentile)
    handle.check(handle.lib.sc_eig_ncluster(handle.raw, cfg, diag), TypeError)
    self.last_diag = diag
    return diag

  def _eig_sweep(self, handle: _lib.Handle,
                 p_values: typing.Sequence[float]) -> typing.List[_lib.ScDiag]:
    """`_eig_resident` for every p of one AutoTune level (no eigenvectors left resident)."""
    count = len(p_values)
    diags = (_lib.ScDiag * count)()
    ps = (ctypes.c_double * count)(*[float(p) for p in p_values])
    handle.check(handle.lib.sc_eig_ncluster_sweep(handle.raw, self.build_config(), ps, count,
                                                  diags), TypeError)
    self.last_sweep_diags = list(diags)
    self._last_sweep_ps = [float(p) for p in p_values]
    return self.last_sweep_diags

  def _adopt_or_evaluate(self, handle: _lib.Handle, p: float) -> _lib.ScDiag:
    """Make the eigenvectors for p_percentile `p` resident: adopted from the last sweep when
    `p` was one of its values (the AutoTune winner normally is: the reference keeps the
    winner's eigenvectors from the search, spectral_clusterer.py:274-292), evaluated with
    `_eig_resident` otherwise."""
    ps = getattr(self, "_last_sweep_ps", None)
    if ps and float(p) in ps:
      diag = _lib.ScDiag()
      if handle.lib.sc_sweep_adopt(handle.raw, self.build_config(p), ps.index(float(p)),
                                   diag) == _lib.SC_OK:
        self.last_diag = diag
        return diag
    return self._eig_resident(handle, p)

  def consumed_eigenvalues(self) -> np.ndarray:
    """Every eigenvalue the last eigen call consumed, in the reference's order
    (`compute_sorted_eigenvectors`, utils.py:62-70): max_clusters + 1 of them, or all n
    with max_clusters=None and a Laplacian (`last_diag.eigenvalues` holds at most 128)."""
    handle = self._handle()
    count = handle.lib.sc_num_eigenvalues(handle.raw)
    out = np.empty(count, dtype=np.float64)
    handle.check(handle.lib.sc_get_eigenvalues(handle.raw, _lib.as_double_p(out), count))
    return out

  def _download_eigenvectors(self, handle: _lib.Handle, n: int,
                             cols: typing.Optional[int] = None) -> np.ndarray:
    have = handle.lib.sc_num_eigenvectors(handle.raw)
    cols = have if cols is None else cols
    out = np.empty((n, cols), dtype=np.float64)
    handle.check(handle.lib.sc_get_eigenvectors(handle.raw, _lib.as_double_p(out), n,
                                                cols))
    return out

  # ------------------------------------------------------------- reference API
  def _compute_eigenvectors_ncluster(
      self, affinity: np.ndarray, constraint_matrix=None
  ) -> typing.Tuple[np.ndarray, int, float]:
    """Refinement + eigen-decomposition + eigengap for a given affinity matrix
    (reference spectral_clusterer.py:108-168).

    Returns (eigenvectors, n_clusters, max_delta_norm).  For n <= 128 the
    eigenvector matrix is (n, n) like the reference's; above that it has only the
    columns the eigengap search can select (max_clusters + 1, at most 64 from a Krylov solve;
    on the dense routes -- max_clusters=None with a Laplacian, where every eigenvalue is read,
    more than 64 selected clusters, the general path up to n = 512 -- the
    max(n_clusters, min_clusters) columns predict() uses).
    """
    a = np.ascontiguousarray(affinity, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
      raise ValueError("affinity must be a square matrix")
    handle = self._handle()
    handle.check(handle.lib.sc_set_affinity(handle.raw, _lib.as_double_p(a), a.shape[0]))
    # only the after-refinement branch lives in this method (reference :137-142)
    self._set_constraint(handle, a.shape[0], constraint_matrix)
    diag = self._eig_resident(handle)
    vectors = self._download_eigenvectors(handle, a.shape[0])
    return vectors, int(diag.n_clusters_raw), float(diag.max_delta)

  def _reduce_size_and_predict(self, embeddings: np.ndarray) -> np.ndarray:
    """Complete-linkage cosine AHC down to `max_spectral_size` clusters, spectral
    clustering of their centroids, labels chained back (reference :170-199)."""
    ahc_labels = utils.cosine_agglomerative_clustering(
        embeddings, n_clusters=self.max_spectral_size, linkage="complete")
    ahc_centroids = utils.get_cluster_centroids(embeddings, ahc_labels)
    spectral_labels = self.predict(ahc_centroids)
    return utils.chain_labels(ahc_labels, spectral_labels)

  def predict(self, embeddings: np.ndarray, constraint_matrix=None) -> np.ndarray:
    """Cluster `embeddings` (n_samples, n_features); returns int64 labels
    (reference spectral_clusterer.py:201-314)."""
    with _lib.use_device(self.device):  # helpers without a device argument follow self.device
      return self._predict(embeddings, constraint_matrix)

  def _predict(self, embeddings: np.ndarray, constraint_matrix=None) -> np.ndarray:
    if not isinstance(embeddings, np.ndarray):
      raise TypeError("embeddings must be a numpy array")
    if len(embeddings.shape) != 2:
      raise ValueError("embeddings must be 2-dimensional")
    n = embeddings.shape[0]
    if n < self.fallback_options.spectral_min_embeddings:
      # too few embeddings for spectral clustering (reference :229-233)
      return fallback_clusterer.FallbackClusterer(self.fallback_options).predict(embeddings)
    if self.max_spectral_size is not None and n > self.max_spectral_size:
      # reference spectral_clusterer.py:236-248
      if constraint_matrix is not None:
        raise RuntimeError(
            "Cannot handle constraint_matrix when max_spectral_size is set")
      if (self.max_spectral_size < 2 or
          (self.max_clusters and self.max_spectral_size <= self.max_clusters) or
          (self.min_clusters and self.max_spectral_size <= self.min_clusters)):
        raise ValueError("max_spectral_size should be a relatively big number")
      return self._reduce_size_and_predict(embeddings)
    handle = self._handle()
    default_tail = (self.post_eigen_cluster_function is custom_distance_kmeans.run_kmeans)
    if default_tail:
      _lib.kmeans_metric_code(self.custom_dist)  # raises for metrics that are not on the device
    constrained = self._set_constraint(handle, n, constraint_matrix)

    single_check = self.min_clusters == 1
    by_fallback = (self.fallback_options.single_cluster_condition ==
                   fallback_clusterer.SingleClusterCondition.FallbackClusterer)
    if single_check and by_fallback:
      # this condition only needs the embeddings; it runs on the same device handle, so
      # it goes first and the affinity is built afterwards (reference :253-256)
      if fallback_clusterer.check_single_cluster(self.fallback_options, embeddings, None):
        return np.array([0] * n)

    if (self.autotune is None and default_tail and not single_check
        and self.affinity_function is utils.compute_affinity_matrix):
      # the whole path in one call: H2D(X), device pipeline, D2H(labels)
      x = np.ascontiguousarray(embeddings, dtype=np.float64)
      labels = np.empty(n, dtype=np.int64)
      diag = _lib.ScDiag()
      handle.check(handle.lib.sc_predict(
          handle.raw, _lib.as_double_p(x), n, x.shape[1], self.build_config(),
          _lib.as_int64_p(labels), diag), TypeError)
      self.last_diag = diag
      return labels

    self._upload(handle, embeddings)
    if single_check and not by_fallback:
      # single-vs-multi cluster(s) decision on the resident affinity (reference :253-256)
      if fallback_clusterer.check_single_cluster(self.fallback_options, embeddings, None,
                                                 _resident_on=handle):
        return np.array([0] * n)
    if constrained and self.constraint_options.apply_before_refinement:
      # reference :259-264 -- once, before the AutoTune sweep re-reads the affinity
      handle.check(handle.lib.sc_apply_constraint(handle.raw, self.build_config()))
    if self.autotune:
      sequence = self.refinement_options.refinement_sequence or []
      if RefinementName.RowWiseThreshold not in sequence:
        raise ValueError(
            "AutoTune is only effective when the refinement sequence"
            "contains RowWiseThreshold")
      evaluated = []

      def evaluate_level(ps):
        # one search level = one library call: the values of a level differ only in the row
        # threshold, the device evaluates them as a group (sc_eig_ncluster_sweep)
        diags = self._eig_sweep(handle, ps)
        evaluated.extend(ps)
        return [(self.autotune.ratio(p, d.max_delta), p, int(d.n_clusters_raw))
                for p, d in zip(ps, diags)]

      _, n_clusters, best_p = self.autotune.tune(None, evaluate_many=evaluate_level)
      # reference closure leaves refinement_options.p_percentile at the LAST
      # evaluated value (spectral_clusterer.py:277); keep that observable state
      self.refinement_options.p_percentile = evaluated[-1]
      self.last_best_p = best_p  # (not in the reference: the value the labels come from)
      diag = self._adopt_or_evaluate(handle, best_p)  # the winner's vectors, resident
    else:
      diag = self._eig_resident(handle)
      n_clusters = int(diag.n_clusters_raw)

    if self.min_clusters is not None:
      n_clusters = max(n_clusters, self.min_clusters)

    if default_tail:
      labels = np.empty(n, dtype=np.int64)
      handle.check(handle.lib.sc_cluster(handle.raw, self.build_config(), n_clusters,
                                         _lib.as_int64_p(labels), diag))
      return labels
    # user-supplied post_eigen_cluster_function: hand it the spectral embedding
    spectral = self._download_eigenvectors(handle, n, n_clusters)
    if self.row_wise_renorm:
      spectral = spectral / np.linalg.norm(spectral, axis=1, ord=2)[:, None]
    return self.post_eigen_cluster_function(
        spectral_embeddings=spectral, n_clusters=n_clusters,
        custom_dist=self.custom_dist, max_iter=self.max_iter)

  # -------------------------------------------------------------- batch (new)
  def predict_batch(self, utterances: typing.Sequence[np.ndarray],
                    streams: typing.Optional[int] = None,
                    group: typing.Optional[int] = None) -> typing.List[np.ndarray]:
    """Independent predict() calls (the reference has no batch API: a batch is a
    Python loop).

    Small utterances cannot fill 256 CUs, and their pipeline is a chain of short dependent
    launches.  Two ways around that, both ONE library call with the GIL released:
      group       `sc_predict_batch_grouped` (the default, group=16): `group` (<= 16)
                  utterances per launch -- the eigensolver and k-means chains of a group
                  advance in lockstep on one stream while the GEMMs and refinement passes of
                  the next group run on another; the groups are dealt to three lanes (a host
                  thread and a set of streams each) inside the call;
      streams     `sc_predict_batch_streams` (when `streams` is given and `group` is not):
                  the batch spread (longest-processing-time first) over `streams` HIP
                  streams, one host thread and arena per stream; streams=1 is a plain loop.
    Per-utterance results agree with predict() to the solver's tolerance, not bit for bit:
    the grouped GEMMs sum K in whole tiles where a short single call splits K, and the
    lockstep eigensolver checks convergence on the group's schedule, so a member can leave
    with a different basis size (eigenvalues within 1e-6 relative, labels equal unless an
    eigengap decision is a near tie).  A batch call is deterministic: the same list gives
    the same results.
    """
    if group is None:
      group = 16 if streams is None else 0
    if streams is None:
      streams = 1
    if (self.autotune is not None or self.max_spectral_size is not None or
        self.min_clusters == 1 or self.fallback_options.spectral_min_embeddings > 1 or
        self.affinity_function is not utils.compute_affinity_matrix or
        self.post_eigen_cluster_function is not custom_distance_kmeans.run_kmeans):
      # anything sc_predict_batch does not cover (user-supplied affinity / clustering
      # functions, AutoTune, size reduction, fallback decisions) goes through predict().
      # A batch never carries a constraint matrix -- same as predict(u) without one.
      return [self.predict(u) for u in utterances]
    _lib.kmeans_metric_code(self.custom_dist)  # raises for metrics that are not on the device
    if not utterances:
      return []
    for u in utterances:
      if not isinstance(u, np.ndarray):
        raise TypeError("embeddings must be a numpy array")
    xs = [np.ascontiguousarray(u, dtype=np.float64) for u in utterances]
    d = xs[0].shape[1] if xs[0].ndim == 2 else -1
    for x in xs:
      if x.ndim != 2 or x.shape[1] != d:
        raise ValueError("all utterances must be (n_i, d) with the same d")
    count = len(xs)
    labels = [np.empty(x.shape[0], dtype=np.int64) for x in xs]
    handle = self._handle()
    handle.check(handle.lib.sc_clear_constraint(handle.raw))  # a batch carries none
    xp = (ctypes.POINTER(ctypes.c_double) * count)(*[_lib.as_double_p(x) for x in xs])
    lp = (ctypes.POINTER(ctypes.c_int64) * count)(*[_lib.as_int64_p(l) for l in labels])
    ns = (ctypes.c_int * count)(*[x.shape[0] for x in xs])
    diags = (_lib.ScDiag * count)()
    if int(group) > 1:
      handle.check(handle.lib.sc_predict_batch_grouped(
          handle.raw, xp, ns, d, count, self.build_config(), lp, diags, int(group)),
          TypeError)
    else:
      handle.check(handle.lib.sc_predict_batch_streams(
          handle.raw, xp, ns, d, count, self.build_config(), lp, diags,
          max(1, int(streams))), TypeError)
    self.last_batch_diags = list(diags)
    return labels
