"""ctypes binding of libspectralcluster_amd.so (include/spectralcluster_amd.h).

The HIP library IS the product path: there is no NumPy fallback.  If the shared
object is missing, or no MI355X is visible, every compute entry point raises.
"""

from __future__ import annotations

import ctypes
import os
import threading
import typing

import numpy as np

SC_MAX_OPS = 16
SC_MAX_BLUR_RADIUS = 32
SC_MAX_EIG = 128
SC_MAX_STAGES = 16

SC_OK = 0
SC_ERR_INVALID = -1
SC_ERR_OOM = -2
SC_ERR_HIP = -3
SC_ERR_NOT_CONVERGED = -4
SC_ERR_UNSUPPORTED = -5
SC_ERR_NON_FINITE = -6

STAGE_NAMES = ("affinity", "refine", "diffuse", "scaling", "eig", "kmeans",
               "total", "blur", "threshold_sym", "matvec", "affinity_gemm",
               "free_quantize", "free_product", "free_scan", "free_stats")

DIFFUSE_PATH_NONE, DIFFUSE_PATH_EXPLICIT, DIFFUSE_PATH_FREE, DIFFUSE_PATH_FREE_THEN_EXPLICIT = range(4)

_LIB_NAME = "libspectralcluster_amd.so"
_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


class DeviceLibraryError(RuntimeError):
  """The HIP shared library (or a device) is not available."""


class UnsupportedOnDeviceError(NotImplementedError):
  """The requested configuration is outside the device hot path."""


class EigenSolverNotConverged(RuntimeError):
  """The block-Lanczos eigensolver did not reach its tolerance."""


SC_ABI_VERSION = 7


class ScConfig(ctypes.Structure):
  """Mirror of `sc_config`."""
  _fields_ = [
      ("n_ops", ctypes.c_int32),
      ("ops", ctypes.c_int32 * SC_MAX_OPS),
      ("blur_radius", ctypes.c_int32),
      ("blur_weights", ctypes.c_double * (2 * SC_MAX_BLUR_RADIUS + 1)),
      ("p_percentile", ctypes.c_double),
      ("soft_multiplier", ctypes.c_double),
      ("threshold_type", ctypes.c_int32),
      ("binarize", ctypes.c_int32),
      ("preserve_diagonal", ctypes.c_int32),
      ("symmetrize_type", ctypes.c_int32),
      ("laplacian_type", ctypes.c_int32),
      ("min_clusters", ctypes.c_int32),
      ("max_clusters", ctypes.c_int32),
      ("stop_eigenvalue", ctypes.c_double),
      ("eigengap_type", ctypes.c_int32),
      ("row_wise_renorm", ctypes.c_int32),
      ("max_iter", ctypes.c_int32),
      ("eig_value_tol", ctypes.c_double),
      ("eig_vector_tol", ctypes.c_double),
      ("eig_max_cycles", ctypes.c_int32),
      ("constraint_name", ctypes.c_int32),
      ("constraint_before_refinement", ctypes.c_int32),
      ("integration_type", ctypes.c_int32),
      ("constraint_alpha", ctypes.c_double),
      ("kmeans_metric", ctypes.c_int32),
      ("diffuse_mode", ctypes.c_int32),
      ("reserved", ctypes.c_int32 * 4),
  ]


class ScDiag(ctypes.Structure):
  """Mirror of `sc_diag`."""
  _fields_ = [
      ("n", ctypes.c_int32),
      ("n_clusters_raw", ctypes.c_int32),
      ("n_clusters", ctypes.c_int32),
      ("eig_path", ctypes.c_int32),
      ("max_delta", ctypes.c_double),
      ("n_eigenvalues", ctypes.c_int32),
      ("eig_descending", ctypes.c_int32),
      ("eigenvalues", ctypes.c_double * SC_MAX_EIG),
      ("eig_matvec_passes", ctypes.c_int32),
      ("eig_block", ctypes.c_int32),
      ("eig_basis", ctypes.c_int32),
      ("eig_cycles", ctypes.c_int32),
      ("eig_max_residual", ctypes.c_double),
      ("kmeans_iterations", ctypes.c_int32),
      ("symmetry_state", ctypes.c_int32),
      ("eig_host_chain", ctypes.c_int32),
      ("eig_fallback", ctypes.c_int32),
      ("stage_ms", ctypes.c_float * SC_MAX_STAGES),
      ("diffuse_path", ctypes.c_int32),
      ("free_candidates", ctypes.c_int32),
      ("free_overflow_rows", ctypes.c_int32),
      ("free_tiles_run", ctypes.c_int32),
  ]

  def eigenvalue_array(self) -> np.ndarray:
    return np.array(self.eigenvalues[:self.n_eigenvalues], dtype=np.float64)

  def stage_times_ms(self) -> dict:
    return {name: float(self.stage_ms[i]) for i, name in enumerate(STAGE_NAMES)}


_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_handle_t = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/spectralcluster_amd.h declares
PROTOTYPES = {
    "sc_abi_version": (ctypes.c_int, []),
    "sc_struct_sizes": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int)]),
    "sc_device_count": (ctypes.c_int, []),
    "sc_device_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.c_char_p, ctypes.c_int, _c_int_p,
                                      _c_int64_p]),
    "sc_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_handle_t)]),
    "sc_destroy": (ctypes.c_int, [_handle_t]),
    "sc_reserve": (ctypes.c_int, [_handle_t, ctypes.c_int, ctypes.c_int]),
    "sc_last_error": (ctypes.c_char_p, [_handle_t]),
    "sc_synchronize": (ctypes.c_int, [_handle_t]),
    "sc_set_profiling": (ctypes.c_int, [_handle_t, ctypes.c_int]),
    "sc_set_diffuse_mode": (ctypes.c_int, [_handle_t, ctypes.c_int]),
    "sc_set_free_prune": (ctypes.c_int, [_handle_t, ctypes.c_int]),
    "sc_stage_diffuse_rowstats": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                                 ctypes.c_int, _c_double_p, _c_double_p,
                                                 ctypes.POINTER(ctypes.c_int32)]),
    "sc_config_default": (ctypes.c_int, [ctypes.POINTER(ScConfig)]),
    "sc_gaussian_weights": (ctypes.c_int, [ctypes.c_double,
                                           ctypes.POINTER(ctypes.c_int32),
                                           _c_double_p]),
    "sc_predict": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ScConfig), _c_int64_p,
                                  ctypes.POINTER(ScDiag)]),
    "sc_set_embeddings": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                         ctypes.c_int]),
    "sc_compute_affinity": (ctypes.c_int, [_handle_t]),
    "sc_set_affinity": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int]),
    "sc_eig_ncluster": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig),
                                       ctypes.POINTER(ScDiag)]),
    "sc_num_eigenvalues": (ctypes.c_int, [_handle_t]),
    "sc_get_eigenvalues": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int]),
    "sc_num_eigenvectors": (ctypes.c_int, [_handle_t]),
    "sc_get_eigenvectors": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                           ctypes.c_int]),
    "sc_cluster": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig), ctypes.c_int,
                                  _c_int64_p, ctypes.POINTER(ScDiag)]),
    "sc_set_constraint": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int]),
    "sc_clear_constraint": (ctypes.c_int, [_handle_t]),
    "sc_apply_constraint": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig)]),
    "sc_stage_constraint": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig), _c_double_p,
                                           _c_double_p, ctypes.c_int, _c_double_p]),
    "sc_ahc": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, ctypes.c_double, _c_int64_p,
                              ctypes.POINTER(ctypes.c_int)]),
    "sc_cluster_centroids": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int, ctypes.c_int,
                                            _c_int64_p, ctypes.c_int, _c_double_p]),
    "sc_affinity_stats": (ctypes.c_int, [_handle_t, _c_double_p]),
    "sc_affinity_gmm_bic": (ctypes.c_int, [_handle_t, ctypes.c_int, _c_double_p, _c_double_p]),
    "sc_naive_cluster": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, _c_double_p,
                                        ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.c_int,
                                        _c_int64_p]),
    "sc_run_resident": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig),
                                       _c_int64_p, ctypes.POINTER(ScDiag)]),
    "sc_predict_batch": (ctypes.c_int, [_handle_t, ctypes.POINTER(_c_double_p),
                                        _c_int_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ScConfig),
                                        ctypes.POINTER(_c_int64_p),
                                        ctypes.POINTER(ScDiag)]),
    "sc_predict_batch_streams": (ctypes.c_int, [_handle_t, ctypes.POINTER(_c_double_p),
                                                _c_int_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ScConfig),
                                                ctypes.POINTER(_c_int64_p),
                                                ctypes.POINTER(ScDiag), ctypes.c_int]),
    "sc_eig_ncluster_sweep": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig), _c_double_p,
                                             ctypes.c_int, ctypes.POINTER(ScDiag)]),
    "sc_set_blur_weights": (ctypes.c_int, [_handle_t, ctypes.c_int, _c_double_p]),
    "sc_sweep_adopt": (ctypes.c_int, [_handle_t, ctypes.POINTER(ScConfig), ctypes.c_int,
                                      ctypes.POINTER(ScDiag)]),
    "sc_predict_batch_grouped": (ctypes.c_int, [_handle_t, ctypes.POINTER(_c_double_p),
                                                _c_int_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ScConfig),
                                                ctypes.POINTER(_c_int64_p),
                                                ctypes.POINTER(ScDiag), ctypes.c_int]),
    "sc_stage_affinity": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                         ctypes.c_int, _c_double_p]),
    "sc_stage_refine": (ctypes.c_int, [_handle_t, ctypes.c_int,
                                       ctypes.POINTER(ScConfig), _c_double_p,
                                       ctypes.c_int, _c_double_p]),
    "sc_stage_laplacian": (ctypes.c_int, [_handle_t, ctypes.c_int, _c_double_p,
                                          ctypes.c_int, _c_double_p]),
    "sc_stage_sym_eig": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, _c_double_p,
                                        _c_double_p, ctypes.POINTER(ScDiag)]),
    "sc_stage_eig": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, _c_double_p,
                                        _c_double_p, ctypes.POINTER(ScDiag)]),
    "sc_random_state_doubles": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_int,
                                               _c_double_p]),
    "sc_uniform_choice": (ctypes.c_int, [ctypes.c_int, ctypes.c_double]),
    "sc_host_symmetric_eig": (ctypes.c_int, [_c_double_p, ctypes.c_int, _c_double_p,
                                             _c_double_p]),
    "sc_host_symmetric_eig_partial": (ctypes.c_int, [_c_double_p, ctypes.c_int, ctypes.c_int,
                                                     _c_double_p, _c_double_p]),
    "sc_host_tridiag_eigvectors": (ctypes.c_int, [_c_double_p, _c_double_p, ctypes.c_int,
                                                  _c_double_p, ctypes.c_int, _c_double_p]),
    "sc_host_general_eig": (ctypes.c_int, [_c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p,
                                           _c_double_p, _c_double_p, _c_double_p]),
    "sc_host_general_eig_fast": (ctypes.c_int, [_c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p,
                                           _c_double_p, _c_double_p, _c_double_p]),
    "sc_host_hessenberg_eig": (ctypes.c_int, [_c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int32), _c_double_p,
                                              _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "sc_host_value_error_bound": (ctypes.c_int, [_c_double_p, _c_double_p, ctypes.c_int,
                                                 ctypes.c_int, _c_double_p]),
    "sc_eigengap": (ctypes.c_int, [_c_double_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   _c_int_p, _c_double_p]),
    "sc_stage_kmeans": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, _c_int64_p,
                                       _c_double_p, _c_int_p]),
    "sc_stage_kmeans_metric": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              _c_int64_p, _c_double_p, _c_int_p]),
    "sc_comm_available": (ctypes.c_int, []),
    "sc_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "sc_comm_init_rank": (ctypes.c_int, [_handle_t, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_char_p, ctypes.POINTER(_handle_t)]),
    "sc_comm_init_all": (ctypes.c_int, [ctypes.POINTER(_handle_t), ctypes.c_int,
                                        ctypes.POINTER(_handle_t)]),
    "sc_comm_destroy": (ctypes.c_int, [_handle_t]),
    "sc_comm_rank": (ctypes.c_int, [_handle_t]),
    "sc_comm_size": (ctypes.c_int, [_handle_t]),
    "sc_comm_last_error": (ctypes.c_char_p, [_handle_t]),
    "sc_comm_broadcast": (ctypes.c_int, [_handle_t, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_int]),
    "sc_comm_allgather": (ctypes.c_int, [_handle_t, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_size_t]),
    "sc_comm_allreduce_max": (ctypes.c_int, [_handle_t, _c_double_p, ctypes.c_int]),
    "sc_comm_barrier": (ctypes.c_int, [_handle_t]),
}

# custom_dist values that run on the device (scipy cdist names)
KMEANS_METRICS = {"cosine": 0, "euclidean": 1, "sqeuclidean": 2, "cityblock": 3,
                  "chebyshev": 4, "correlation": 5, "braycurtis": 6, "canberra": 7,
                  # scipy's cdist without a `p` argument (the reference passes none,
                  # custom_distance_kmeans.py:123-124) takes p = 2: the Euclidean distance
                  "minkowski": 1}
# ... and the aliases scipy.spatial.distance accepts for them (scipy 1.15 _METRIC_INFOS;
# cdist lower-cases the name first)
_KMEANS_METRIC_ALIASES = {
    "cos": "cosine", "e": "euclidean", "eu": "euclidean", "euclid": "euclidean",
    "sqe": "sqeuclidean", "sqeuclid": "sqeuclidean", "c": "cityblock", "cb": "cityblock",
    "cblock": "cityblock", "ch": "chebyshev", "cheb": "chebyshev", "cheby": "chebyshev",
    "chebychev": "chebyshev", "co": "correlation", "m": "minkowski", "mi": "minkowski",
    "pnorm": "minkowski"}


class NotFittedError(ValueError, AttributeError):
  """Same bases as sklearn.exceptions.NotFittedError, which the reference raises for a
  falsy custom_dist."""


def kmeans_metric_code(custom_dist) -> int:
  """sc_config.kmeans_metric for a `custom_dist` argument (reference
  custom_distance_kmeans.py:13-52)."""
  if not custom_dist:
    # reference :33-36 builds an sklearn KMeans and :51 calls .predict() on it without
    # ever fitting it: custom_dist=None / "" always ends in sklearn's NotFittedError
    raise NotFittedError(
        "This KMeans instance is not fitted yet. Call 'fit' with appropriate arguments "
        "before using this estimator.")
  if isinstance(custom_dist, str):
    name = custom_dist.lower()
    name = _KMEANS_METRIC_ALIASES.get(name, name)
    if name in KMEANS_METRICS:
      return KMEANS_METRICS[name]
  raise UnsupportedOnDeviceError(
      "custom_dist=%r: the device path implements %s (and scipy's aliases of those); other "
      "scipy metrics and callables are not on it" % (custom_dist, ", ".join(sorted(KMEANS_METRICS))))


_lib = None
_lib_lock = threading.Lock()


def library_path() -> str:
  return os.environ.get("SPECTRALCLUSTER_AMD_LIB", os.path.join(_LIB_DIR, _LIB_NAME))


def load() -> ctypes.CDLL:
  """Load the shared library (once) and declare every prototype."""
  global _lib
  if _lib is not None:
    return _lib
  with _lib_lock:
    if _lib is not None:
      return _lib
    path = library_path()
    if not os.path.exists(path):
      raise DeviceLibraryError(
          "%s not found: build it with `make -C spectralcluster_amd/csrc` "
          "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is "
          "no CPU fallback for the hot path." % path)
    try:
      lib = ctypes.CDLL(path)
    except OSError as exc:  # missing libamdhip64 etc.
      raise DeviceLibraryError("cannot load %s: %s" % (path, exc)) from exc
    for name, (restype, argtypes) in PROTOTYPES.items():
      fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
      fn.restype = restype
      fn.argtypes = argtypes
    if lib.sc_abi_version() != SC_ABI_VERSION:
      raise DeviceLibraryError("ABI version mismatch in %s (rebuild it)" % path)
    cfg_bytes, diag_bytes = ctypes.c_int(0), ctypes.c_int(0)
    lib.sc_struct_sizes(ctypes.byref(cfg_bytes), ctypes.byref(diag_bytes))
    if (cfg_bytes.value != ctypes.sizeof(ScConfig) or
        diag_bytes.value != ctypes.sizeof(ScDiag)):
      raise DeviceLibraryError("sc_config / sc_diag layout mismatch in %s" % path)
    _lib = lib
    return lib


def device_count() -> int:
  return int(load().sc_device_count())


def as_double_p(a: np.ndarray):
  return a.ctypes.data_as(_c_double_p)


def sync_blur_weights(handle, cfg) -> None:
  """A GaussianBlur of sigma > 8 has more weights than `sc_config` holds (radius > 32):
  `refinement.fill_config` leaves them on the config object and they are uploaded to the
  handle here, before the config is used with it (sc_set_blur_weights)."""
  w = getattr(cfg, "_blur_ext", None)
  if w is not None:
    handle.check(handle.lib.sc_set_blur_weights(handle.raw, int(cfg.blur_radius),
                                                as_double_p(w)))


def as_int64_p(a: np.ndarray):
  return a.ctypes.data_as(_c_int64_p)


class Handle:
  """Owns one `sc_handle` (one device, one stream).  Not thread-safe."""

  def __init__(self, device: int = 0):
    self._lib = load()
    if self._lib.sc_device_count() <= 0:
      raise DeviceLibraryError(
          "no HIP device visible: the hot path runs on an MI355X only "
          "(there is no CPU fallback)")
    h = _handle_t()
    rc = self._lib.sc_create(int(device), ctypes.byref(h))
    if rc != SC_OK:
      raise DeviceLibraryError("sc_create(device=%d) failed with %d" % (device, rc))
    self._h = h
    self.device = int(device)

  def close(self):
    if getattr(self, "_h", None):
      self._lib.sc_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass

  @property
  def raw(self):
    return self._h

  @property
  def lib(self):
    return self._lib

  def last_error(self) -> str:
    msg = self._lib.sc_last_error(self._h)
    return msg.decode("utf-8", "replace") if msg else ""

  def check(self, rc: int, invalid_exc: typing.Type[Exception] = ValueError):
    """Translate an sc_status into the exception the reference would raise."""
    if rc == SC_OK:
      return
    msg = self.last_error() or "status %d" % rc
    if rc == SC_ERR_INVALID:
      raise invalid_exc(msg)
    if rc == SC_ERR_NON_FINITE:
      raise np.linalg.LinAlgError(msg)   # what np.linalg.eig raises (a ValueError)
    if rc == SC_ERR_UNSUPPORTED:
      raise UnsupportedOnDeviceError(msg)
    if rc == SC_ERR_NOT_CONVERGED:
      raise EigenSolverNotConverged(msg)
    if rc == SC_ERR_OOM:
      raise MemoryError(msg)
    raise DeviceLibraryError(msg)


_default_handles = {}
_scope = threading.local()


class use_device:
  """`with use_device(k):` -- module-level helpers called inside (which take no device
  argument, like the reference's functions) run on device k in this thread.  A
  `SpectralClusterer(device=k)` wraps its predict() in it, so its pre-clustering, fallback
  and single-cluster helpers share its GPU and arena."""

  def __init__(self, device: typing.Optional[int]):
    self.device = device

  def __enter__(self):
    self.previous = getattr(_scope, "device", None)
    if self.device is not None:
      _scope.device = self.device
    return self

  def __exit__(self, *exc):
    _scope.device = self.previous


def default_handle(device: typing.Optional[int] = None) -> Handle:
  """Process-wide handle per device (the enclosing `use_device`, else
  SPECTRALCLUSTER_AMD_DEVICE, else LOCAL_RANK, else 0)."""
  if device is None:
    device = getattr(_scope, "device", None)
  if device is None:
    device = int(os.environ.get("SPECTRALCLUSTER_AMD_DEVICE",
                                os.environ.get("LOCAL_RANK", "0")))
    count = device_count()
    if count > 0:
      device %= count
  if device not in _default_handles:
    _default_handles[device] = Handle(device)
  return _default_handles[device]


_handle_pools = {}


def handle_pool(device: typing.Optional[int], size: int) -> typing.List[Handle]:
  """`size` independent handles (streams) on one device, created once per process.
  Handle 0 of the pool is the device's default handle."""
  first = default_handle(device)
  pool = _handle_pools.setdefault(first.device, [first])
  while len(pool) < size:
    pool.append(Handle(first.device))
  return pool[:size]
