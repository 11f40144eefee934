"""CPU: the shared library loads and exports every symbol of
include/spectralcluster_amd.h; host-side logic (eigengap, MT19937 stream, blur
weights, config flattening, AutoTune search, LPT sharding) matches the oracle.
No compute call is made (there is no GPU here)."""

import ctypes
import os
import re

import numpy as np
import pytest

import spectral_oracle as so
import spectralcluster_amd as sca
from spectralcluster_amd import _lib, multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, "include", "spectralcluster_amd.h")).read()
  declared = set(re.findall(r"^(?:int|const char\*)\s+(sc_[a-z0-9_]+)\s*\(", header,
                            flags=re.M))
  assert len(declared) >= 28
  lib = _lib.load()
  for name in sorted(declared):
    assert hasattr(lib, name), name
  assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
  assert lib.sc_abi_version() == _lib.SC_ABI_VERSION == 7


def test_graft_entry_build_runs():
  """The driver's "does it build" check: make + import + ABI/header agreement."""
  import importlib
  entry = importlib.import_module("__graft_entry__")
  entry.build()


def test_struct_layout_matches_header():
  # sizes implied by the C declaration (natural alignment)
  cfg = _lib.ScConfig()
  assert _lib.load().sc_config_default(cfg) == 0
  assert cfg.n_ops == 0 and cfg.p_percentile == 0.95 and cfg.soft_multiplier == 0.01
  assert cfg.threshold_type == 1 and cfg.symmetrize_type == 1 and cfg.max_iter == 300
  assert cfg.stop_eigenvalue == 1e-2 and cfg.blur_radius == 4
  np.testing.assert_allclose(list(cfg.blur_weights)[:9], so.gaussian_weights(1), rtol=1e-15)


def test_no_device_means_loud_failure():
  if _lib.device_count() > 0:
    pytest.skip("a GPU is visible")
  with pytest.raises(sca.DeviceLibraryError):
    sca.configs.icassp2018_clusterer.predict(np.zeros((4, 2)))
  with pytest.raises(sca.DeviceLibraryError):
    sca.utils.compute_affinity_matrix(np.ones((3, 2)))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
  monkeypatch.setenv("SPECTRALCLUSTER_AMD_LIB", str(tmp_path / "nope.so"))
  monkeypatch.setattr(_lib, "_lib", None)
  with pytest.raises(sca.DeviceLibraryError):
    _lib.load()


def test_eigengap_matches_oracle():
  rng = np.random.default_rng(0)
  for trial in range(200):
    m = int(rng.integers(1, 30))
    w = np.sort(rng.random(m) * (10.0 if trial % 2 else 1e-2))[::-1]
    for descend in (True, False):
      ww = w if descend else w[::-1].copy()
      for mc in (None, 3, 7, 40):
        for gt, og in ((sca.EigenGapType.Ratio, so.EIGENGAP_RATIO),
                       (sca.EigenGapType.NormalizedDiff, so.EIGENGAP_NORMALIZED_DIFF)):
          got = sca.utils.compute_number_of_clusters(ww, mc, 1e-2, gt, descend)
          want = so.eigengap(ww, mc, 1e-2, og, descend)
          assert got[0] == want[0] and got[1] == want[1]
  with pytest.raises(TypeError):
    sca.utils.compute_number_of_clusters(w, eigengap_type="Ratio")


def test_eigengap_reference_known_answers():
  # reference tests/utils_test.py:43-67
  w = np.array([1.0, 0.9, 0.8, 0.2, 0.1])
  k, d = sca.utils.compute_number_of_clusters(w)
  assert k == 3 and abs(d - 4.0) < 0.01
  k, d = sca.utils.compute_number_of_clusters(w, max_clusters=3, descend=False)
  assert k == 2 and abs(d - 0.88) < 0.01


def test_random_state_stream_and_choice():
  lib = _lib.load()
  out = np.empty(2000)
  assert lib.sc_random_state_doubles(0, 2000, _lib.as_double_p(out)) == 0
  assert np.array_equal(out, np.random.RandomState(0).random_sample(2000))
  for n in (1, 2, 7, 300, 8192, 100003):
    rs = np.random.RandomState(0)
    want = rs.choice(n, p=np.ones(n) / float(n))
    assert lib.sc_uniform_choice(n, out[0]) == want


def test_gaussian_weights_c_helper():
  lib = _lib.load()
  for sigma in (0.5, 1.0, 2.0, 3.3):
    radius = ctypes.c_int32(0)
    w = np.zeros(65)
    assert lib.sc_gaussian_weights(sigma, ctypes.byref(radius), _lib.as_double_p(w)) == 0
    ref = so.gaussian_weights(sigma)
    assert radius.value == (ref.size - 1) // 2
    np.testing.assert_allclose(w[:ref.size], ref, rtol=4e-16)
    # the Python facade hands the kernel scipy-identical weights (bit for bit)
    assert np.array_equal(sca.refinement.gaussian_weights(sigma), ref)
  radius = ctypes.c_int32(7)
  assert lib.sc_gaussian_weights(0.0, ctypes.byref(radius), _lib.as_double_p(w)) == 0
  assert radius.value == 0


def test_config_flattening():
  c = sca.SpectralClusterer(
      min_clusters=2, max_clusters=20, laplacian_type=sca.LaplacianType.GraphCut,
      refinement_options=sca.configs.icassp2018_refinement_options,
      eigengap_type=sca.EigenGapType.NormalizedDiff, row_wise_renorm=True, max_iter=17,
      stop_eigenvalue=0.05)
  cfg = c.build_config(p_percentile=0.7)
  assert cfg.n_ops == 6 and list(cfg.ops)[:6] == [1, 2, 3, 4, 5, 6]
  assert cfg.p_percentile == 0.7 and cfg.laplacian_type == 4 and cfg.eigengap_type == 2
  assert cfg.min_clusters == 2 and cfg.max_clusters == 20 and cfg.row_wise_renorm == 1
  assert cfg.max_iter == 17 and cfg.stop_eigenvalue == 0.05 and cfg.blur_radius == 4
  none = sca.SpectralClusterer().build_config()
  assert none.n_ops == 0 and none.laplacian_type == 0 and none.min_clusters == 0
  with pytest.raises(TypeError):
    sca.SpectralClusterer(laplacian_type=4).build_config()
  with pytest.raises(TypeError):
    sca.RefinementOptions(thresholding_type="RowMax",
                          refinement_sequence=[sca.RefinementName.RowWiseThreshold]
                          ).to_config(_lib.ScConfig())
  with pytest.raises(ValueError):
    sca.RefinementOptions(refinement_sequence=["Diffuse"]).to_config(_lib.ScConfig())
  with pytest.raises(ValueError):
    sca.RefinementOptions().get_refinement_operator("nope")


def test_public_surface_mirrors_reference():
  # every public name of the reference package (dir(spectralcluster) at 2024_10_08)
  for name in ("AutoTune", "AutoTuneProxy", "ConstraintMatrix", "ConstraintName",
               "ConstraintOptions", "Deflicker", "EigenGapType", "FallbackClustererType",
               "FallbackOptions", "ICASSP2018_REFINEMENT_SEQUENCE", "IntegrationType",
               "LaplacianType", "MultiStageClusterer", "NaiveClusterer", "RefinementName",
               "RefinementOptions", "SingleClusterCondition", "SpectralClusterer",
               "SymmetrizeType", "TURNTODIARIZE_REFINEMENT_SEQUENCE", "ThresholdType",
               "autotune", "configs", "constraint", "custom_distance_kmeans",
               "fallback_clusterer", "laplacian", "multi_stage_clusterer", "naive_clusterer",
               "refinement", "spectral_clusterer", "utils"):
    assert hasattr(sca, name), name
    assert name in sca.__all__, name
  assert [m.name for m in sca.RefinementName] == [
      "CropDiagonal", "GaussianBlur", "RowWiseThreshold", "Symmetrize", "Diffuse",
      "RowWiseNormalize"]
  assert [m.name for m in sca.LaplacianType] == ["Affinity", "Unnormalized",
                                                 "RandomWalk", "GraphCut"]
  c = sca.configs.icassp2018_clusterer
  assert (c.min_clusters, c.max_clusters, c.laplacian_type, c.custom_dist) == (
      2, 7, None, "cosine")
  opts = sca.RefinementOptions()
  assert (opts.gaussian_blur_sigma, opts.p_percentile, opts.thresholding_soft_multiplier,
          opts.refinement_sequence) == (1, 0.95, 0.01, None)


def test_constraint_host_side():
  from conftest import golden
  g = golden("constraint_ops_n40.npz")
  scores = list(g["turn_scores"])
  np.testing.assert_array_equal(sca.ConstraintMatrix(scores, 1).compute_diagonals(),
                                g["turn_matrix"])
  np.testing.assert_array_equal(sca.ConstraintMatrix(scores, 3).compute_diagonals(),
                                g["turn_matrix_t3"])
  assert sca.ConstraintMatrix([], 1).compute_diagonals().shape == (0, 0)
  assert sca.ConstraintMatrix([0.0], 1).compute_diagonals().shape == (1, 1)
  with pytest.raises(ValueError):
    sca.ConstraintMatrix([0, -0.5])
  # names / members of the reference enums (constraint.py:10-22)
  assert [m.name for m in sca.ConstraintName] == ["AffinityIntegration",
                                                  "ConstraintPropagation"]
  assert [m.name for m in sca.IntegrationType] == ["Max", "Average"]
  # flattening into sc_config
  c = sca.configs.turntodiarize_clusterer
  cfg = c.build_config()
  assert (cfg.constraint_name, cfg.constraint_before_refinement, cfg.constraint_alpha) == (
      2, 1, 0.4)
  assert (c.min_clusters, c.max_clusters, c.laplacian_type, c.row_wise_renorm) == (
      2, 7, sca.LaplacianType.GraphCut, True)
  opts = sca.ConstraintOptions(sca.ConstraintName.AffinityIntegration, False,
                               sca.IntegrationType.Average)
  cfg = sca.SpectralClusterer(constraint_options=opts).build_config()
  assert (cfg.constraint_name, cfg.constraint_before_refinement, cfg.integration_type) == (
      1, 0, 2)
  # integration_type=None only fails when the operator is used (reference :117-118)
  broken = sca.ConstraintOptions(sca.ConstraintName.AffinityIntegration, False)
  with pytest.raises(ValueError):
    sca.SpectralClusterer(constraint_options=broken).build_config()
  # check_input messages (constraint.py:54-76) fire before any device work
  op = sca.constraint.ConstraintPropagation()
  with pytest.raises(ValueError, match="same shape"):
    op.adjust_affinity(np.zeros((3, 3)), np.zeros((2, 2)))
  with pytest.raises(ValueError, match="constraint matrix must be a square"):
    op.adjust_affinity(np.zeros((3, 3)), np.zeros((3, 2)))
  with pytest.raises(ValueError, match="affinity must be 2-dimensional"):
    op.adjust_affinity(np.zeros(3), np.zeros((3, 3)))


def test_match_labels_reference_known_answers():
  # reference tests/multi_stage_clusterer_test.py:11-60
  from spectralcluster_amd import multi_stage_clusterer as ms
  cases = [([1, 0], [0], [0, 1]),
           ([0, 1, 2, 3, 4, 5], [0, 0, 0, 1, 2], [0, 3, 4, 1, 2, 5]),
           ([0, 0, 0, 1, 1, 1, 2, 2], [0, 0, 1, 2, 2, 3, 4], [0, 0, 0, 2, 2, 2, 4, 4]),
           ([1, 1, 1, 0, 0, 1], [0, 0, 0, 1, 1], [0, 0, 0, 1, 1, 0]),
           ([1, 1, 1, 0, 0, 2], [0, 0, 0, 1, 1], [0, 0, 0, 1, 1, 2]),
           ([0, 1, 1, 0, 0, 2], [0, 0, 0, 1, 1], [1, 0, 0, 1, 1, 2]),
           ([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5], [0, 0, 3, 3, 1, 1, 4, 4, 5, 5, 2],
            [0, 0, 3, 3, 1, 1, 4, 4, 5, 5, 2, 2])]
  for current, previous, expected in cases:
    np.testing.assert_equal(ms.match_labels(np.array(current), np.array(previous)), expected)
  with pytest.raises(ValueError):
    ms.match_labels(np.array([0, 1]), np.array([0, 1]))


def test_linear_sum_assignment_matches_scipy():
  """The reference calls scipy's solver; ours must return the same assignment, ties
  included (it drives the label numbering of the Hungarian deflicker)."""
  from scipy.optimize import linear_sum_assignment as scipy_lsa
  from spectralcluster_amd import multi_stage_clusterer as ms
  rng = np.random.default_rng(0)
  for _ in range(1500):
    shape = rng.integers(1, 8, size=2)
    cost = rng.integers(0, 4, size=shape)
    maximize = bool(rng.integers(0, 2))
    want = scipy_lsa(cost, maximize=maximize)
    got = ms.linear_sum_assignment(cost, maximize=maximize)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
  cost = rng.random((5, 9))
  np.testing.assert_array_equal(ms.linear_sum_assignment(cost)[1], scipy_lsa(cost)[1])


def test_fallback_surface_mirrors_reference():
  fo = sca.FallbackOptions()
  assert (fo.spectral_min_embeddings, fo.single_cluster_condition,
          fo.single_cluster_affinity_threshold, fo.single_cluster_affinity_diagonal_offset,
          fo.fallback_clusterer_type, fo.agglomerative_threshold, fo.naive_threshold,
          fo.naive_adaptation_threshold) == (
              1, sca.SingleClusterCondition.AffinityGmmBic, 0.75, 1,
              sca.FallbackClustererType.Naive, 0.5, 0.5, None)
  assert [m.name for m in sca.SingleClusterCondition] == [
      "AffinityGmmBic", "AllAffinity", "NeighborAffinity", "AffinityStd", "FallbackClusterer"]
  assert [m.name for m in sca.Deflicker] == ["NoDeflicker", "OrderBased", "Hungarian"]
  # a clusterer without explicit options gets the defaults (reference :93-96)
  assert sca.SpectralClusterer().fallback_options == fo
  main = sca.SpectralClusterer()
  stage = sca.MultiStageClusterer(main, fallback_threshold=0.3, L=11, U1=22, U2=33)
  assert (main.fallback_options.spectral_min_embeddings,
          main.fallback_options.agglomerative_threshold,
          main.fallback_options.single_cluster_condition,
          main.fallback_options.fallback_clusterer_type, stage.U1, stage.U2) == (
              11, 0.3, sca.SingleClusterCondition.FallbackClusterer,
              sca.FallbackClustererType.Agglomerative, 22, 33)
  with pytest.raises(ValueError):
    sca.naive_clusterer.NaiveClusterer(0.5, 0.4)


def test_enforce_ordered_labels():
  # reference tests/utils_test.py (TestEnforceOrderedLabels)
  got = sca.utils.enforce_ordered_labels(np.array([9, 9, 1, 1, 9, 5]))
  assert np.array_equal(got, [0, 0, 1, 1, 0, 2])
  assert np.array_equal(so.ordered_labels(np.array([9, 9, 1, 1, 9, 5])), got)


def test_autotune_ranges_and_search():
  # reference tests/autotune_test.py:18-38
  t = sca.AutoTune(p_percentile_min=0.9, p_percentile_max=0.95, init_search_step=0.01,
                   search_level=1)
  np.testing.assert_allclose(t.get_percentile_range(), [0.9, 0.9125, 0.925, 0.9375, 0.95],
                             atol=0.01)
  t = sca.AutoTune(0.40, 0.95, 0.05, 1)
  assert np.array_equal(t.get_percentile_range(), so.autotune_range(0.40, 0.95, 0.05))
  with pytest.raises(TypeError):
    sca.AutoTune(proxy="x")
  # synthetic ratio curve: the search must behave like the oracle's restatement
  def curve(p):
    return (p - 0.73) ** 2 + 1.0

  for level in (1, 2, 3):
    t = sca.AutoTune(0.40, 0.95, 0.05, level)
    calls = []

    def fn(p):
      calls.append(p)
      return curve(p), "vec%g" % p, 3

    vec, k, best = t.tune(fn)
    # oracle restatement with the same curve
    grid = so.autotune_range(0.40, 0.95, 0.05)
    seen, step, pmin, pmax = {}, 0.05, 0.40, 0.95
    bestp, besti = None, None
    for _ in range(level):
      lowest = np.inf
      for i, p in enumerate(grid):
        if p in seen:
          continue
        seen[p] = curve(p)
        if seen[p] < lowest:
          lowest, bestp, besti = seen[p], p, i
      reach = max(2, len(grid) // 8)
      lo, hi = max(0, besti - reach), min(len(grid) - 1, besti + reach)
      pmin, pmax, step = grid[lo], grid[hi], step / 2
      grid = so.autotune_range(pmin, pmax, step)
    assert best == bestp and vec == "vec%g" % bestp and k == 3
    assert calls == list(seen)
  # ratio proxies (reference spectral_clusterer.py:281-286)
  assert sca.AutoTune().ratio(0.75, 2.0) == np.sqrt(0.25) / 2.0
  assert sca.AutoTune(proxy=sca.AutoTuneProxy.PercentileOverNME).ratio(0.75, 2.0) == 0.125


def test_lpt_assignment():
  rng = np.random.default_rng(512)
  sizes = rng.integers(300, 3001, 512).tolist()
  owned = multigpu.lpt_assignment(sizes, 8)
  flat = sorted(i for o in owned for i in o)
  assert flat == list(range(512))
  loads = [sum(multigpu.cost_model(sizes[i]) for i in o) for o in owned]
  assert max(loads) / (sum(loads) / 8) < 1.02   # well balanced: >= 7.8x on 8 GPUs
  assert multigpu.lpt_assignment([5, 5, 5], 2) == [[0, 2], [1]]
  assert multigpu.first_strict_minimum(np.array([3.0, 1.0, 1.0, 2.0])) == 1


def test_predict_validation_happens_before_the_device():
  """Argument errors of predict() (reference spectral_clusterer.py:222-246) are raised by the
  host mirror before any device call -- so they are checkable without a GPU."""
  import spectralcluster_amd as sca
  with pytest.raises(TypeError, match="embeddings must be a numpy array"):
    sca.SpectralClusterer().predict([[1.0, 2.0]])
  with pytest.raises(ValueError, match="embeddings must be 2-dimensional"):
    sca.SpectralClusterer().predict(np.zeros(5))
  with pytest.raises(ValueError, match="embeddings must be 2-dimensional"):
    sca.SpectralClusterer().predict(np.zeros((2, 2, 2)))
  with pytest.raises(RuntimeError, match="Cannot handle constraint_matrix"):
    sca.SpectralClusterer(max_spectral_size=10).predict(np.zeros((20, 2)), np.zeros((20, 20)))
  for kwargs in (dict(max_spectral_size=1), dict(max_spectral_size=5, max_clusters=5),
                 dict(max_spectral_size=4, min_clusters=4)):
    with pytest.raises(ValueError, match="max_spectral_size should be a relatively big"):
      sca.SpectralClusterer(**kwargs).predict(np.zeros((20, 2)))


def test_predict_batch_falls_back_to_predict_for_custom_functions(monkeypatch):
  """sc_predict_batch runs the device cosine affinity + device k-means; a clusterer built
  with a user affinity_function / post_eigen_cluster_function (or a k-means metric that is
  not on the device) must not be routed there silently."""
  seen = []

  def fake_predict(self, u, constraint_matrix=None):
    seen.append(u.shape[0])
    return np.zeros(u.shape[0], dtype=np.int64)

  monkeypatch.setattr(sca.SpectralClusterer, "predict", fake_predict)
  utts = [np.ones((5, 3)), np.ones((7, 3))]
  for kwargs in ({"affinity_function": lambda x: np.ones((x.shape[0], x.shape[0]))},
                 {"post_eigen_cluster_function": lambda **kw: np.zeros(3)}):
    del seen[:]
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=4, **kwargs)
    out = c.predict_batch(utts, streams=2)
    assert seen == [5, 7] and [o.shape[0] for o in out] == [5, 7]
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=4, custom_dist="mahalanobis")
  with pytest.raises(sca.UnsupportedOnDeviceError):
    c.predict_batch(utts)


def test_predict_batch_mode_selection(monkeypatch):
  """predict_batch(utts) is the grouped batch (16 utterances per launch); `streams` alone
  selects the multi-stream form; an explicit `group` wins."""
  calls = []

  class FakeLib:
    def sc_clear_constraint(self, raw):
      return 0

    def sc_predict_batch_grouped(self, raw, xp, ns, d, count, cfg, lp, diags, group):
      calls.append(("grouped", group, count))
      return 0

    def sc_predict_batch_streams(self, raw, xp, ns, d, count, cfg, lp, diags, streams):
      calls.append(("streams", streams, count))
      return 0

  class FakeHandle:
    lib, raw = FakeLib(), None

    def check(self, rc, *a):
      assert rc == 0

  c = sca.SpectralClusterer(min_clusters=2, max_clusters=4)
  monkeypatch.setattr(c, "_handle", lambda: FakeHandle())
  monkeypatch.setattr(c, "build_config", lambda: None)
  utts = [np.ones((5, 3)), np.ones((7, 3)), np.ones((4, 3))]
  c.predict_batch(utts)
  c.predict_batch(utts, streams=4)
  c.predict_batch(utts, streams=1)
  c.predict_batch(utts, group=8)
  c.predict_batch(utts, streams=2, group=3)
  assert calls == [("grouped", 16, 3), ("streams", 4, 3), ("streams", 1, 3), ("grouped", 8, 3),
                   ("grouped", 3, 3)]


def test_autotune_hands_whole_levels_to_the_sweep(monkeypatch):
  """predict() with AutoTune: every search level is ONE _eig_sweep call with the level's new
  p values (reference autotune.py:98-111 evaluates them one by one), the winner is evaluated
  once more for its eigenvectors, and refinement_options.p_percentile is left at the LAST
  evaluated value like the reference's closure does (spectral_clusterer.py:277)."""
  levels, singles = [], []

  class Diag:
    def __init__(self, p):
      self.max_delta = 1.0 + 10.0 * (1.0 - abs(p - 0.7))  # proxy minimal at p = 0.7
      self.n_clusters_raw = 3
      self.n_clusters = 3

  c = sca.SpectralClusterer(
      min_clusters=2, max_clusters=5,
      refinement_options=sca.RefinementOptions(
          p_percentile=0.95, refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
      autotune=sca.AutoTune(p_percentile_min=0.5, p_percentile_max=0.9, init_search_step=0.1,
                            search_level=2, proxy=sca.AutoTuneProxy.PercentileOverNME))

  class FakeLib:
    def sc_cluster(self, raw, cfg, k, labels, diag):
      return 0

  class FakeHandle:
    lib, raw = FakeLib(), None

    def check(self, rc, *a):
      assert rc == 0

  def fake_sweep(handle, ps):
    levels.append(list(ps))
    return [Diag(p) for p in ps]

  def fake_single(handle, p=None):
    singles.append(p)
    c.last_diag = Diag(p)
    return c.last_diag

  monkeypatch.setattr(c, "_handle", lambda: FakeHandle())
  monkeypatch.setattr(c, "_set_constraint", lambda *a, **k: False)
  monkeypatch.setattr(c, "_upload", lambda *a, **k: None)
  monkeypatch.setattr(c, "_eig_sweep", fake_sweep)
  monkeypatch.setattr(c, "_eig_resident", fake_single)
  c.predict(np.ones((12, 3)))
  assert len(levels) == 2 and len(levels[0]) == 4
  assert not set(levels[0]) & set(levels[1])  # a level only evaluates what is new
  assert len(singles) == 1
  # (the reference restarts its running minimum at every level, autotune.py:99: the winner is
  #  the best of the LAST level's new values)
  best = min(levels[1], key=lambda p: (1 - p) / Diag(p).max_delta)
  assert singles[0] == best
  assert c.refinement_options.p_percentile == levels[1][-1]


def test_comm_header_and_id_file(monkeypatch, tmp_path):
  """Rendezvous plumbing of RcclComm.from_env that needs no GPU: the id file name is
  unique per launch (MASTER_PORT + torchrun's run id + parent pid), lives in a private
  per-user directory, and can be overridden."""
  monkeypatch.delenv("SC_COMM_ID_FILE", raising=False)
  monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
  monkeypatch.setenv("MASTER_PORT", "29517")
  monkeypatch.setenv("TMPDIR", str(tmp_path))
  path = multigpu._id_file()
  base = os.path.join(str(tmp_path), "sc_comm_%d" % os.getuid())
  assert path == os.path.join(base, "29517_x_%d.id" % os.getppid())
  assert (os.stat(base).st_mode & 0o777) == 0o700
  monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run/../42")
  assert multigpu._id_file() == os.path.join(base, "29517_run42_%d.id" % os.getppid())
  os.chmod(base, 0o755)  # a directory others could write into is refused
  with pytest.raises(PermissionError):
    multigpu._id_file()
  os.chmod(base, 0o700)
  monkeypatch.setenv("SC_COMM_ID_FILE", str(tmp_path / "x.id"))
  assert multigpu._id_file() == str(tmp_path / "x.id")
  assert _lib.load().sc_comm_rank(None) == -1 and _lib.load().sc_comm_size(None) == 0


def test_host_rayleigh_ritz_solver_vs_numpy():
  """The host routine behind the small Rayleigh-Ritz problems (tred2 / tql2)."""
  lib = _lib.load()
  rng = np.random.default_rng(0)
  for m in (1, 2, 3, 8, 24, 32, 48, 64, 96, 128):
    a = rng.standard_normal((m, m))
    a = 0.5 * (a + a.T)
    if m == 24:  # the shape of a projected Laplacian: a few informative values + a tight bulk
      q, _ = np.linalg.qr(rng.standard_normal((m, m)))
      lam = np.concatenate([[0.0, -0.07, -0.073, -0.075], -1 + 1e-6 * rng.standard_normal(m - 4)])
      a = (q * lam) @ q.T
      a = 0.5 * (a + a.T)
    a = np.ascontiguousarray(a)
    w = np.empty(m)
    v = np.empty((m, m))
    assert lib.sc_host_symmetric_eig(_lib.as_double_p(a), m, _lib.as_double_p(w),
                                     _lib.as_double_p(v)) == 0
    scale = max(1.0, np.abs(a).max())
    np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(a), rtol=0, atol=1e-13 * scale)
    assert np.abs(a @ v - v * w).max() < 1e-13 * scale * m
    assert np.abs(v.T @ v - np.eye(m)).max() < 1e-13 * m


def test_host_partial_symmetric_eig_vs_numpy():
  """The Rayleigh-Ritz solve of bases above 64 vectors: all eigenvalues, leading vectors."""
  lib = _lib.load()
  rng = np.random.default_rng(3)
  for m, need in ((1, 1), (2, 2), (5, 3), (24, 24), (72, 32), (96, 40), (128, 48), (128, 128)):
    if m >= 72:  # a dense bulk next to a few separated values, like a hard Ritz problem
      q, _ = np.linalg.qr(rng.standard_normal((m, m)))
      lam = np.r_[np.linspace(-1.0, -0.98, m - 3), [-0.5, 0.0, -0.97]]
      a = (q * lam) @ q.T
    else:
      a = rng.standard_normal((m, m))
    a = np.ascontiguousarray(0.5 * (a + a.T))
    w = np.empty(m)
    v = np.empty((m, need))
    assert lib.sc_host_symmetric_eig_partial(_lib.as_double_p(a), m, need, _lib.as_double_p(w),
                                             _lib.as_double_p(v)) == 0
    scale = max(1.0, np.abs(a).max())
    np.testing.assert_allclose(w, np.linalg.eigvalsh(a)[::-1], rtol=0, atol=1e-13 * scale * m)
    assert np.abs(a @ v - v * w[:need]).max() < 1e-12 * scale * m
    assert np.abs(v.T @ v - np.eye(need)).max() < 1e-12 * m


@pytest.mark.parametrize("entry", ["sc_host_general_eig", "sc_host_general_eig_fast"])
def test_host_general_eig_vs_numpy(entry):
  """The Rayleigh-Ritz solves of block Arnoldi (general eigen path, projected problems of order
  up to 128): eigenvalues sorted by real part, unit-norm eigenvectors.  `_fast` (round 6, what
  the checks call: Hessenberg + real double-shift QR + inverse iteration for the leading vectors)
  and the complex-Schur solver that is its fallback, both against numpy."""
  lib = _lib.load()
  solve = getattr(lib, entry)
  rng = np.random.default_rng(11)
  for m, nvec in ((1, 1), (2, 2), (7, 7), (24, 10), (64, 64), (100, 50), (128, 128)):
    for kind in ("random", "nearly symmetric", "block Hessenberg"):
      a = rng.standard_normal((m, m))
      if kind == "nearly symmetric":
        a = a + a.T + 0.01 * rng.standard_normal((m, m))
      if kind == "block Hessenberg":  # what Q^T Op Q of a block Arnoldi looks like
        a = np.triu(a, -8)
      a = np.ascontiguousarray(a)
      wr, wi = np.empty(m), np.empty(m)
      vr, vi = np.empty((m, nvec)), np.empty((m, nvec))
      assert solve(_lib.as_double_p(a), m, nvec, _lib.as_double_p(wr), _lib.as_double_p(wi),
                   _lib.as_double_p(vr), _lib.as_double_p(vi)) == 0
      assert np.all(np.diff(wr) <= 0.0)  # sorted by real part, descending
      w = wr + 1j * wi
      ref = np.linalg.eigvals(a)
      scale = np.abs(ref).max()
      dist = np.abs(w[:, None] - ref[None, :])
      assert max(dist.min(axis=1).max(), dist.min(axis=0).max()) < 1e-10 * scale * m, (m, kind)
      v = vr + 1j * vi
      assert np.abs(a @ v - v * w[None, :nvec]).max() < 1e-12 * scale * m, (m, kind)
      assert np.abs(np.linalg.norm(v, axis=0) - 1.0).max() < 1e-13 * m


def _gehd2(a):
  """LAPACK dgehd2 in NumPy, in the storage the device reduction (hessenberg.hip) leaves: the
  Hessenberg matrix on and above the subdiagonal, reflector k (v[k + 1] = 1 implied) below the
  subdiagonal of column k, tau."""
  a = a.copy()
  n = a.shape[0]
  tau = np.zeros(max(n - 2, 1))
  for k in range(n - 2):
    x = a[k + 1:, k].copy()
    alpha, xnorm = x[0], np.linalg.norm(x[1:])
    if xnorm == 0.0:
      continue
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau[k] = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    a[:, k + 1:] -= tau[k] * np.outer(a[:, k + 1:] @ v, v)
    a[k + 1:, k + 1:] -= tau[k] * np.outer(v, v @ a[k + 1:, k + 1:])
    a[k + 1, k] = beta
    a[k + 2:, k] = v[1:]
  return np.ascontiguousarray(a), tau


def _host_hessenberg_eig(a, pick):
  lib = _lib.load()
  n = a.shape[0]
  packed, tau = _gehd2(a)
  count = len(pick)
  wr, wi = np.empty(n), np.empty(n)
  vr, vi = np.empty((n, max(count, 1))), np.empty((n, max(count, 1)))
  res = np.zeros(1)
  pk = np.asarray(pick, dtype=np.int32)
  rc = lib.sc_host_hessenberg_eig(
      _lib.as_double_p(packed), _lib.as_double_p(tau), n, count,
      pk.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _lib.as_double_p(wr), _lib.as_double_p(wi),
      _lib.as_double_p(vr), _lib.as_double_p(vi), _lib.as_double_p(res))
  return rc, wr + 1j * wi, (vr + 1j * vi)[:, :count], float(res[0])


def test_host_hessenberg_eigenvalues_vs_numpy():
  """The QR iteration behind eig_path 7 (every eigenvalue of a general matrix of order > 64):
  random, graph-Laplacian, defective-looking and block-diagonal inputs against np.linalg.eig."""
  rng = np.random.default_rng(70)
  cases = []
  for n in (3, 4, 17, 65, 200, 421):
    cases.append(("randn", rng.standard_normal((n, n))))
  x = so.blobs(300, 24, 5, seed=9)
  a = so.row_wise_threshold(so.affinity(x), p_percentile=0.9,
                             threshold_type=so.THRESHOLD_PERCENTILE)
  cases.append(("thresholded affinity", a))
  cases.append(("graph-cut laplacian", so.laplacian(a, so.LAPLACIAN_GRAPH_CUT)))
  cases.append(("random-walk laplacian", so.laplacian(a, so.LAPLACIAN_RANDOM_WALK)))
  blk = np.zeros((120, 120))
  for b in range(4):  # four disconnected components: eigenvalue 0 four times
    blk[30 * b:30 * b + 30, 30 * b:30 * b + 30] = rng.uniform(0.5, 1.0, (30, 30))
  cases.append(("disconnected", np.diag(blk.sum(1)) - blk))
  jordan = np.diag(np.full(40, 2.0)) + np.diag(np.full(39, 1e-3), 1)
  q, _ = np.linalg.qr(rng.standard_normal((40, 40)))
  cases.append(("nearly defective", q @ jordan @ q.T))
  cases.append(("already hessenberg", np.triu(rng.standard_normal((50, 50)), -1)))
  cases.append(("zero", np.zeros((10, 10))))
  for name, m in cases:
    rc, w, _, _ = _host_hessenberg_eig(m, [])
    assert rc == 0, name
    ref = np.linalg.eigvals(m)
    scale = max(np.abs(ref).max(), 1e-300)
    dist = np.abs(w[:, None] - ref[None, :])
    # (a 40 x 40 Jordan-like block with coupling 1e-3: rounding moves its eigenvalues by
    #  ~1e-3 * eps^(1/40) ~ 4e-4 in ANY backward-stable solver, numpy's included)
    tol = (1e-4 if name == "nearly defective" else 1e-11) * scale * max(m.shape[0], 10)
    assert max(dist.min(axis=1).max(), dist.min(axis=0).max()) <= tol, name
    # conjugate pairs sit next to each other, positive imaginary part first
    for i in np.nonzero(w.imag > 0)[0]:
      assert w[i + 1] == np.conj(w[i]), name


def test_host_hessenberg_eigenvectors_are_eigenvectors_of_the_original_matrix():
  """Inverse iteration on the Hessenberg form + back-transform through the reflectors: residual
  of A x = lambda x on the ORIGINAL matrix, independence inside clusters of equal eigenvalues,
  conjugate pairs."""
  rng = np.random.default_rng(71)
  x = so.blobs(260, 24, 6, seed=4)
  a = so.row_wise_threshold(so.affinity(x), p_percentile=0.9,
                             threshold_type=so.THRESHOLD_PERCENTILE)
  lap = so.laplacian(a, so.LAPLACIAN_GRAPH_CUT)
  blk = np.zeros((120, 120))
  for b in range(4):
    blk[30 * b:30 * b + 30, 30 * b:30 * b + 30] = rng.uniform(0.5, 1.0, (30, 30))
  q = rng.standard_normal((60, 60))
  rep = q @ np.diag([2.0] * 5 + [1.0] * 5 + list(rng.uniform(-1, 0.5, 50))) @ np.linalg.inv(q)
  for name, m, count in (("randn", rng.standard_normal((150, 150)), 40),
                         ("laplacian", -lap, 80), ("affinity", a, 80),
                         ("disconnected", blk - np.diag(blk.sum(1)), 8), ("repeated", rep, 12)):
    rc, w, _, _ = _host_hessenberg_eig(m, [])
    assert rc == 0
    pick = np.argsort(-w.real, kind="stable")[:count]
    rc, w2, v, res = _host_hessenberg_eig(m, pick)
    assert rc == 0 and np.array_equal(w, w2)
    lam = w[pick]
    scale = np.abs(m).max()
    r = np.linalg.norm(m @ v - v * lam[None, :], axis=0) / np.linalg.norm(v, axis=0)
    assert r.max() < 1e-10 * scale * m.shape[0], (name, r.max())
    assert res < 1e-10
    unit = v / np.linalg.norm(v, axis=0)
    assert np.linalg.svd(unit, compute_uv=False).min() > 1e-4, name  # an independent set
    for i in range(count - 1):
      if lam[i].imag > 0 and lam[i + 1] == np.conj(lam[i]):
        assert np.array_equal(v[:, i + 1], np.conj(v[:, i])), name


def test_host_hessenberg_vectors_residual_gate_on_a_defective_eigenvalue():
  """inverse iteration accepted whatever six solves left.  A DEFECTIVE eigenvalue
  (exact Jordan block in an already-triangular matrix: the QR iteration returns 2.0 twice) has ONE
  eigenvector; the second member of its cluster, kept orthogonal to the first, is rounding noise.
  The gate must reject that iterate and fall back to what LAPACK returns (the eigenvector again):
  every returned column has a residual at rounding level."""
  m = np.zeros((40, 40))
  m[0, 0] = m[1, 1] = 2.0
  m[0, 1] = 1.0
  m[2:, 2:] = np.triu(np.random.default_rng(5).uniform(-1.0, 1.0, (38, 38)))
  m[0:2, 2:] = 0.3
  rc, w, _, _ = _host_hessenberg_eig(m, [])
  assert rc == 0
  pick = np.argsort(-w.real, kind="stable")[:4]
  assert np.all(w[pick[:2]] == 2.0)
  rc, _, v, res = _host_hessenberg_eig(m, pick)
  assert rc == 0 and res < 1e-8 * np.sqrt(40)
  r = np.linalg.norm(m @ v - v * w[pick][None, :], axis=0) / np.linalg.norm(v, axis=0)
  assert r.max() < 1e-7 * np.abs(m).max() * 40, r


def test_cost_model_matches_its_calibration_record():
  """multigpu.cost_model against the measured per-utterance times it was fitted on
  (profiles/r30_cost_fit.txt, written by tests/probes/cost_model_fit.py on the GPU box):
  the code and the record the docs cite must not drift apart."""
  import re
  path = os.path.join(ROOT, "profiles", "r30_cost_fit.txt")
  rows = [(int(m.group(1)), float(m.group(2)))
          for m in (re.match(r"n=(\d+): ([0-9.]+) us", line) for line in open(path)) if m]
  assert len(rows) >= 12
  for n, measured in rows:
    # (the matrix-free members' cost depends on how many tiles their skip list keeps: 8.2 % is
    #  the fit's own worst case on that branch, 2 % below it)
    tol = 0.09 if n >= 1536 else 0.03
    assert abs(multigpu.cost_model(n) / measured - 1.0) < tol, (n, measured)


def test_value_error_bound_holds_for_rayleigh_ritz():
  """The stopping rule's bound (residual, or Kato-Temple where the neighbours fence a Ritz value
  off) against the TRUE error of Rayleigh-Ritz values: random subspaces of matrices with
  well-separated, clustered and repeated leading eigenvalues.  The bound must hold, never
  exceed the residual, and be much smaller than it where the spectrum is separated."""
  lib = _lib.load()
  rng = np.random.default_rng(21)
  n, m = 300, 40
  shrunk = 0
  for kind in ("separated", "clustered", "repeated"):
    for trial in range(6):
      lead = {"separated": np.linspace(3.0, 2.0, 12),
              "clustered": 2.0 + 1e-4 * np.arange(12),
              "repeated": np.repeat([3.0, 2.5, 2.0], 4)}[kind]
      lam = np.concatenate([lead, rng.uniform(-1.0, 1.0, n - lead.size)])
      q, _ = np.linalg.qr(rng.standard_normal((n, n)))
      a = (q * lam) @ q.T
      a = 0.5 * (a + a.T)
      # a subspace that holds the leading eigenvectors up to a perturbation of size eps
      eps = 10.0 ** rng.uniform(-7, -2)
      basis = np.concatenate([q[:, :lead.size] + eps * rng.standard_normal((n, lead.size)),
                              rng.standard_normal((n, m - lead.size))], axis=1)
      v, _ = np.linalg.qr(basis)
      t = v.T @ a @ v
      theta, y = np.linalg.eigh(0.5 * (t + t.T))
      theta, y = theta[::-1].copy(), y[:, ::-1]
      x = v @ y
      resid = np.linalg.norm(a @ x - x * theta, axis=0)
      true = np.sort(lam)[::-1]
      for i in range(lead.size):
        b = ctypes.c_double()
        assert lib.sc_host_value_error_bound(_lib.as_double_p(theta), _lib.as_double_p(resid), m,
                                             i, ctypes.byref(b)) == 0
        err = abs(true[i] - theta[i])  # (Cauchy interlacing: the i-th Ritz value belongs to the i-th eigenvalue)
        assert b.value <= resid[i] * (1 + 1e-15)
        assert err <= b.value * (1 + 1e-9) + 1e-13, (kind, trial, i, err, b.value, resid[i])
        if kind == "separated" and b.value < 1e-2 * resid[i]:
          shrunk += 1
  assert shrunk >= 12  # the quadratic bound is what makes separated spectra cheap


def test_host_tridiag_eigvectors_vs_scipy():
  """The host step of the dense landing pad (inverse iteration on the tridiagonal form,
  LAPACK dstein's method) against scipy's eigh_tridiagonal, including clustered spectra."""
  from scipy.linalg import eigh_tridiagonal
  lib = _lib.load()
  rng = np.random.default_rng(5)
  cases = []
  for n in (2, 3, 50, 1000, 3001):
    cases.append((rng.standard_normal(n), rng.standard_normal(n - 1), min(n, 20)))
  cases.append((np.abs(np.arange(1001) - 500.0), np.ones(1000), 20))      # Wilkinson pairs
  cases.append((np.ones(500), np.full(499, 1e-9), 20))                    # one tight cluster
  cases.append((np.r_[np.ones(10), np.zeros(90)], np.zeros(99), 20))      # exact repeats
  cases.append((np.full(2048, 2.0), np.full(2047, -1.0), 64))             # 1-D Laplacian edge
  for d, e, k in cases:
    n = d.size
    d = np.ascontiguousarray(d)
    e = np.ascontiguousarray(e)
    w = eigh_tridiagonal(d, e, eigvals_only=True)
    lam = np.ascontiguousarray(w[::-1][:k])  # the driver asks for the largest ones first
    out = np.empty((n, k))
    assert lib.sc_host_tridiag_eigvectors(_lib.as_double_p(d), _lib.as_double_p(e), n,
                                          _lib.as_double_p(lam), k, _lib.as_double_p(out)) == 0
    tv = d[:, None] * out
    tv[:-1] += e[:, None] * out[1:]
    tv[1:] += e[:, None] * out[:-1]
    scale = max(1.0, np.abs(d).max() + 2 * np.abs(e).max())
    assert np.abs(tv - out * lam).max() < 1e-12 * scale
    assert np.abs(out.T @ out - np.eye(k)).max() < 1e-12
  one = np.empty((1, 1))
  assert lib.sc_host_tridiag_eigvectors(_lib.as_double_p(np.array([3.0])),
                                        _lib.as_double_p(np.zeros(1)), 1,
                                        _lib.as_double_p(np.array([3.0])), 1,
                                        _lib.as_double_p(one)) == 0 and one[0, 0] == 1.0


def test_use_device_scope_is_thread_local_and_nested():
  assert getattr(_lib._scope, "device", None) is None
  with _lib.use_device(3):
    assert _lib._scope.device == 3
    with _lib.use_device(None):      # None does not override
      assert _lib._scope.device == 3
    with _lib.use_device(1):
      assert _lib._scope.device == 1
    assert _lib._scope.device == 3
  assert _lib._scope.device is None


def test_integration_md_stub_mirrors_and_config_flattening():
  """INTEGRATION.md section 2 (the reference-side ctypes binding) without a GPU: its code
  blocks execute against the built library up to handle creation, its struct mirrors have
  the sizes the library reports, and its `_to_config` flattens a clusterer like
  SpectralClusterer.build_config does.  (The GPU twin runs predict() through it:
  tests/test_gpu_integration_stub.py.)"""
  import re
  import ctypes
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  text = open(os.path.join(root, "INTEGRATION.md")).read()
  section = text[text.index("## 2."):text.index("## 3.")]
  blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
  assert len(blocks) == 2
  so_path = os.path.join(root, "spectralcluster_amd", "csrc", "libspectralcluster_amd.so")
  code = "\n".join(blocks).replace('ctypes.CDLL("libspectralcluster_amd.so")',
                                   "ctypes.CDLL(%r)" % so_path)
  create = "assert _lib.sc_create(0, ctypes.byref(_handle)) == 0"
  assert create in code
  ns = {}
  exec(compile(code.replace(create, "pass"), "INTEGRATION.md#2", "exec"), ns)  # pylint: disable=exec-used
  assert ctypes.sizeof(ns["_ScConfig"]) == ctypes.sizeof(_lib.ScConfig)
  assert ctypes.sizeof(ns["_ScDiag"]) == ctypes.sizeof(_lib.ScDiag)
  import spectralcluster_amd as sca
  for c in (sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                                  refinement_options=sca.configs.icassp2018_refinement_options,
                                  laplacian_type=sca.LaplacianType.GraphCut),
            sca.configs.turntodiarize_clusterer):
    got, want = ns["_to_config"](c), c.build_config()
    for field, _ in ns["_ScConfig"]._fields_:
      if field in ("reserved", "integration_type"):  # (unused by ConstraintPropagation)
        continue
      a, b = getattr(got, field), getattr(want, field)
      if hasattr(a, "__len__"):
        a, b = list(a), list(b)
      assert a == b, field
