"""GPU parity tests for the constraint operators (reference
constraint.py:95-164 and spectral_clusterer.py:137-142, 259-264): the device kernels and
the Neumann-product inverse against the CPU oracle and golden outputs of the reference.

Tolerances: AffinityIntegration is elementwise -> bit-exact.  ConstraintPropagation
replaces LAPACK's LU inverse by a product of fp64 GEMMs; I - alpha*A_norm has condition
number <= (1 + alpha) / (1 - alpha), so the two agree to ~1e-13 relative to the largest
entry (asserted: 1e-11).
"""

import copy

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden

import spectralcluster_amd as sca
from spectralcluster_amd import constraint as con

pytestmark = pytest.mark.gpu

CP_TOL = 1e-11


def max_err(got, want):
  return float(np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want))))


# --- the reference's own known answers (tests/constraint_test.py) -----------------
def test_reference_3by3_known_answers():
  a = np.array([[1, 0.25, 0], [0.31, 1, 0], [0, 0, 1]])
  q = np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]], dtype=np.float64)
  got = con.AffinityIntegration(con.IntegrationType.Max).adjust_affinity(a, q)
  np.testing.assert_allclose(got, [[1, 1, 0], [1, 1, 0], [0, 0, 1]], atol=0.01)
  got = con.ConstraintPropagation(alpha=0.6).adjust_affinity(a, q)
  np.testing.assert_allclose(got, [[1, 0.97, 0], [1.03, 1, 0], [0, 0, 1]], atol=0.01)
  assert max_err(got, so.constraint_propagation(a, q, 0.6)) < CP_TOL


# --- per-op parity against golden outputs of the reference ------------------------
@pytest.mark.parametrize("aname", ["sym", "gen"])
@pytest.mark.parametrize("qname", ["sym", "gen"])
def test_ops_vs_reference_golden(aname, qname):
  g = golden("constraint_ops_n40.npz")
  a, q = g["a_" + aname], g["q_" + qname]
  tag = "a%s_q%s" % (aname, qname)
  got = con.AffinityIntegration(con.IntegrationType.Max).adjust_affinity(a, q)
  assert np.array_equal(got, g["integ_max_" + tag])
  got = con.AffinityIntegration(con.IntegrationType.Average).adjust_affinity(a, q)
  assert np.array_equal(got, g["integ_avg_" + tag])
  for alpha in (0.4, 0.6, 0.9):
    got = con.ConstraintPropagation(alpha).adjust_affinity(a, q)
    assert max_err(got, g["cp_%02d_%s" % (round(alpha * 10), tag)]) < CP_TOL
    if aname == "sym" and qname == "sym":
      # symmetric up to the rounding of (d_i a_ij) d_j vs (d_j a_ji) d_i, as in the reference
      np.testing.assert_allclose(got, got.T, rtol=0, atol=1e-14)


@pytest.mark.parametrize("n,alpha", [(1, 0.6), (2, 0.6), (17, 0.4), (129, 0.6), (300, 0.4),
                                     (1000, 0.6), (2049, 0.4)])
def test_constraint_propagation_vs_oracle(n, alpha):
  x, _, scores = so.turn_blobs(n, 16, 3, seed=n) if n > 2 else (
      so.blobs(n, 4, 1, seed=n), None, np.zeros(n))
  a = so.affinity(x)
  q = so.constraint_matrix_diagonals(list(scores), 1)
  got = con.ConstraintPropagation(alpha).adjust_affinity(a, q)
  assert max_err(got, so.constraint_propagation(a, q, alpha)) < CP_TOL


def test_constraint_propagation_alpha_edge_cases():
  x = so.blobs(60, 8, 3, seed=5)
  a = so.affinity(x)
  q = so.constraint_matrix_diagonals([0, 0, 5, 0, 0.5, 0] * 10, 1)
  # alpha = 0: T = I, F = Q
  got = con.ConstraintPropagation(0.0).adjust_affinity(a, q)
  assert max_err(got, so.constraint_propagation(a, q, 0.0)) < 1e-15
  got = con.ConstraintPropagation(0.99).adjust_affinity(a, q)
  assert max_err(got, so.constraint_propagation(a, q, 0.99)) < 1e-9  # cond ~ 200
  with pytest.raises(sca.UnsupportedOnDeviceError):
    con.ConstraintPropagation(1.0).adjust_affinity(a, q)
  with pytest.raises(sca.UnsupportedOnDeviceError):
    con.ConstraintPropagation(1.5).adjust_affinity(a, q)


def test_integration_nan_propagates_like_numpy():
  a = np.array([[1.0, np.nan], [0.2, 1.0]])
  q = np.array([[0.0, 1.0], [np.nan, 0.0]])
  got = con.AffinityIntegration(con.IntegrationType.Max).adjust_affinity(a, q)
  np.testing.assert_array_equal(got, np.maximum(a, q))


# --- the reference's 6x2 clusterer tests (tests/spectral_clusterer_test.py:243-328) ---
TOY = np.array([[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.0], [0.9, -0.1], [0.0, 1.2]])


def toy_refinement():
  return sca.RefinementOptions(
      p_percentile=0.95, thresholding_type=sca.ThresholdType.Percentile,
      thresholding_with_binarization=True, thresholding_preserve_diagonal=True,
      symmetrize_type=sca.SymmetrizeType.Average,
      refinement_sequence=sca.TURNTODIARIZE_REFINEMENT_SEQUENCE)


def test_6by2_affinity_integration():
  q = np.zeros((6, 6))
  q[0, 0] = q[1, 1] = 1
  q[2:, 2:] = 1
  clusterer = sca.SpectralClusterer(
      max_clusters=2, refinement_options=toy_refinement(),
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName.AffinityIntegration,
          apply_before_refinement=False, integration_type=sca.IntegrationType.Max),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY, q))
  np.testing.assert_equal(labels, [0, 0, 1, 1, 1, 1])


def test_6by2_constraint_propagation():
  q = np.eye(6)
  q[0, 1] = q[1, 0] = 1
  q[4, 5] = q[5, 4] = -1
  clusterer = sca.SpectralClusterer(
      max_clusters=2, refinement_options=toy_refinement(),
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName.ConstraintPropagation,
          apply_before_refinement=True, constraint_propagation_alpha=0.6),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY, q))
  np.testing.assert_equal(labels, [0, 0, 1, 1, 0, 1])
  # without the matrix the options are inert (reference :259-264)
  free = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  cfg = so.turntodiarize_config(min_clusters=None, max_clusters=2, p_percentile=0.95)
  want = so.ordered_labels(so.predict(TOY, cfg))
  np.testing.assert_equal(free, want)


# --- Turn-to-Diarize preset end to end, against the reference's outputs ----------------
@pytest.mark.parametrize("n", [120, 300, 700])
def test_turntodiarize_preset_vs_reference(n):
  g = golden("turntodiarize_n%d.npz" % n)
  x, truth, scores = so.turn_blobs(n, int(g["d"]), int(g["k"]), int(g["seed"]))
  q = sca.ConstraintMatrix(list(scores), threshold=1).compute_diagonals()
  pristine = sca.configs.turntodiarize_clusterer
  # (a) constraint-adjusted affinity
  adj = pristine.constraint_options.constraint_operator.adjust_affinity(so.affinity(x), q)
  np.testing.assert_allclose(
      [adj.sum(), np.abs(adj).max(), adj[0, 1], adj[n // 2, n // 3]],
      g["adjusted_checksum"], rtol=1e-11)
  # (b) the AutoTune proxy over the whole grid
  clusterer = copy.deepcopy(pristine)
  grid = np.array(clusterer.autotune.get_percentile_range())
  np.testing.assert_array_equal(grid, g["grid"])
  ratios, ks = [], []
  for p in grid:
    clusterer.refinement_options.p_percentile = p
    _, k, delta = clusterer._compute_eigenvectors_ncluster(adj, q)
    ratios.append(np.sqrt(1 - p) / delta)
    ks.append(k)
  np.testing.assert_allclose(ratios, g["ratios"], rtol=1e-5)
  np.testing.assert_array_equal(ks, g["n_clusters"])
  # (c) labels, with and without the constraint matrix
  labels = copy.deepcopy(pristine).predict(x, q)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  free = copy.deepcopy(pristine).predict(x)
  assert so.adjusted_rand_index(free, g["labels_unconstrained"]) == 1.0


@pytest.mark.parametrize("n,noise", [(400, 1.0), (1200, 1.2)])
def test_turntodiarize_noisy_vs_oracle(n, noise):
  """Harder conversations (constraints change the outcome): same labels as the oracle."""
  x, truth, scores = so.turn_blobs(n, 24, 4, seed=n + 1, noise=noise)
  q = so.constraint_matrix_diagonals(list(scores), 1)
  cfg = so.turntodiarize_config()
  want = so.predict(x, cfg, constraint_matrix=q, autotune=so.TURNTODIARIZE_AUTOTUNE)
  got = copy.deepcopy(sca.configs.turntodiarize_clusterer).predict(x, q)
  assert so.adjusted_rand_index(got, want) == 1.0


def test_integration_after_refinement_vs_reference():
  g = golden("integration_n200.npz")
  x, _, _ = so.turn_blobs(200, 16, 3, 17)
  for tag, kind in (("max", sca.IntegrationType.Max), ("avg", sca.IntegrationType.Average)):
    opts = toy_refinement()
    opts.p_percentile = 0.9
    clusterer = sca.SpectralClusterer(
        max_clusters=6, refinement_options=opts,
        constraint_options=sca.ConstraintOptions(
            constraint_name=sca.ConstraintName.AffinityIntegration,
            apply_before_refinement=False, integration_type=kind),
        laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
    labels = clusterer.predict(x, g["q"])
    assert so.adjusted_rand_index(labels, g["labels_" + tag]) == 1.0
    _, k, delta = clusterer._compute_eigenvectors_ncluster(so.affinity(x), g["q"])
    assert k == int(g["n_clusters_" + tag])
    np.testing.assert_allclose(delta, float(g["max_delta_" + tag]), rtol=1e-5)


def test_propagation_after_refinement_and_icassp_sequence():
  """ConstraintPropagation after an ICASSP-style refinement that ends in Diffuse (the
  refined matrix is symmetric, so the symmetric GEMM path runs on it)."""
  x, _, scores = so.turn_blobs(500, 32, 4, seed=23)
  q = so.constraint_matrix_diagonals(list(scores), 1)
  seq = sca.ICASSP2018_REFINEMENT_SEQUENCE[:-1]  # without RowWiseNormalize
  opts = sca.RefinementOptions(gaussian_blur_sigma=1, p_percentile=0.95,
                               refinement_sequence=seq)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=7, refinement_options=opts,
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName.ConstraintPropagation,
          apply_before_refinement=False, constraint_propagation_alpha=0.6))
  got = clusterer.predict(x, q)
  cfg = so.icassp2018_config(sequence=so.ICASSP2018_SEQUENCE[:-1],
                             constraint_name=so.CONSTRAINT_PROPAGATION,
                             apply_before_refinement=False,
                             constraint_propagation_alpha=0.6)
  dump = {}
  want = so.predict(x, cfg, dump, constraint_matrix=q)
  assert so.adjusted_rand_index(got, want) == 1.0
  idx = so.consumed_eigen_indices(500, 7, True, dump["eigenvalues"], 1e-2)
  w = clusterer.last_diag.eigenvalue_array()
  assert np.max(np.abs(w[idx] - dump["eigenvalues"][idx]) /
                np.abs(dump["eigenvalues"][idx])) < 1e-6


def test_constraint_errors():
  x = so.blobs(50, 8, 2, seed=1)
  clusterer = copy.deepcopy(sca.configs.turntodiarize_clusterer)
  with pytest.raises(ValueError, match="same shape"):
    clusterer.predict(x, np.zeros((49, 49)))
  with pytest.raises(ValueError, match="square"):
    clusterer.predict(x, np.zeros((50, 49)))
  # a non-symmetric constraint matrix after a symmetric refinement leaves a general
  # matrix: the general eigen path takes it
  q = np.zeros((50, 50))
  q[3, 7] = 1.0
  after = sca.SpectralClusterer(
      max_clusters=4, refinement_options=toy_refinement(),
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName.AffinityIntegration,
          apply_before_refinement=False, integration_type=sca.IntegrationType.Max))
  got = after.predict(x, q)
  assert after.last_diag.symmetry_state == 3
  cfg = so.turntodiarize_config(min_clusters=None, max_clusters=4, p_percentile=0.95,
                                row_wise_renorm=False, laplacian_type=so.LAPLACIAN_NONE,
                                constraint_name=so.CONSTRAINT_AFFINITY_INTEGRATION,
                                apply_before_refinement=False,
                                integration_type=so.INTEGRATION_MAX)
  assert so.adjusted_rand_index(got, so.predict(x, cfg, constraint_matrix=q)) == 1.0
  # ... and before a symmetrising refinement the symmetric path still runs
  before = sca.SpectralClusterer(
      max_clusters=4, refinement_options=toy_refinement(),
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName.AffinityIntegration,
          apply_before_refinement=True, integration_type=sca.IntegrationType.Max),
      laplacian_type=sca.LaplacianType.GraphCut)
  got = before.predict(x, q)
  cfg = so.turntodiarize_config(min_clusters=None, max_clusters=4, p_percentile=0.95,
                                row_wise_renorm=False,
                                constraint_name=so.CONSTRAINT_AFFINITY_INTEGRATION,
                                integration_type=so.INTEGRATION_MAX)
  assert so.adjusted_rand_index(got, so.predict(x, cfg, constraint_matrix=q)) == 1.0
  with pytest.raises(RuntimeError):
    sca.SpectralClusterer(max_spectral_size=20).predict(x, np.zeros((50, 50)))


def test_stale_constraint_is_not_reused():
  """A constraint matrix given to one predict() must not leak into the next call."""
  x, _, scores = so.turn_blobs(150, 16, 3, seed=3, noise=1.0)
  q = so.constraint_matrix_diagonals(list(scores), 1)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=7, refinement_options=toy_refinement(),
      constraint_options=copy.deepcopy(sca.configs.turntodiarize_constraint_options),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  first = clusterer.predict(x)
  clusterer.predict(x, q)
  again = clusterer.predict(x)
  np.testing.assert_array_equal(first, again)
  batch = clusterer.predict_batch([x, x], streams=1)
  np.testing.assert_array_equal(batch[0], first)


# --- randomised sweep over constraint configurations ----------------------------------------
def _constraint_fuzz_cases():
  rng = np.random.default_rng(404)
  cases = []
  for i in range(16):
    n = int(rng.integers(30, 600))
    k = int(rng.integers(2, 5))
    name = int(rng.choice([so.CONSTRAINT_AFFINITY_INTEGRATION, so.CONSTRAINT_PROPAGATION]))
    before = bool(rng.integers(0, 2))
    kind = int(rng.choice([so.INTEGRATION_MAX, so.INTEGRATION_AVERAGE]))
    alpha = float(rng.choice([0.2, 0.4, 0.6, 0.8]))
    seq = str(rng.choice(["ttd", "icassp_nonorm"]))
    lap = int(rng.choice([0, 4]))
    sym_q = bool(rng.integers(0, 4))     # one in four gets a non-symmetric constraint matrix
    cases.append((i, n, k, name, before, kind, alpha, seq, lap, sym_q))
  return cases


@pytest.mark.parametrize("case", _constraint_fuzz_cases(), ids=lambda c: "cfuzz%d" % c[0])
def test_fuzz_constraints_vs_oracle(case):
  i, n, k, name, before, kind, alpha, seq, lap, sym_q = case
  x, _, scores = so.turn_blobs(n, 24, k, seed=7000 + i, noise=0.8)
  q = so.constraint_matrix_diagonals(list(scores), 1)
  if not sym_q:
    q = np.triu(q)                        # cannot-/must-links recorded one way only
  if seq == "ttd":
    ocfg = so.turntodiarize_config(p_percentile=0.9, laplacian_type=lap, row_wise_renorm=False)
    options = toy_refinement()
    options.p_percentile = 0.9
  else:
    ocfg = so.icassp2018_config(sequence=so.ICASSP2018_SEQUENCE[:-1], laplacian_type=lap)
    options = sca.RefinementOptions(gaussian_blur_sigma=1, p_percentile=0.95,
                                    refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE[:-1])
  import dataclasses
  ocfg = dataclasses.replace(ocfg, min_clusters=2, max_clusters=7, constraint_name=name,
                             apply_before_refinement=before, integration_type=kind,
                             constraint_propagation_alpha=alpha)
  dump = {}
  want = so.predict(x, ocfg, dump, constraint_matrix=q)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=7, refinement_options=options,
      laplacian_type=sca.LaplacianType(lap) if lap else None,
      constraint_options=sca.ConstraintOptions(
          constraint_name=sca.ConstraintName(name), apply_before_refinement=before,
          integration_type=sca.IntegrationType(kind), constraint_propagation_alpha=alpha))
  got = clusterer.predict(x, q)
  diag = clusterer.last_diag
  if dump["max_delta"] < 1e-9:
    return
  assert diag.n_clusters == dump["n_clusters"]
  np.testing.assert_allclose(diag.max_delta, dump["max_delta"], rtol=1e-5)
  assert so.adjusted_rand_index(got, want) == 1.0
