"""CPU: pin the oracle (`oracle/spectral_oracle.py`) against golden outputs of the
REAL reference (tests/golden/*.npz, produced by oracle/make_golden.py which imports
/root/reference) and against the reference's own known-answer test vectors."""

import dataclasses

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden


def test_affinity_known_answer():
  # reference tests/utils_test.py:10-15
  m = np.array([[3, 4], [-4, 3], [6, 8], [-3, -4]], dtype=np.float64)
  expected = np.array([[1, 0.5, 1, 0], [0.5, 1, 0.5, 0.5], [1, 0.5, 1, 0],
                       [0, 0.5, 0, 1]])
  np.testing.assert_equal(so.affinity(m), expected)


def test_ops_bit_exact_vs_reference():
  g = golden("ops_n40.npz")
  m = g["input"]
  assert np.array_equal(so.crop_diagonal(m), g["crop"])
  assert np.array_equal(so.gaussian_blur(m, 1), g["blur_s1"])
  assert np.array_equal(so.gaussian_blur(m, 2), g["blur_s2"])
  assert np.array_equal(so.symmetrize(m, so.SYMMETRIZE_MAX), g["sym_max"])
  assert np.array_equal(so.symmetrize(m, so.SYMMETRIZE_AVERAGE), g["sym_avg"])
  assert np.array_equal(so.diffuse(m), g["diffuse"])
  assert np.array_equal(so.row_wise_normalize(m), g["rownorm"])
  for tname, tt in (("rowmax", so.THRESHOLD_ROW_MAX), ("pct", so.THRESHOLD_PERCENTILE)):
    for bz in (0, 1):
      for pd in (0, 1):
        got = so.row_wise_threshold(m, 0.8, 0.01, tt, bool(bz), bool(pd))
        assert np.array_equal(got, g["thr_%s_b%d_d%d" % (tname, bz, pd)])
  for lap in (2, 3, 4):
    assert np.array_equal(so.laplacian(g["sym_max"], lap), g["lap%d" % lap])


def test_refinement_known_answers():
  # reference tests/refinement_test.py
  m = np.array([[1, 2, 3], [3, 4, 5], [4, 2, 1]], dtype=np.float64)
  np.testing.assert_equal(so.crop_diagonal(m), [[3, 2, 3], [3, 5, 5], [4, 2, 4]])
  b = np.array([[1.0, 2.0, 3.0], [3.0, 4.0, 5.0], [4.0, 2.0, 1.0]])
  np.testing.assert_allclose(so.gaussian_blur(b, 1),
                             [[2.12, 2.61, 3.10], [2.76, 2.90, 3.06],
                              [3.16, 2.78, 2.46]], atol=0.01)
  t = np.array([[0.5, 2.0, 3.0], [3.0, 4.0, 5.0], [4.0, 2.0, 1.0]])
  np.testing.assert_allclose(
      so.row_wise_threshold(t, 0.5, 0.01, so.THRESHOLD_PERCENTILE),
      [[0.005, 2.0, 3.0], [0.03, 4.0, 5.0], [4.0, 2.0, 0.01]], atol=0.001)
  np.testing.assert_allclose(
      so.row_wise_threshold(t, 0.5, 0.01, so.THRESHOLD_ROW_MAX),
      [[0.005, 2.0, 3.0], [3.0, 4.0, 5.0], [4.0, 2.0, 0.01]], atol=0.001)
  np.testing.assert_allclose(
      so.row_wise_threshold(t, 0.5, 0.01, so.THRESHOLD_ROW_MAX, True),
      [[0.005, 1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 1.0, 0.01]], atol=0.001)
  np.testing.assert_equal(so.symmetrize(m), [[1, 3, 4], [3, 4, 5], [4, 5, 1]])
  np.testing.assert_equal(so.diffuse(np.array([[1.0, 2.0], [3.0, 4.0]])),
                          [[5, 11], [11, 25]])
  np.testing.assert_allclose(so.row_wise_normalize(np.array([[1.0, 2.0], [3.0, 4.0]])),
                             [[0.5, 1.0], [0.75, 1.0]], atol=0.001)
  with pytest.raises(ValueError):
    so.crop_diagonal(np.zeros((2, 3)))


def test_eigengap_known_answers():
  # reference tests/utils_test.py:43-67
  w = np.array([1.0, 0.9, 0.8, 0.2, 0.1])
  k, d = so.eigengap(w)
  assert k == 3 and abs(d - 4.0) < 0.01
  w6 = np.array([1.0, 0.9, 0.8, 0.7, 0.6, 0.5])
  k, d = so.eigengap(w6)
  assert k == 5 and abs(d - 1.2) < 0.01
  k, d = so.eigengap(w6, max_clusters=2)
  assert k == 2 and abs(d - 1.125) < 0.01
  k, d = so.eigengap(w, max_clusters=3, descend=False)
  assert k == 2 and abs(d - 0.88) < 0.01
  with pytest.raises(TypeError):
    so.eigengap(w, eigengap_type="Ratio")


@pytest.mark.parametrize("lap", [0, 2, 3, 4])
def test_toy_stage_dump(lap):
  g = golden("toy6x2_lap%d.npz" % lap)
  cfg = so.OracleConfig(sequence=so.ICASSP2018_SEQUENCE, gaussian_blur_sigma=0,
                        p_percentile=0.95, laplacian_type=lap)
  dump = {}
  labels = so.predict(g["x"], cfg, dump)
  assert np.array_equal(dump["affinity"], g["affinity"])
  for i, st in enumerate(dump["stages"]):
    assert np.array_equal(st, g["stage%d" % i])
  assert np.array_equal(dump["eigenvalues"], g["eigenvalues"])
  assert dump["n_clusters"] == int(g["n_clusters_raw"])  # (min_clusters is None here)
  assert np.array_equal(labels, g["labels"])
  assert np.array_equal(so.ordered_labels(labels), [0, 0, 1, 1, 0, 1])


@pytest.mark.parametrize("lap", [0, 4])
def test_n64_stage_dump(lap):
  g = golden("stages_n64_lap%d.npz" % lap)
  cfg = so.icassp2018_config(laplacian_type=lap)
  dump = {}
  labels = so.predict(g["x"], cfg, dump)
  for i, st in enumerate(dump["stages"]):
    assert np.array_equal(st, g["stage%d" % i]), "stage %d" % i
  if lap == 4:
    assert np.array_equal(dump["laplacian"], g["laplacian"])
  assert np.array_equal(dump["eigenvalues"], g["eigenvalues"])
  assert np.array_equal(labels, g["labels"])
  assert dump["max_delta"] == float(g["max_delta"])


def test_kmeans_vs_sklearn_and_reference():
  g = golden("kmeans.npz")
  for tag, k in (("a", 4), ("b", 8), ("c", 2), ("d", 20)):
    e = g["e_" + tag]
    np.testing.assert_allclose(so.sklearn_init_centroids(e, k), g["centers_" + tag],
                               rtol=0, atol=1e-14)
    assert np.array_equal(so.run_kmeans(e, k, 300), g["labels_" + tag])


def test_mt19937_matches_numpy_randomstate():
  rng = so.Mt19937(0)
  want = np.random.RandomState(0).random_sample(1500)
  got = np.array([rng.next_double() for _ in range(1500)])
  assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["e2e_n200_lap0_max7.npz", "e2e_n200_lap4_max7.npz",
                                  "e2e_n1000_lap0_max7.npz", "e2e_n1000_lap4_max20.npz",
                                  "e2e_n1000_lap3_max20.npz", "e2e_n1000_lap2_max20.npz"])
def test_e2e_small(name):
  g = golden(name)
  n, d, k, seed, lap, max_clusters = [int(v) for v in g["params"]]
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=max_clusters)
  dump = {}
  labels = so.predict(so.blobs(n, d, k, seed), cfg, dump)
  assert np.array_equal(labels, g["labels"])
  assert np.array_equal(dump["eigenvalues"][g["consumed_index"]], g["consumed_eigenvalues"])
  assert dump["max_delta"] == float(g["max_delta"])


@pytest.mark.parametrize("name", ["manyk_n1500_k90_lap0_max120.npz",
                                  "manyk_n1500_k90_lap4_max120.npz"])
def test_e2e_more_than_64_clusters(name):
  """The reference selects 90 / 89 clusters here (max_clusters = 120); the oracle follows it
  through the eigengap rule and the 90-centre k-means++ / cosine loop.  (The device path of
  round 3 raises UnsupportedOnDeviceError past 64 selected clusters: these pin the checker for
  the round that lifts it.)"""
  g = golden(name)
  n, d, k, seed, lap, max_clusters = [int(v) for v in g["params"]]
  assert int(g["n_clusters_raw"]) > 64
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=max_clusters)
  dump = {}
  labels = so.predict(so.blobs(n, d, k, seed), cfg, dump)
  assert dump["n_clusters"] == int(g["n_clusters_raw"])
  assert np.array_equal(labels, g["labels"])
  assert np.array_equal(dump["eigenvalues"][g["consumed_index"]], g["consumed_eigenvalues"])
  assert dump["max_delta"] == float(g["max_delta"])


def test_autotune_small():
  g = golden("autotune_n512.npz")
  x = so.blobs(512, 64, 6, 512)
  cfg = so.icassp2018_config(laplacian_type=so.LAPLACIAN_GRAPH_CUT, max_clusters=20)
  grid = so.autotune_range(0.55, 0.95, 0.025)
  np.testing.assert_array_equal(grid, g["grid"])
  _, k, best_p, seen = so.autotune_search(so.affinity(x), cfg, 0.55, 0.95, 0.025)
  np.testing.assert_allclose([seen[p] for p in grid], g["ratios"], rtol=1e-12)
  assert best_p == float(g["best_p"])


def test_constraint_ops_vs_reference():
  g = golden("constraint_ops_n40.npz")
  for aname in ("sym", "gen"):
    for qname in ("sym", "gen"):
      a, q = g["a_" + aname], g["q_" + qname]
      tag = "a%s_q%s" % (aname, qname)
      assert np.array_equal(so.affinity_integration(a, q, so.INTEGRATION_MAX),
                            g["integ_max_" + tag])
      assert np.array_equal(so.affinity_integration(a, q, so.INTEGRATION_AVERAGE),
                            g["integ_avg_" + tag])
      for alpha in (0.4, 0.6, 0.9):
        got = so.constraint_propagation(a, q, alpha)
        # same LAPACK inverse and matmuls; the diag-matrix products of the reference
        # are elementwise scalings here, identical up to the order of two roundings
        np.testing.assert_allclose(got, g["cp_%02d_%s" % (round(alpha * 10), tag)],
                                   rtol=1e-12, atol=1e-14)
  scores = list(g["turn_scores"])
  np.testing.assert_array_equal(so.constraint_matrix_diagonals(scores, 1), g["turn_matrix"])
  np.testing.assert_array_equal(so.constraint_matrix_diagonals(scores, 3),
                                g["turn_matrix_t3"])


def test_constraint_known_answers():
  # reference tests/constraint_test.py
  a = np.array([[1, 0.25, 0], [0.31, 1, 0], [0, 0, 1]])
  q = np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]], dtype=np.float64)
  np.testing.assert_allclose(so.affinity_integration(a, q),
                             [[1, 1, 0], [1, 1, 0], [0, 0, 1]], atol=0.01)
  np.testing.assert_allclose(so.constraint_propagation(a, q, 0.6),
                             [[1, 0.97, 0], [1.03, 1, 0], [0, 0, 1]], atol=0.01)
  np.testing.assert_equal(so.constraint_matrix_diagonals([0, 0, 14.308253288269043], 1),
                          [[0, 1, 0], [1, 0, -1], [0, -1, 0]])
  np.testing.assert_equal(so.constraint_matrix_diagonals([0, 0, 0.12095779925584793], 1),
                          [[0, 1, 0], [1, 0, 0], [0, 0, 0]])
  with pytest.raises(ValueError):
    so.constraint_matrix_diagonals([0, -1.0], 1)
  with pytest.raises(ValueError):
    so.affinity_integration(a, q[:2, :2])


@pytest.mark.parametrize("n", [120, 300])
def test_turntodiarize_constrained_predict(n):
  g = golden("turntodiarize_n%d.npz" % n)
  x, truth, scores = so.turn_blobs(n, int(g["d"]), int(g["k"]), int(g["seed"]))
  np.testing.assert_array_equal(truth, g["truth"])
  np.testing.assert_array_equal(scores, g["scores"])
  q = so.constraint_matrix_diagonals(list(scores), 1)
  cfg = so.turntodiarize_config()
  adj = so.constraint_propagation(so.affinity(x), q, 0.4)
  np.testing.assert_allclose(
      [adj.sum(), np.abs(adj).max(), adj[0, 1], adj[n // 2, n // 3]],
      g["adjusted_checksum"], rtol=1e-11)
  grid = so.autotune_range(*so.TURNTODIARIZE_AUTOTUNE[:3])
  np.testing.assert_array_equal(grid, g["grid"])
  _, _, _, seen = so.autotune_search(adj, cfg, *so.TURNTODIARIZE_AUTOTUNE,
                                     constraint_matrix=q)
  np.testing.assert_allclose([seen[p] for p in grid], g["ratios"], rtol=1e-7)
  labels = so.predict(x, cfg, constraint_matrix=q, autotune=so.TURNTODIARIZE_AUTOTUNE)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  free = so.predict(x, cfg, autotune=so.TURNTODIARIZE_AUTOTUNE)
  assert so.adjusted_rand_index(free, g["labels_unconstrained"]) == 1.0


def test_integration_after_refinement_predict():
  g = golden("integration_n200.npz")
  x, truth, _ = so.turn_blobs(200, 16, 3, 17)
  np.testing.assert_array_equal(truth, g["truth"])
  for tag, kind in (("max", so.INTEGRATION_MAX), ("avg", so.INTEGRATION_AVERAGE)):
    cfg = so.turntodiarize_config(
        min_clusters=None, max_clusters=6, p_percentile=0.9,
        constraint_name=so.CONSTRAINT_AFFINITY_INTEGRATION,
        apply_before_refinement=False, integration_type=kind)
    dump = {}
    labels = so.predict(x, cfg, dump, constraint_matrix=g["q"])
    assert so.adjusted_rand_index(labels, g["labels_" + tag]) == 1.0
    _, k, delta = so.eig_ncluster(so.affinity(x), cfg, constraint_matrix=g["q"])
    assert k == int(g["n_clusters_" + tag])
    np.testing.assert_allclose(delta, float(g["max_delta_" + tag]), rtol=1e-8)


@pytest.mark.parametrize("n", [60, 300])
def test_general_matrix_autotune_vs_reference(n):
  """[RowWiseThreshold] + GraphCut: np.linalg.eig on a genuinely non-symmetric matrix."""
  g = golden("general_n%d.npz" % n)
  x = so.blobs(n, int(g["d"]), int(g["k"]), int(g["seed"]))
  cfg = so.OracleConfig(min_clusters=2, max_clusters=6, sequence=(so.OP_ROW_WISE_THRESHOLD,),
                        threshold_type=so.THRESHOLD_PERCENTILE,
                        laplacian_type=so.LAPLACIAN_GRAPH_CUT, row_wise_renorm=True)
  grid = so.autotune_range(0.60, 0.95, 0.05)
  np.testing.assert_array_equal(grid, g["grid"])
  a = so.affinity(x)
  for i, p in enumerate(grid):
    dump = {}
    _, k, delta = so.eig_ncluster(a, dataclasses.replace(cfg, p_percentile=p), dump)
    assert k == int(g["n_clusters"][i])
    np.testing.assert_allclose(delta, g["max_delta"][i], rtol=1e-9)
    np.testing.assert_allclose(dump["eigenvalues"][1:8], g["eigenvalues"][i][1:8], rtol=1e-9)
  labels = so.predict(x, cfg, autotune=(0.60, 0.95, 0.05, 1, True))
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


def test_general_matrix_more_than_32_eigenpairs():
  """[RowWiseThreshold] + GraphCut with min_clusters = 40, max_clusters = 48: the reference
  embeds in 40 eigenvectors of a non-symmetric matrix.  (Groundwork: the device's general path
  of round 3 holds at most 32 eigenpairs for n > 64.)"""
  g = golden("general_wide_n400.npz")
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  cfg = so.OracleConfig(min_clusters=int(g["min_clusters"]), max_clusters=maxc,
                        sequence=(so.OP_ROW_WISE_THRESHOLD,), p_percentile=float(g["p_percentile"]),
                        threshold_type=so.THRESHOLD_PERCENTILE,
                        laplacian_type=so.LAPLACIAN_GRAPH_CUT, row_wise_renorm=True)
  dump = {}
  labels = so.predict(so.blobs(n, d, k, seed), cfg, dump)
  assert len(np.unique(labels)) == len(np.unique(g["labels"])) > 32
  assert np.array_equal(labels, g["labels"])
  np.testing.assert_allclose(dump["eigenvalues"][:maxc + 2], g["head_eigenvalues"], rtol=1e-9,
                             atol=1e-12)
  np.testing.assert_allclose(dump["max_delta"], float(g["max_delta"]), rtol=1e-9)


def test_size_reduction_vs_reference():
  g = golden("size_reduction.npz")
  x = g["x_1000by6"]
  assert np.array_equal(so.agglomerative(x, 100, "complete"), g["ahc_1000by6"])
  cfg = so.icassp2018_config(min_clusters=None, max_clusters=None, gaussian_blur_sigma=0)
  got = so.reduce_size_and_predict(x, cfg, 100)
  assert got.dtype == np.float64 and g["labels_1000by6"].dtype == np.float64
  assert so.adjusted_rand_index(got.astype(int), g["labels_1000by6"].astype(int)) == 1.0
  assert np.array_equal(so.ordered_labels(got.astype(int)),
                        [0] * 400 + [1] * 300 + [2] * 200 + [3] * 100)
  xx = so.blobs(1500, 32, 5, 91)
  dump = {}
  got = so.reduce_size_and_predict(xx, so.icassp2018_config(), 200, dump)
  assert np.array_equal(dump["ahc_labels"], g["ahc_a"])
  assert so.adjusted_rand_index(got.astype(int), g["labels_a"].astype(int)) == 1.0
  # centroids: np.mean(axis=0) adds the members in index order
  c = so.get_cluster_centroids(xx, g["ahc_a"])
  members = np.flatnonzero(g["ahc_a"] == 3)
  acc = np.zeros(32)
  for i in members:
    acc = acc + xx[i]
  assert np.array_equal(c[3], acc / len(members))
  with pytest.raises(ValueError):
    so.chain_labels(np.array([0, 1, 2]), np.array([0, 1]))


def test_kmeans_other_metrics_vs_reference():
  g = golden("kmeans_metrics.npz")
  for tag, k in (("a", 4), ("b", 7), ("c", 2)):
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev", "correlation",
                   "braycurtis", "canberra", "minkowski"):
      got = so.run_kmeans_metric(g["e_" + tag], k, 300, metric)
      assert np.array_equal(got, g["labels_%s_%s" % (tag, metric)])
  # more than 128 clusters (numpy's pairwise row mean inside scipy's correlation)
  gk = golden("kmeans_correlation_k150.npz")
  assert np.array_equal(so.run_kmeans_metric(gk["e"], 150, 300, "correlation"), gk["labels"])
  # cosine through the generic function equals the bit-exact restatement
  e = g["e_a"]
  assert np.array_equal(so.run_kmeans_metric(e, 4, 300, "cosine"), so.run_kmeans(e, 4, 300))
  from sklearn.exceptions import NotFittedError
  with pytest.raises(NotFittedError):   # the reference's latent bug (:33-36, :51)
    so.run_kmeans_metric(e, 4, 300, None)


def test_adjusted_rand_index():
  from sklearn.metrics import adjusted_rand_score
  rng = np.random.default_rng(0)
  for _ in range(5):
    a = rng.integers(0, 4, 200)
    b = rng.integers(0, 5, 200)
    assert abs(so.adjusted_rand_index(a, b) - adjusted_rand_score(a, b)) < 1e-12
  assert so.adjusted_rand_index(a, (a + 1) % 4) == 1.0


@pytest.mark.parametrize("name", ["e2e_n200_lap0_max7", "e2e_n200_lap4_max7",
                                  "e2e_n1000_lap0_max7", "e2e_n1000_lap4_max20",
                                  "e2e_n1000_lap3_max20", "e2e_n1000_lap2_max20"])
def test_algorithm_matched_cpu_path_vs_reference_golden(name):
  """`predict_algorithm_matched` (bench.py's second CPU leg and the checker of the large
  ragged GPU cases): the device's algorithm -- folded scaling vectors + a symmetric Krylov
  solver -- reproduces the real reference's consumed eigenvalues and labels."""
  g = golden(name + ".npz")
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=maxc)
  labels, w = so.predict_algorithm_matched(x, cfg)
  idx, ref = g["consumed_index"], g["consumed_eigenvalues"]
  if lap == 0:
    keep = so.consumed_eigen_indices(n, maxc, True, ref, 1e-2)
    idx, ref = idx[keep], ref[keep]
  rel = np.abs(w[idx] - ref) / np.maximum(np.abs(ref), 1e-12)
  assert rel.max() < 1e-8, rel.max()
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


@pytest.mark.parametrize("kind", so.HARD_KINDS)
@pytest.mark.parametrize("lap", [0, 4])
def test_hard_inputs_n1000(kind, lap):
  """Unfriendly spectra (oracle/make_golden.py --hard): the restatement must reproduce the
  real reference there too -- consumed eigenvalues, cluster count, labels."""
  g = golden("hard_%s_n1000_lap%d.npz" % (kind, lap))
  n, d, seed, lap_g, maxc = (int(v) for v in g["params"])
  x = so.hard_inputs(kind, n, d, seed)
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=maxc)
  dump = {}
  labels = so.predict(x, cfg, dump)
  w = np.real(dump["eigenvalues"])
  assert np.array_equal(w[g["consumed_index"]], g["consumed_eigenvalues"])
  assert dump["max_delta"] == float(g["max_delta"])
  assert dump["n_clusters"] == max(int(g["n_clusters_raw"]), 2)
  assert np.array_equal(labels, g["labels"])
