"""GPU parity tests for the general (non-symmetric) eigen path:
what `np.linalg.eig(...)` + `.real` + argsort give in reference utils.py:44-71 when the
refined matrix is NOT diagonally similar to a symmetric one (e.g. a refinement sequence
that ends in RowWiseThreshold -- the reference's own AutoTune tests,
tests/spectral_clusterer_test.py:156-241).

Tolerances: eigenvalues 1e-10 (dense) / 1e-7 (Arnoldi, whose stop rule is a 1e-10 residual
for the stage API and the 1e-6 bound of the pipeline); eigenvectors are compared up to the
sign of each column, like every other test (LAPACK fixes a complex vector's phase -- largest
component real -- and so does the device; a sign is not fixed by either).
"""

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden

import spectralcluster_amd as sca

pytestmark = pytest.mark.gpu

TOY = np.array([[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.0], [0.9, -0.1], [0.0, 1.2]])


def reference_sorted(m, descend=True):
  w, v = np.linalg.eig(m)          # utils.py:59
  w, v = w.real, v.real            # :60-61
  idx = np.argsort(-w if descend else w, kind="stable")
  return w[idx], v[:, idx]


def column_error(got, want):
  """max over columns of min(|g - w|, |g + w|) (sign-insensitive)."""
  worst = 0.0
  for j in range(want.shape[1]):
    worst = max(worst, min(np.abs(got[:, j] - want[:, j]).max(),
                           np.abs(got[:, j] + want[:, j]).max()))
  return worst


@pytest.mark.parametrize("n", [1, 2, 3, 6, 17, 33, 64])
@pytest.mark.parametrize("descend", [True, False])
def test_dense_general_eig_vs_numpy(n, descend):
  rng = np.random.default_rng(100 + n)
  m = rng.random((n, n))           # complex pairs are typical for n >= 3
  if n >= 2:
    m[0, 1] += 0.5                 # never symmetric
  w, v = sca.utils.compute_sorted_eigenvectors(m, descend=descend)
  wr, vr = reference_sorted(m, descend)
  assert w.shape == (n,) and v.shape == (n, n)
  np.testing.assert_allclose(w, wr, rtol=0, atol=1e-12 * max(1.0, np.abs(wr).max()))
  # columns of equal real part (conjugate pairs) may swap; their real parts are identical
  assert column_error(v, vr) < 1e-10


def test_dense_general_eig_defective_and_triangular():
  # already triangular, repeated eigenvalues (a Jordan block): eigenvalues must still come out
  m = np.array([[2.0, 1.0, 0.0], [0.0, 2.0, 1.0], [0.0, 0.0, 3.0]])
  w, _ = sca.utils.compute_sorted_eigenvectors(m)
  np.testing.assert_allclose(w, [3.0, 2.0, 2.0], atol=1e-7)
  # rotation-like block: purely complex pair, real parts both 0.5
  m = np.array([[0.5, -2.0, 0.0], [2.0, 0.5, 0.0], [0.1, 0.2, -1.0]])
  w, v = sca.utils.compute_sorted_eigenvectors(m)
  wr, vr = reference_sorted(m)
  np.testing.assert_allclose(w, wr, atol=1e-13)
  assert column_error(v, vr) < 1e-12


@pytest.mark.parametrize("n,d,k,seed", [(60, 8, 3, 5), (64, 8, 2, 2), (33, 6, 2, 3)])
def test_dense_general_near_defective_laplacians(n, d, k, seed):
  """Laplacians of heavily thresholded affinities carry a highly degenerate, nearly
  defective eigenvalue 1 (2x2 blocks [[a, b], [c, a]] with c ~ 1e-13): the Wilkinson shift
  must be computed without cancellation or the QR iteration stagnates."""
  a = so.affinity(so.blobs(n, d, k, seed))
  for lap in (2, 3, 4):
    for p in (0.3, 0.4, 0.6, 0.7, 0.95):
      m = so.laplacian(so.row_wise_threshold(a, p, 0.01, so.THRESHOLD_PERCENTILE), lap)
      w, _ = sca.utils.compute_sorted_eigenvectors(m, descend=False)
      wr = np.sort(np.linalg.eigvals(m).real)
      np.testing.assert_allclose(w[:8], wr[:8], rtol=0, atol=1e-10 * np.abs(wr).max())
      # the degenerate cluster itself is only defined to ~sqrt(eps) (Jordan-like blocks)
      np.testing.assert_allclose(w, wr, rtol=0, atol=1e-6 * np.abs(wr).max())


@pytest.mark.parametrize("scale", [1e-170, 1e160])
def test_dense_hessenberg_route_on_a_badly_scaled_matrix(scale):
  """the reflectors of the device reduction square their columns without dlarfg's
  safmin rescaling -- entries around 1e-160 underflowed to skipped reflectors, around 1e155
  overflowed.  The route now brings such a matrix to max|a| in [1, 2) by a power of two (exact)
  and gives the eigenvalues the factor back: same relative accuracy as at scale 1."""
  rng = np.random.default_rng(31)
  n, count = 150, 100          # (more than 64 pairs of a general matrix: eig_path 7)
  m = rng.random((n, n))
  m[0, 1] += 0.5
  ref = np.sort(np.linalg.eigvals(m).real)[::-1][:count]
  w, v = sca.utils.compute_sorted_eigenvectors(m * scale, descend=True, count=count)
  np.testing.assert_allclose(w / scale, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
  r = np.linalg.norm((m * scale) @ v[:, :1] - v[:, :1] * w[0]) / (np.abs(w[0]) * np.linalg.norm(v[:, 0]))
  assert r < 1e-10          # the Perron vector: real, simple


def thresholded(n, d, k, seed, p=0.9):
  x = so.blobs(n, d, k, seed=seed)
  return so.row_wise_threshold(so.affinity(x), p, 0.01, so.THRESHOLD_PERCENTILE), x


@pytest.mark.parametrize("n,k", [(65, 2), (300, 3), (1000, 5), (2500, 4)])
def test_arnoldi_top_eigenpairs_vs_numpy(n, k):
  m, _ = thresholded(n, 16, k, seed=n)
  assert not np.allclose(m, m.T)
  count = k + 1                    # the cluster eigenvalues and the edge of the bulk
  w, v = sca.utils.compute_sorted_eigenvectors(m, descend=True, count=count)
  wr, vr = reference_sorted(m)
  np.testing.assert_allclose(w, wr[:count], rtol=1e-8, atol=1e-9 * np.abs(wr).max())
  # the k cluster eigenvectors are well separated; compare those
  assert column_error(v[:, :k], vr[:, :k]) < 1e-6
  w2, _ = sca.utils.compute_sorted_eigenvectors(-m, descend=False, count=count)
  np.testing.assert_allclose(w2, -wr[:count], rtol=1e-8, atol=1e-9 * np.abs(wr).max())


# --- the reference's AutoTune tests: [RowWiseThreshold] + GraphCut (non-symmetric) -----
def threshold_only_options(**kw):
  return sca.RefinementOptions(thresholding_type=sca.ThresholdType.Percentile,
                               refinement_sequence=[sca.RefinementName.RowWiseThreshold], **kw)


def threshold_only_config(**kw):
  base = dict(sequence=(so.OP_ROW_WISE_THRESHOLD,), threshold_type=so.THRESHOLD_PERCENTILE,
              laplacian_type=so.LAPLACIAN_GRAPH_CUT, row_wise_renorm=True)
  base.update(kw)
  return so.OracleConfig(**base)


def test_6by2_auto_tune_reference_known_answer():
  # reference tests/spectral_clusterer_test.py:156-184
  clusterer = sca.SpectralClusterer(
      max_clusters=2, refinement_options=threshold_only_options(),
      autotune=sca.AutoTune(p_percentile_min=0.60, p_percentile_max=0.95,
                            init_search_step=0.05, search_level=1),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  np.testing.assert_equal(labels, [0, 0, 1, 1, 0, 1])
  assert clusterer.last_diag.symmetry_state == 3
  assert clusterer.last_diag.eig_path == 3       # dense general solver


def test_1000by6_auto_tune_reference_known_answer():
  # reference tests/spectral_clusterer_test.py:215-241 (seeded noise here)
  rng = np.random.default_rng(7)
  matrix = np.array([[1.0, 0, 0, 0, 0, 0]] * 400 + [[0, 1.0, 0, 0, 0, 0]] * 300 +
                    [[0, 0, 2.0, 0, 0, 0]] * 200 + [[0, 0, 0, 1.0, 0, 0]] * 100)
  matrix = matrix + (rng.random((1000, 6)) * 2 - 1) * 0.1
  clusterer = sca.SpectralClusterer(
      max_clusters=4, refinement_options=threshold_only_options(),
      autotune=sca.AutoTune(p_percentile_min=0.9, p_percentile_max=0.95,
                            init_search_step=0.03, search_level=1),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(matrix))
  np.testing.assert_equal(labels, [0] * 400 + [1] * 300 + [2] * 200 + [3] * 100)
  assert clusterer.last_diag.eig_path == 4       # block Arnoldi


@pytest.mark.parametrize("lap", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [40, 500])
def test_threshold_only_every_laplacian_vs_oracle(lap, n):
  x = so.blobs(n, 16, 3, seed=3 * n + lap)
  cfg = threshold_only_config(laplacian_type=lap, p_percentile=0.9, min_clusters=2,
                              max_clusters=6, row_wise_renorm=False)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=6,
      refinement_options=threshold_only_options(p_percentile=0.9),
      laplacian_type=sca.LaplacianType(lap) if lap else None)
  got = clusterer.predict(x)
  diag = clusterer.last_diag
  assert diag.symmetry_state == 3
  assert diag.n_clusters == dump["n_clusters"]
  descend = lap in (0, 1)
  idx = so.consumed_eigen_indices(n, 6, descend, dump["eigenvalues"] if descend else None,
                                  1e-2 if descend else None)
  w = diag.eigenvalue_array()
  ref = dump["eigenvalues"]
  # EVERY consumed eigenvalue on the north-star bar (rounds 3-5 held the ones that cannot move
  # the eigengap decision to 5e-3 here)
  for i in idx:
    assert abs(w[i] - ref[i]) <= 1e-5 * max(abs(ref[i]), 1e-12), (i, w[i], ref[i])
  np.testing.assert_allclose(diag.max_delta, dump["max_delta"], rtol=1e-5)
  assert so.adjusted_rand_index(got, want) == 1.0


# --- block Arnoldi where it is the default (n > 512) on the consumed-eigenvalue bar -----------
# [RowWiseThreshold]-only sequences x every Laplacian x both eigengap rules
# against the oracle (np.linalg.eig), ALL consumed eigenvalues <= 1e-5 relative -- the shape of
# the reference's own AutoTune tests (tests/spectral_clusterer_test.py:156-241, utils.py:59,
# 100-128).  One np.linalg.eig per (n, laplacian): the second eigengap rule re-reads its values.
_ORACLE_DUMPS = {}


def _threshold_only_oracle(n, lap, gap_code, maxc):
  key = (n, lap)
  if key not in _ORACLE_DUMPS:
    x = so.blobs(n, 32, 4, seed=7 * n + lap)
    cfg = threshold_only_config(laplacian_type=lap, p_percentile=0.9, min_clusters=2,
                                max_clusters=maxc, row_wise_renorm=False)
    dump = {}
    want = so.predict(x, cfg, dump)
    _ORACLE_DUMPS[key] = (x, dump["eigenvalues"], want)
  x, w, want = _ORACLE_DUMPS[key]
  descend = lap in (0, 1)
  k, delta = so.eigengap(w, maxc, 1e-2 if descend else None, gap_code, descend=descend) \
      if descend else so.eigengap(w, maxc, eigengap_type=gap_code, descend=False)
  return x, w, k, delta, want


@pytest.mark.parametrize("gap", ["Ratio", "NormalizedDiff"])
@pytest.mark.parametrize("lap", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [600, 1000, 2000])
def test_block_arnoldi_every_consumed_eigenvalue_vs_oracle(n, lap, gap):
  maxc = 8
  gap_code = so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF
  x, ref, k_ref, delta_ref, want = _threshold_only_oracle(n, lap, gap_code, maxc)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc,
      refinement_options=threshold_only_options(p_percentile=0.9),
      laplacian_type=sca.LaplacianType(lap) if lap else None,
      eigengap_type=getattr(sca.EigenGapType, gap))
  got = clusterer.predict(x)
  diag = clusterer.last_diag
  descend = lap in (0, 1)
  assert diag.symmetry_state == 3
  # ascending NormalizedDiff reads np.max(eigenvalues), the far end of the spectrum
  # (utils.py:110,123): the main solve's Krylov space has that to 1e-3 at best, so it gets a
  # block Arnoldi solve of its own (the operator with its sign turned, one eigenvalue, to its
  # residual); only a far-end solve that spends its budget goes to the dense route (fallback 9)
  far_end = (not descend) and gap == "NormalizedDiff"
  assert diag.eig_path == 4 or (far_end and diag.eig_path == 7 and diag.eig_fallback == 9), (
      diag.eig_path, diag.eig_fallback)
  assert diag.n_clusters_raw == k_ref
  w = clusterer.consumed_eigenvalues()   # (the dense route reports all n, far end included)
  idx = so.consumed_eigen_indices(n, maxc, descend, ref if descend else None,
                                  1e-2 if descend else None, gap_code)
  assert (n - 1 in idx) == far_end
  for i in idx:
    if i >= w.size:   # block Arnoldi: the far end is held through max_delta = gap / far end below
      assert far_end and i == n - 1 and diag.eig_path == 4
      continue
    assert abs(w[i] - ref[i]) <= 1e-5 * max(abs(ref[i]), 1e-12), (i, w[i], ref[i])
  # (max_delta = (w[k] - w[k - 1]) / np.max(w): with both gap eigenvalues on the bar above, 1e-5
  #  on it is 1e-5 on the far end)
  np.testing.assert_allclose(diag.max_delta, delta_ref, rtol=1e-5)
  if gap == "Ratio":   # (labels of the oracle's predict(), which ran with the Ratio rule)
    assert so.adjusted_rand_index(got, want) == 1.0


def test_row_wise_normalize_after_threshold_is_general_too():
  """[Threshold, RowWiseNormalize]: the normalisation is materialised (no symmetric fold)."""
  x = so.blobs(400, 24, 4, seed=9)
  seq = (so.OP_ROW_WISE_THRESHOLD, so.OP_ROW_WISE_NORMALIZE)
  cfg = so.OracleConfig(min_clusters=2, max_clusters=7, sequence=seq, p_percentile=0.92)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=7,
      refinement_options=sca.RefinementOptions(
          p_percentile=0.92, refinement_sequence=[sca.RefinementName.RowWiseThreshold,
                                                  sca.RefinementName.RowWiseNormalize]))
  got = clusterer.predict(x)
  assert clusterer.last_diag.symmetry_state == 3
  assert so.adjusted_rand_index(got, want) == 1.0


def test_general_path_autotune_vs_oracle():
  x = so.blobs(600, 32, 5, seed=21)
  cfg = threshold_only_config(min_clusters=2, max_clusters=8)
  want = so.predict(x, cfg, autotune=(0.6, 0.95, 0.05, 1, True))
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=8, refinement_options=threshold_only_options(),
      autotune=sca.AutoTune(p_percentile_min=0.60, p_percentile_max=0.95,
                            init_search_step=0.05, search_level=1),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  got = clusterer.predict(x)
  assert so.adjusted_rand_index(got, want) == 1.0


@pytest.mark.parametrize("n", [60, 300])
def test_general_matrix_autotune_vs_reference_golden(n):
  """Outputs of the real reference (oracle/make_golden.py --general): per-p cluster counts,
  maximum gaps and the AutoTune winner's labels on the non-symmetric path."""
  g = golden("general_n%d.npz" % n)
  x = so.blobs(n, int(g["d"]), int(g["k"]), int(g["seed"]))
  def make():
    return sca.SpectralClusterer(
        min_clusters=2, max_clusters=6, refinement_options=threshold_only_options(),
        autotune=sca.AutoTune(p_percentile_min=0.60, p_percentile_max=0.95,
                              init_search_step=0.05, search_level=1),
        laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  clusterer = make()
  a = so.affinity(x)
  for i, p in enumerate(g["grid"]):
    clusterer.refinement_options.p_percentile = float(p)
    _, k, delta = clusterer._compute_eigenvectors_ncluster(a)
    assert k == int(g["n_clusters"][i])
    np.testing.assert_allclose(delta, g["max_delta"][i], rtol=1e-5)
    diag = clusterer.last_diag
    w = diag.eigenvalue_array()
    for j in (k - 1, k):   # the pair that forms the maximum gap
      assert abs(w[j] - g["eigenvalues"][i][j]) <= 1e-5 * abs(g["eigenvalues"][i][j])
  labels = make().predict(x)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


# --- randomised sweep over non-symmetric configurations ------------------------------------
SEQUENCES = {
    "thr": (so.OP_ROW_WISE_THRESHOLD,),
    "crop_blur_thr": (so.OP_CROP_DIAGONAL, so.OP_GAUSSIAN_BLUR, so.OP_ROW_WISE_THRESHOLD),
    "thr_norm": (so.OP_ROW_WISE_THRESHOLD, so.OP_ROW_WISE_NORMALIZE),
    "sym_thr": (so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE, so.OP_ROW_WISE_THRESHOLD),
    "diffuse_thr": (so.OP_CROP_DIAGONAL, so.OP_DIFFUSE, so.OP_ROW_WISE_THRESHOLD),
}


def _general_fuzz_cases():
  rng = np.random.default_rng(77)
  cases = []
  for i in range(30):
    n = int(rng.choice([int(rng.integers(12, 64)), int(rng.integers(65, 900))]))
    d = int(rng.integers(4, 40))
    k = int(rng.integers(2, 6))
    lap = int(rng.choice([0, 1, 2, 3, 4]))
    gap = str(rng.choice(["Ratio", "Ratio", "NormalizedDiff"]))
    seq = str(rng.choice(list(SEQUENCES)))
    p = float(rng.choice([0.95, 0.85, 0.6, 0.4]))
    ttype = int(rng.choice([so.THRESHOLD_ROW_MAX, so.THRESHOLD_PERCENTILE]))
    binarize = bool(rng.integers(0, 2))
    maxc = int(rng.choice([5, 8, 12]))
    noise = float(rng.choice([0.2, 0.4]))
    cases.append((i, n, d, k, lap, gap, seq, p, ttype, binarize, maxc, noise))
  return cases


@pytest.mark.parametrize("case", _general_fuzz_cases(), ids=lambda c: "gfuzz%d" % c[0])
def test_fuzz_general_path_vs_oracle(case):
  i, n, d, k, lap, gap, seq, p, ttype, binarize, maxc, noise = case
  x = so.blobs(n, d, k, seed=5000 + i, noise=noise)
  cfg = so.OracleConfig(
      min_clusters=2, max_clusters=maxc, sequence=SEQUENCES[seq], gaussian_blur_sigma=1,
      p_percentile=p, threshold_type=ttype, binarize=binarize, laplacian_type=lap,
      eigengap_type=so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF)
  dump = {}
  want = so.predict(x, cfg, dump)
  names = {so.OP_CROP_DIAGONAL: "CropDiagonal", so.OP_GAUSSIAN_BLUR: "GaussianBlur",
           so.OP_ROW_WISE_THRESHOLD: "RowWiseThreshold", so.OP_SYMMETRIZE: "Symmetrize",
           so.OP_DIFFUSE: "Diffuse", so.OP_ROW_WISE_NORMALIZE: "RowWiseNormalize"}
  options = sca.RefinementOptions(
      gaussian_blur_sigma=1, p_percentile=p,
      thresholding_type=sca.ThresholdType(ttype), thresholding_with_binarization=binarize,
      refinement_sequence=[getattr(sca.RefinementName, names[op]) for op in SEQUENCES[seq]])
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc, refinement_options=options,
      laplacian_type=sca.LaplacianType(lap) if lap else None,
      eigengap_type=getattr(sca.EigenGapType, gap))
  got = clusterer.predict(x)
  diag = clusterer.last_diag
  assert diag.symmetry_state == 3
  if dump["max_delta"] < 1e-9:
    # exactly degenerate spectrum (e.g. binarised threshold of a diffused matrix: eigenvalues
    # 0, 1, 1, 1, ...): every gap is rounding noise, in the reference too -- nothing to match
    assert got.shape == (n,)
    return
  assert diag.n_clusters == dump["n_clusters"]
  np.testing.assert_allclose(diag.max_delta, dump["max_delta"], rtol=1e-5)
  assert so.adjusted_rand_index(got, want) == 1.0


# --- more than 32 eigenpairs of a non-symmetric matrix (wide block Arnoldi) ------------------
def test_general_matrix_more_than_32_eigenpairs_vs_reference_golden():
  """[RowWiseThreshold] + GraphCut, n = 400, min_clusters = 40, max_clusters = 48: the
  reference embeds in 40 eigenvectors of a non-symmetric matrix and reads 49 eigenvalues
  (golden from the real reference: oracle/make_golden.py --general-wide).  The device takes the
  WIDE block Arnoldi: basis up to 128, projected problems on the host (host_general_eig)."""
  g = golden("general_wide_n400.npz")
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  c = sca.SpectralClusterer(
      min_clusters=int(g["min_clusters"]), max_clusters=maxc,
      refinement_options=threshold_only_options(p_percentile=float(g["p_percentile"])),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = c.predict(x)
  dg = c.last_diag
  # general matrix; n = 400 <= 512: the dense Hessenberg route by default since round 5 (the wide
  # block Arnoldi keeps this golden under SC_GEN_DENSE_MAX_N=64: test_gpu_alternate_paths.py)
  assert dg.symmetry_state == 3 and dg.eig_path == 7 and dg.eig_fallback == 8
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  assert dg.n_clusters == int(g["min_clusters"])
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = c.consumed_eigenvalues()
  assert w.size >= maxc + 1  # (the dense route reports the whole spectrum)
  ref = g["head_eigenvalues"][:maxc + 1]
  # ascending branch: w[1 .. maxc - 1] are read (utils.py:104-115)
  idx = np.arange(1, maxc)
  err = np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max())
  assert err.max() < 1e-5, err.max()
  assert np.unique(labels).size == np.unique(g["labels"]).size > 32
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


@pytest.mark.parametrize("n,count", [(300, 40), (1500, 50)])
def test_arnoldi_more_than_32_eigenpairs_vs_numpy(n, count):
  m, _ = thresholded(n, 24, 12, seed=n + 1)
  assert not np.allclose(m, m.T)
  w, v = sca.utils.compute_sorted_eigenvectors(m, descend=True, count=count)
  ev, _ = np.linalg.eig(m)
  order = np.argsort(-ev.real, kind="stable")
  ev = ev[order]
  np.testing.assert_allclose(w, ev.real[:count], rtol=1e-7, atol=1e-8 * np.abs(ev).max())
  # the returned vectors are REAL PARTS (utils.py:61): an eigenvector only where the
  # eigenvalue is real -- check the residual there
  for j in range(count):
    if abs(ev[j].imag) == 0.0:
      r = m @ v[:, j] - w[j] * v[:, j]
      assert np.abs(r).max() < 1e-6 * np.abs(ev).max(), j


def test_arnoldi_64_separated_eigenpairs_of_a_nonnormal_matrix():
  """The widest request the general path takes at n > 64: 64 pairs.  (A thresholded affinity
  has 64 of its eigenvalues deep in a dense bulk, where a 128-vector Krylov basis stalls at a
  residual of ~1e-9 against the stage API's 1e-10 bar: the matrix here has 64 separated
  leading eigenvalues, a bulk behind them and non-orthogonal eigenvectors.)"""
  rng = np.random.default_rng(64)
  n, count = 700, 64
  lam = np.concatenate([np.linspace(10.0, 2.0, count), rng.uniform(-1.0, 1.0, n - count)])
  x = np.eye(n) + 0.3 * rng.standard_normal((n, n)) / np.sqrt(n)
  m = np.ascontiguousarray((x * lam) @ np.linalg.inv(x))
  assert not np.allclose(m, m.T)
  w, v = sca.utils.compute_sorted_eigenvectors(m, descend=True, count=count)
  np.testing.assert_allclose(w, lam[:count], rtol=1e-8)
  r = m @ v - v * w
  assert np.abs(r).max() < 1e-7
  np.testing.assert_allclose(np.linalg.norm(v, axis=0), 1.0, rtol=1e-10)


def test_general_path_has_no_pair_limit_any_more():
  """Rounds 1-4 refused more than 64 eigenpairs of a non-symmetric matrix for n > 64
  (UnsupportedOnDeviceError); the dense Hessenberg route now serves them."""
  m, _ = thresholded(300, 16, 3, seed=5)
  w, _ = sca.utils.compute_sorted_eigenvectors(m, descend=True, count=65)
  ev = np.linalg.eigvals(m)
  ev = ev[np.argsort(-ev.real, kind="stable")]
  np.testing.assert_allclose(w, ev.real[:65], rtol=1e-8, atol=1e-9 * np.abs(ev).max())


# --- the dense Hessenberg route: the WHOLE spectrum of a non-symmetric matrix, n > 64 --------
def _dense_case(name):
  g = golden(name)
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  ttype = sca.ThresholdType.Percentile if int(g["percentile"]) else sca.ThresholdType.RowMax
  opts = sca.RefinementOptions(
      p_percentile=float(g["p_percentile"]), thresholding_soft_multiplier=0.01,
      thresholding_type=ttype, refinement_sequence=[sca.RefinementName.RowWiseThreshold])
  c = sca.SpectralClusterer(
      min_clusters=int(g["min_clusters"]), max_clusters=None if maxc < 0 else maxc,
      refinement_options=opts,
      laplacian_type={0: None, 4: sca.LaplacianType.GraphCut}[lap])
  return g, x, c, n, lap, maxc


@pytest.mark.parametrize("name", ["general_dense_n300_lap4.npz", "general_dense_n1000_lap4.npz"])
def test_max_clusters_none_with_laplacian_on_a_general_matrix_vs_reference_golden(name):
  """[RowWiseThreshold] + GraphCut with max_clusters=None: the ascending eigengap loop reads EVERY
  eigenvalue of a matrix that is not diagonally similar to a symmetric one (utils.py:59,
  100-115).  Rounds 1-4 raised here for n > 64; now: Hessenberg reduction on the device, QR
  iteration + inverse iteration on the host (eig_path 7).  Golden: the real reference
  (oracle/make_golden.py --general-dense)."""
  g, x, c, n, lap, maxc = _dense_case(name)
  labels = c.predict(x)
  dg = c.last_diag
  assert dg.symmetry_state == 3 and dg.eig_path == 7
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = c.consumed_eigenvalues()
  ref = g["eigenvalues"]
  assert w.size == n
  idx = np.arange(1, n)  # ascending branch: w[1 .. n - 1] are read
  err = np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max())
  assert err.max() < 1e-5, err.max()
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


def test_more_than_64_eigenpairs_of_a_general_matrix_vs_reference_golden():
  """max_clusters = 80 and min_clusters = 66 on a non-symmetric refined matrix: 81 eigenvalues
  read, 66 eigenvectors (complex pairs among them) embedded.  Rounds 1-4 raised beyond 64."""
  g, x, c, n, lap, maxc = _dense_case("general_dense_n500_max80.npz")
  labels = c.predict(x)
  dg = c.last_diag
  assert dg.symmetry_state == 3 and dg.eig_path == 7
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  assert dg.n_clusters == int(g["min_clusters"]) == 66
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = c.consumed_eigenvalues()
  ref = g["eigenvalues"]
  idx = np.arange(1, maxc)
  err = np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max())
  assert err.max() < 1e-5, err.max()
  assert np.unique(labels).size == np.unique(g["labels"]).size
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


def test_descending_request_that_reads_past_64_values_of_a_general_matrix():
  """[RowWiseThreshold] without a Laplacian and max_clusters=None: the descending loop reads on
  until an eigenvalue falls below stop_eigenvalue (181 values here, utils.py:116-128): the wide
  block Arnoldi finds out and hands over to the dense route."""
  g, x, c, n, lap, maxc = _dense_case("general_dense_n300_lap0.npz")
  labels = c.predict(x)
  dg = c.last_diag
  assert dg.symmetry_state == 3 and dg.eig_path == 7
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = c.consumed_eigenvalues()
  ref = g["eigenvalues"]
  idx = so.consumed_eigen_indices(n, None, True, ref, 1e-2)
  assert idx.size > 64
  err = np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max())
  assert err.max() < 1e-5, err.max()
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


@pytest.mark.parametrize("n,count,descend", [(150, 150, True), (700, 100, False), (2100, 70, True)])
def test_dense_general_route_vs_numpy(n, count, descend):
  """sc_stage_eig with more than 64 pairs of a non-symmetric matrix: eigenvalues against
  np.linalg.eig (all of them when count = n), residuals of the real eigenvectors."""
  m, _ = thresholded(n, 24, 12, seed=n + 7)
  assert not np.allclose(m, m.T)
  w, v = sca.utils.compute_sorted_eigenvectors(m, descend=descend, count=count)
  ev = np.linalg.eigvals(m)
  order = np.argsort(-ev.real if descend else ev.real, kind="stable")
  ev = ev[order]
  scale = np.abs(ev).max()
  np.testing.assert_allclose(w, ev.real[:count], rtol=1e-8, atol=1e-9 * scale)
  checked = 0
  for j in range(count):
    if abs(ev[j].imag) == 0.0:
      r = m @ v[:, j] - w[j] * v[:, j]
      assert np.abs(r).max() < 1e-8 * scale, j
      assert abs(np.linalg.norm(v[:, j]) - 1.0) < 1e-10
      checked += 1
  assert checked > 0


def test_block_arnoldi_that_spends_its_restart_budget_lands_on_the_dense_route():
  """np.linalg.eig always returns (utils.py:59): with a zero restart budget the block Arnoldi of a
  slowly converging request gives up at its first full basis and the dense route takes over --
  same eigengap decision and labels as with the default budget."""
  g = golden("general_wide_n400.npz")
  n, d, k, seed, lap, maxc = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  c = sca.SpectralClusterer(
      min_clusters=int(g["min_clusters"]), max_clusters=maxc,
      refinement_options=threshold_only_options(p_percentile=float(g["p_percentile"])),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  c.eig_max_cycles = -1
  labels = c.predict(x)
  dg = c.last_diag
  # (n = 400: the dense route by default; the budget only matters above SC_GEN_DENSE_MAX_N)
  assert dg.eig_path == 7 and dg.eig_fallback in (1, 8)
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
