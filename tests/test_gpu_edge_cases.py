"""GPU edge cases: tiny and boundary sizes, ragged dimensions, degenerate inputs,
arena reuse across sizes, determinism, API misuse."""

import numpy as np
import pytest

import spectral_oracle as so

import spectralcluster_amd as sca
from spectralcluster_amd import refinement as rf

pytestmark = pytest.mark.gpu


def icassp(sigma=1, p=0.95):
  return sca.RefinementOptions(
      gaussian_blur_sigma=sigma, p_percentile=p, thresholding_soft_multiplier=0.01,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


@pytest.mark.parametrize("n", [2, 3, 5, 16, 17, 127, 128, 129, 143, 144, 145, 160, 255, 256,
                               257])
def test_boundary_sizes_vs_oracle(n):
  """n around every internal threshold: dense Jacobi limit (128), basis cap (n - 8),
  GEMM / blur / symmetrize tile edges."""
  k = 2 if n < 40 else 3
  x = so.blobs(n, 7, k, seed=n)
  for lap, lt in ((0, None), (4, sca.LaplacianType.GraphCut)):
    cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=min(7, max(2, n - 1)))
    dump = {}
    want = so.predict(x, cfg, dump)
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=cfg.max_clusters,
                              refinement_options=icassp(), laplacian_type=lt)
    got = c.predict(x)
    assert so.adjusted_rand_index(got, want) == 1.0, (n, lap)
    idx = so.consumed_eigen_indices(n, cfg.max_clusters, lap == 0, dump["eigenvalues"], 1e-2)
    w = c.last_diag.eigenvalue_array()
    ref = dump["eigenvalues"][idx]
    assert np.max(np.abs(w[idx] - ref) / np.maximum(np.abs(ref), 1e-9)) < 1e-5


@pytest.mark.parametrize("d", [1, 2, 15, 16, 17, 300])
def test_feature_dimension_edges(d):
  n = 150
  rng = np.random.default_rng(d)
  x = rng.standard_normal((n, d)) + 3.0 * (np.arange(n)[:, None] % 2)  # no zero rows
  got = sca.utils.compute_affinity_matrix(x)
  np.testing.assert_allclose(got, so.affinity(x), rtol=0, atol=5e-15)


def test_arena_reuse_across_sizes():
  """One handle, sizes going up and down: stale data in padded / larger buffers must
  never leak into a smaller problem."""
  c = sca.configs.icassp2018_clusterer
  sizes = [700, 130, 1500, 64, 300, 1499, 131, 2100, 200]
  first = {}
  for rep in range(2):
    for n in sizes:
      x = so.blobs(n, 24, 3, seed=n)
      lab = c.predict(x)
      if rep == 0:
        first[n] = lab
        assert so.adjusted_rand_index(lab, so.predict(x, so.icassp2018_config())) == 1.0
      else:
        assert np.array_equal(lab, first[n])  # identical on the second visit


def test_identical_embeddings_and_duplicates():
  # two groups of exactly duplicated rows (rank-2 affinity: Krylov space collapses)
  a = np.tile(np.array([[1.0, 0.2, 0.0]]), (90, 1))
  b = np.tile(np.array([[0.0, 0.3, 1.0]]), (70, 1))
  x = np.vstack([a, b])
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=5, refinement_options=icassp())
  lab = c.predict(x)
  assert so.adjusted_rand_index(lab, [0] * 90 + [1] * 70) == 1.0
  want = so.predict(x, so.icassp2018_config(max_clusters=5))
  assert so.adjusted_rand_index(lab, want) == 1.0


def test_float32_and_noncontiguous_inputs():
  x64 = so.blobs(200, 16, 3, seed=1)
  c = sca.configs.icassp2018_clusterer
  want = c.predict(x64)
  assert np.array_equal(c.predict(np.asfortranarray(x64)), want)
  wide = np.zeros((200, 40))
  wide[:, ::2][:, :16] = x64
  assert np.array_equal(c.predict(wide[:, ::2][:, :16]), want)
  assert so.adjusted_rand_index(c.predict(x64.astype(np.float32)), want) == 1.0


def test_stage_ops_on_tiny_and_ragged():
  rng = np.random.default_rng(0)
  for n in (1, 2, 3, 9, 31, 33, 63, 65):
    m = rng.random((n, n))
    assert np.array_equal(rf.CropDiagonal().refine(m), so.crop_diagonal(m))
    for sigma in (1, 2, 3):
      assert np.array_equal(rf.GaussianBlur(sigma).refine(m), so.gaussian_blur(m, sigma))
    assert np.array_equal(rf.Symmetrize(rf.SymmetrizeType.Average).refine(m),
                          so.symmetrize(m, so.SYMMETRIZE_AVERAGE))
    np.testing.assert_allclose(rf.Diffuse().refine(m), so.diffuse(m), rtol=1e-13)
    assert np.array_equal(rf.RowWiseNormalize().refine(m), so.row_wise_normalize(m))


def test_large_sigma_uses_generic_blur():
  rng = np.random.default_rng(1)
  m = rng.random((300, 300))
  for sigma in (0.5, 1.5, 3.0, 5.0, 8.0):
    assert np.array_equal(rf.GaussianBlur(sigma).refine(m), so.gaussian_blur(m, sigma))


@pytest.mark.parametrize("n,sigma", [(300, 8.5), (300, 9.0), (257, 12.0), (100, 40.0),
                                     (37, 20.0), (640, 16.0)])
def test_blur_sigma_above_8_any_radius(n, sigma):
  """The reference has no limit on gaussian_blur_sigma (refinement.py:154-162).  A radius
  above 32 -- above n, even: scipy's `reflect` extension reflects as often as it takes --
  goes through the two-pass kernel with the weights resident in the handle
  (sc_set_blur_weights): bit-exact against the oracle, which is bit-exact against scipy."""
  rng = np.random.default_rng(int(n + 10 * sigma))
  m = rng.random((n, n))
  assert int(4 * sigma + 0.5) > 32
  assert np.array_equal(rf.GaussianBlur(sigma).refine(m), so.gaussian_blur(m, sigma))
  # and a smaller radius right after it on the same handle (the weight cache must notice)
  assert np.array_equal(rf.GaussianBlur(1).refine(m), so.gaussian_blur(m, 1))


def test_predict_with_sigma_above_8_vs_oracle():
  """End to end, single call and both batch forms (member arenas and pooled handles inherit
  the extended weights)."""
  x = so.blobs(400, 24, 3, seed=9)
  cfg = so.icassp2018_config(gaussian_blur_sigma=9.0, max_clusters=7)
  want = so.predict(x, cfg)
  opts = sca.RefinementOptions(gaussian_blur_sigma=9.0, p_percentile=0.95,
                               thresholding_soft_multiplier=0.01,
                               refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
  assert so.adjusted_rand_index(c.predict(x), want) == 1.0
  x2 = so.blobs(350, 24, 2, seed=10)
  want2 = so.predict(x2, cfg)
  for how in ({"group": 16}, {"streams": 2}):
    out = c.predict_batch([x, x2], **how)
    assert so.adjusted_rand_index(out[0], want) == 1.0
    assert so.adjusted_rand_index(out[1], want2) == 1.0


def test_fused_and_unfused_pipelines_agree():
  """predict() uses fused kernels (crop->blur, threshold+symmetrize); chaining the
  per-op stage entry points must give bit-identical refined matrices."""
  x = so.blobs(400, 32, 4, seed=3)
  a = sca.utils.compute_affinity_matrix(x)
  opts = icassp()
  m = a
  for name in sca.ICASSP2018_REFINEMENT_SEQUENCE[:-1]:   # up to Diffuse
    m = opts.get_refinement_operator(name).refine(m)
  ref = so.refine(so.affinity(x), so.icassp2018_config(sequence=so.ICASSP2018_SEQUENCE[:-1]))
  np.testing.assert_allclose(m, ref, rtol=1e-12)
  # and the end-to-end labels agree with the oracle (fused path)
  got = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts).predict(x)
  assert so.adjusted_rand_index(got, so.predict(x, so.icassp2018_config())) == 1.0


def test_compute_eigenvectors_ncluster_api():
  x = so.blobs(500, 20, 3, seed=4)
  a = so.affinity(x)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=icassp(),
                            laplacian_type=sca.LaplacianType.GraphCut)
  vecs, k, delta = c._compute_eigenvectors_ncluster(a)
  v_ref, k_ref, d_ref = so.eig_ncluster(a, so.icassp2018_config(laplacian_type=4))
  assert k == k_ref and vecs.shape[0] == 500 and vecs.shape[1] >= 8
  np.testing.assert_allclose(delta, d_ref, rtol=1e-6)
  np.testing.assert_allclose(np.linalg.norm(vecs[:, :k], axis=0), 1.0, atol=1e-12)
  cos = np.abs(np.einsum("ij,ij->j", vecs[:, :k], v_ref[:, :k]))
  np.testing.assert_allclose(cos, 1.0, atol=1e-7)
  with pytest.raises(ValueError):
    c._compute_eigenvectors_ncluster(np.zeros((3, 4)))


def test_min_clusters_exceeds_eigengap():
  x = so.blobs(300, 16, 2, seed=5)
  c = sca.SpectralClusterer(min_clusters=5, max_clusters=7, refinement_options=icassp())
  lab = c.predict(x)
  assert len(set(lab.tolist())) <= 5 and c.last_diag.n_clusters == 5
  want = so.predict(x, so.icassp2018_config(min_clusters=5))
  assert so.adjusted_rand_index(lab, want) == 1.0


def test_multi_stream_batch_is_deterministic():
  rng = np.random.default_rng(6)
  utts = [so.blobs(int(n), 32, 3, seed=i) for i, n in enumerate(rng.integers(130, 600, 24))]
  c = sca.configs.icassp2018_clusterer
  one = c.predict_batch(utts, streams=1)
  four = c.predict_batch(utts, streams=4)
  again = c.predict_batch(utts, streams=4)
  for a, b, d in zip(one, four, again):
    assert np.array_equal(a, b) and np.array_equal(b, d)


@pytest.mark.parametrize("n", [136, 200, 520])
def test_numerically_low_rank_operator(n):
  """Tight clusters + a threshold that removes nothing: the refined matrix has three
  eigenvalues of order n and the rest below 1e-5.  Op * V is then a block of condition
  > 1e8; its small columns come out of R^-1 entries that amplify the large columns' rounding
  against the basis, so the block must be projected against the basis once more (a
  regression: the Ritz values drifted above lambda_max and the solve never converged)."""
  rng = np.random.default_rng(n)
  counts = [n // 10, n // 5, 3 * n // 10, n - n // 10 - n // 5 - 3 * n // 10]
  base = np.concatenate([np.tile(row, (c, 1)) for row, c in zip(
      ([1.0, 0, 0, 0, 0, 0], [0, 1.0, 0, 0, 0, 0], [0, 0, 2.0, 0, 0, 0], [0, 0, 0, 1.0, 0, 0]),
      counts)])
  x = base + (rng.random((n, 6)) * 2 - 1) * 0.02
  opts = sca.RefinementOptions(gaussian_blur_sigma=0, p_percentile=0.2,
                               refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  clusterer = sca.SpectralClusterer(refinement_options=opts, stop_eigenvalue=0.01)
  got = clusterer.predict(x)
  cfg = so.icassp2018_config(min_clusters=None, max_clusters=None, gaussian_blur_sigma=0,
                             p_percentile=0.2, stop_eigenvalue=0.01)
  dump = {}
  want = so.predict(x, cfg, dump)
  assert clusterer.last_diag.n_clusters == dump["n_clusters"]
  assert so.adjusted_rand_index(got, want) == 1.0
  w = clusterer.last_diag.eigenvalue_array()
  k = dump["n_clusters"]
  np.testing.assert_allclose(w[:k], dump["eigenvalues"][:k], rtol=1e-6)


def test_non_finite_input_raises_like_numpy():
  """A zero embedding row makes the cosine affinity NaN (reference utils.py:33, no guard);
  np.linalg.eig then raises LinAlgError (a ValueError): "Array must not contain infs or
  NaNs".  The device path reports the same instead of iterating on garbage."""
  for n in (40, 300):                       # dense and Krylov paths
    x = so.blobs(n, 8, 3, seed=n)
    x[n // 2] = 0.0
    for clusterer in (
        sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                              refinement_options=sca.configs.icassp2018_refinement_options),
        sca.SpectralClusterer(max_clusters=5, laplacian_type=sca.LaplacianType.GraphCut,
                              refinement_options=sca.RefinementOptions(
                                  refinement_sequence=[sca.RefinementName.RowWiseThreshold]))):
      with pytest.raises(ValueError, match="infs or NaNs"):
        clusterer.predict(x)
    # the handle stays usable
    good = so.blobs(n, 8, 3, seed=n)
    labels = sca.configs.icassp2018_clusterer.predict(good)
    assert so.adjusted_rand_index(labels, so.predict(good, so.icassp2018_config())) == 1.0
  w, _ = sca.utils.compute_sorted_eigenvectors(np.diag([3.0, 2.0, 1.0]))
  np.testing.assert_allclose(w, [3.0, 2.0, 1.0])


@pytest.mark.parametrize("n,lap", [(4100, 4), (4233, 0)])
def test_ragged_size_above_the_symmetric_matvec_threshold(n, lap):
  """n >= 4096 takes the upper-triangle block matvec (128 x 128 tiles) and the persistent
  GEMM loop; an n that is not a multiple of 128 exercises their edge tiles.  Checked against
  the algorithm-matched CPU path (NumPy refinement + scipy eigsh on the symmetric operator,
  itself checked against the reference-shaped oracle in test_oracle_vs_golden.py), which
  finishes in seconds at this size."""
  x = so.blobs(n, 48, 5, seed=n)
  maxc = 20 if lap else 7
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=maxc)
  want, w_ref = so.predict_algorithm_matched(x, cfg)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc,
                            refinement_options=sca.configs.icassp2018_refinement_options,
                            laplacian_type=sca.LaplacianType.GraphCut if lap else None)
  got = c.predict(x)
  assert c.last_diag.eig_path == 2 and c.last_diag.eig_host_chain == 0
  idx = so.consumed_eigen_indices(n, maxc, lap == 0, w_ref, 1e-2)
  w = c.last_diag.eigenvalue_array()
  floor = 1e-9  # the Laplacian's null eigenvalue is not in idx; eigsh's tolerance is 1e-10
  assert np.max(np.abs(w[idx] - w_ref[idx]) / np.maximum(np.abs(w_ref[idx]), floor)) < 1e-6
  assert so.adjusted_rand_index(got, want) == 1.0
