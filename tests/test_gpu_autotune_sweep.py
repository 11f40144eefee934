"""One AutoTune search level as a group (sc_eig_ncluster_sweep): every p_percentile of the level
gets what its own sc_eig_ncluster call reports -- eigengap decision, proxy input, consumed
eigenvalues -- and predict() with AutoTune picks the same p and labels as before."""
import numpy as np
import pytest

import spectralcluster_amd as sca
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


def clusterer(lap, max_clusters, **kw):
  return sca.SpectralClusterer(
      min_clusters=2, max_clusters=max_clusters, laplacian_type=lap,
      refinement_options=sca.RefinementOptions(
          gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
          refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE), **kw)


@pytest.mark.parametrize("n,lap,count", [(700, None, 9), (1500, sca.LaplacianType.GraphCut, 20),
                                         (4100, sca.LaplacianType.GraphCut, 5)])
def test_sweep_reports_what_single_evaluations_report(n, lap, count):
  x = so.blobs(n, 64, 5, seed=n)
  c = clusterer(lap, 7 if lap is None else 12)
  handle = c._handle()
  c._upload(handle, x)
  ps = [float(p) for p in np.linspace(0.55, 0.95, count)]
  diags = c._eig_sweep(handle, ps)
  for p, d in zip(ps, diags):
    one = c._eig_resident(handle, p)
    assert d.n_clusters_raw == one.n_clusters_raw, p
    assert abs(d.max_delta - one.max_delta) <= 1e-6 * abs(one.max_delta), p
    w, w1 = d.eigenvalue_array(), one.eigenvalue_array()
    idx = so.consumed_eigen_indices(n, c.max_clusters, lap is None, w1, 1e-2)
    np.testing.assert_allclose(w[idx], w1[idx], rtol=2e-6)
    assert d.eig_path == 2 and d.n == n


def test_sweep_falls_back_for_sequences_it_does_not_cover():
  x = so.blobs(600, 32, 3, seed=6)
  opts = sca.RefinementOptions(
      p_percentile=0.9, thresholding_soft_multiplier=0.01,
      refinement_sequence=[sca.RefinementName.RowWiseThreshold, sca.RefinementName.Symmetrize,
                           sca.RefinementName.Diffuse, sca.RefinementName.RowWiseNormalize])
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
  handle = c._handle()
  c._upload(handle, x)
  ps = [0.6, 0.8, 0.95]
  for p, d in zip(ps, c._eig_sweep(handle, ps)):
    one = c._eig_resident(handle, p)
    assert d.n_clusters_raw == one.n_clusters_raw and d.max_delta == one.max_delta


@pytest.mark.parametrize("lap", [None, sca.LaplacianType.GraphCut])
def test_autotune_predict_vs_oracle(lap):
  """two search levels through the grouped sweep: same winner and labels as the oracle's
  one-by-one search (reference autotune.py:76-132)"""
  x = so.blobs(900, 48, 4, seed=17)
  c = clusterer(lap, 8, autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                              init_search_step=0.05, search_level=2))
  got = c.predict(x)
  cfg = so.icassp2018_config(laplacian_type=0 if lap is None else 4, max_clusters=8)
  vecs, k, best_p, seen = so.autotune_search(so.affinity(x), cfg, 0.55, 0.95, 0.05,
                                             search_level=2)
  k = max(k, 2)
  want = so.run_kmeans(vecs[:, :k], k, 300)
  assert so.adjusted_rand_index(got, want) == 1.0
  assert c.last_diag.n_clusters == k


@pytest.mark.parametrize("n,lap", [(900, None), (2000, sca.LaplacianType.GraphCut)])
def test_adopted_eigenvectors_equal_a_fresh_evaluation(n, lap):
  """sc_sweep_adopt: the eigenvectors a sweep left in a member arena span what
  sc_eig_ncluster with that p_percentile computes (the AutoTune winner is no longer
  evaluated a second time) -- same eigenvalues, same labels; refused for a configuration
  that is not the sweep's, and after a new affinity."""
  from spectralcluster_amd import _lib
  x = so.blobs(n, 64, 5, seed=n + 3)
  c = clusterer(lap, 7 if lap is None else 12)
  handle = c._handle()
  c._upload(handle, x)
  ps = [0.6, 0.7, 0.8, 0.9, 0.95]
  diags = c._eig_sweep(handle, ps)
  for i in (1, 4):
    dg = _lib.ScDiag()
    assert handle.lib.sc_sweep_adopt(handle.raw, c.build_config(ps[i]), i, dg) == _lib.SC_OK
    assert dg.n_clusters_raw == diags[i].n_clusters_raw and dg.max_delta == diags[i].max_delta
    k = max(int(dg.n_clusters_raw), 2)
    adopted = c._download_eigenvectors(handle, n, k)
    labels_a = np.empty(n, dtype=np.int64)
    handle.check(handle.lib.sc_cluster(handle.raw, c.build_config(ps[i]), k,
                                       _lib.as_int64_p(labels_a), dg))
    one = c._eig_resident(handle, ps[i])
    fresh = c._download_eigenvectors(handle, n, k)
    labels_f = np.empty(n, dtype=np.int64)
    handle.check(handle.lib.sc_cluster(handle.raw, c.build_config(ps[i]), k,
                                       _lib.as_int64_p(labels_f), one))
    cos = np.abs(np.einsum("ij,ij->j", adopted, fresh))
    np.testing.assert_allclose(cos, 1.0, atol=1e-7)
    assert so.adjusted_rand_index(labels_a, labels_f) == 1.0
  dg = _lib.ScDiag()
  other = c.build_config(ps[1])
  other.stop_eigenvalue = 0.5  # not the sweep's configuration
  assert handle.lib.sc_sweep_adopt(handle.raw, other, 1, dg) == _lib.SC_ERR_UNSUPPORTED
  assert handle.lib.sc_sweep_adopt(handle.raw, c.build_config(0.61), 0, dg) == _lib.SC_ERR_UNSUPPORTED
  c._upload(handle, x)  # a new affinity: nothing of the old sweep may be adopted
  assert handle.lib.sc_sweep_adopt(handle.raw, c.build_config(ps[1]), 1, dg) == _lib.SC_ERR_UNSUPPORTED


@pytest.mark.parametrize("n", [2100])
def test_sweep_with_binarisation_on_both_diffuse_routes(n):
  """thresholding_with_binarization puts exact ones into the thresholded matrix: max|a| of the
  matrix-free route's quantiser is floored at 1 (the sweep once passed a floor of 0, so that
  sigma * 1 overflowed the 15-bit digits and was clamped silently -- ADVICE r4).  Every value of
  the level must report on the matrix-free route what the explicit fp64 product reports, and what
  a single evaluation reports."""
  from spectralcluster_amd import _lib
  x = so.blobs(n, 64, 5, seed=n)
  opts = sca.RefinementOptions(
      gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
      thresholding_with_binarization=True,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  ps = [float(p) for p in np.linspace(0.6, 0.95, 6)]
  out = {}
  for mode in (1, 2):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=12, refinement_options=opts,
                              laplacian_type=sca.LaplacianType.GraphCut)
    c.diffuse_mode = mode
    handle = c._handle()
    c._upload(handle, x)
    diags = c._eig_sweep(handle, ps)
    for p, d in zip(ps, diags):
      one = c._eig_resident(handle, p)
      assert d.n_clusters_raw == one.n_clusters_raw, (mode, p)
      assert abs(d.max_delta - one.max_delta) <= 1e-6 * abs(one.max_delta), (mode, p)
      want = (_lib.DIFFUSE_PATH_EXPLICIT,) if mode == 1 else (
          _lib.DIFFUSE_PATH_FREE, _lib.DIFFUSE_PATH_FREE_THEN_EXPLICIT)
      assert d.diffuse_path in want, (mode, p, d.diffuse_path)
    out[mode] = diags
  for p, de, df in zip(ps, out[1], out[2]):
    assert de.n_clusters_raw == df.n_clusters_raw, p
    np.testing.assert_allclose(df.max_delta, de.max_delta, rtol=1e-6)
    we, wf = de.eigenvalue_array(), df.eigenvalue_array()
    idx = so.consumed_eigen_indices(n, 12, False, we, 1e-2)
    np.testing.assert_allclose(wf[idx], we[idx], rtol=2e-6, atol=2e-6 * np.abs(we).max())
