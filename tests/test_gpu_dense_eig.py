"""GPU: the dense full-spectrum symmetric eigenvalue path (E1; eig_dense.hip: Householder
tridiagonalisation + Sturm bisection, vectors from block Lanczos) through the C ABI.

  * every eigenvalue vs numpy's eigvalsh on the same matrix (stage API, values only);
  * the reference's DEFAULT max_clusters=None with a Laplacian, where the eigengap reads
    all n eigenvalues (spectral_clusterer.py:32, utils.py:100-115), and the ascending
    NormalizedDiff gap with its np.max(eigenvalues) (utils.py:110) -- against goldens
    produced by the real reference (oracle/make_golden.py --dense) and against the oracle.
"""

import ctypes
import glob
import os

import numpy as np
import pytest

import spectral_oracle as so
import spectralcluster_amd as sca
from spectralcluster_amd import _lib
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

LAP = {0: None, 1: sca.LaplacianType.Affinity, 2: sca.LaplacianType.Unnormalized,
       3: sca.LaplacianType.RandomWalk, 4: sca.LaplacianType.GraphCut}


def icassp_options(sigma=1, p=0.95):
  return sca.RefinementOptions(
      gaussian_blur_sigma=sigma, p_percentile=p, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.RowMax,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


def all_eigenvalues(handle, m, descend):
  n = m.shape[0]
  m = np.ascontiguousarray(m, dtype=np.float64)
  values = np.empty(n)
  diag = _lib.ScDiag()
  handle.check(handle.lib.sc_stage_sym_eig(handle.raw, _lib.as_double_p(m), n, n,
                                           int(descend), _lib.as_double_p(values), None, diag))
  return values, diag


@pytest.mark.parametrize("n", [129, 200, 333, 512, 1000])
def test_all_eigenvalues_vs_eigvalsh(handle, n):
  rng = np.random.default_rng(n)
  a = rng.standard_normal((n, n))
  a = 0.5 * (a + a.T)
  want = np.linalg.eigvalsh(a)
  scale = np.abs(want).max()
  got, diag = all_eigenvalues(handle, a, descend=False)
  assert diag.eig_path == 5  # SC_EIG_PATH_DENSE_TRIDIAG
  assert np.max(np.abs(got - want)) < 1e-12 * scale * np.sqrt(n)
  got_d, _ = all_eigenvalues(handle, a, descend=True)
  assert np.array_equal(got_d, got[::-1])


def test_structured_spectra(handle):
  n = 300
  rng = np.random.default_rng(1)
  # diagonal matrix: every reflector is the identity (tau = 0 path)
  d = rng.standard_normal(n)
  got, _ = all_eigenvalues(handle, np.diag(d), False)
  np.testing.assert_allclose(got, np.sort(d), rtol=0, atol=1e-14)
  # rank-3 PSD matrix: n - 3 eigenvalues are (numerically) zero -- repeated eigenvalues
  b = rng.standard_normal((n, 3))
  got, _ = all_eigenvalues(handle, b @ b.T, False)
  want = np.linalg.eigvalsh(b @ b.T)
  assert np.max(np.abs(got - want)) < 1e-12 * want[-1]
  # graded spectrum over 12 decades (absolute accuracy ~ ulp * ||A||)
  q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  lam = np.logspace(-12, 0, n)
  a = (q * lam) @ q.T
  a = 0.5 * (a + a.T)
  got, _ = all_eigenvalues(handle, a, False)
  assert np.max(np.abs(got - np.linalg.eigvalsh(a))) < 1e-13
  # tridiagonal input (Toeplitz -1 2 -1): known closed form
  t = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
  got, _ = all_eigenvalues(handle, t, False)
  want = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


def _dense_goldens():
  return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "dense_n*.npz")))


@pytest.mark.parametrize("name", _dense_goldens())
def test_every_eigenvalue_consumed_vs_reference(name):
  g = dict(np.load(os.path.join(GOLDEN, name)))
  n, d, k, seed, lap, maxc, gt = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc or None, refinement_options=icassp_options(),
      laplacian_type=LAP[lap],
      eigengap_type=sca.EigenGapType.Ratio if gt == 1 else sca.EigenGapType.NormalizedDiff)
  labels = clusterer.predict(x)
  diag = clusterer.last_diag
  ref_w = g["eigenvalues"]
  descend = lap in (0, 1)
  idx = so.consumed_eigen_indices(n, maxc or None, descend, ref_w, 1e-2,
                                  so.EIGENGAP_RATIO if gt == 1 else so.EIGENGAP_NORMALIZED_DIFF)
  w = clusterer.consumed_eigenvalues()
  if not descend:
    assert diag.eig_path == 5 and w.shape[0] == n   # the whole spectrum, like np.linalg.eig
    # the ~1e-17 null eigenvalue has no relative accuracy; everything the eigengap reads does
    assert abs(w[0]) < 1e-9
  assert idx.max() < w.shape[0]
  rel = np.abs(w[idx] - ref_w[idx]) / np.maximum(np.abs(ref_w[idx]), 1e-12)
  assert rel.max() < 1e-5, (rel.max(), idx[np.argmax(rel)])   # north-star bar, ALL consumed
  assert diag.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(diag.max_delta, float(g["max_delta"]), rtol=1e-6)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


@pytest.mark.parametrize("lap,n", [(4, 150), (2, 400), (3, 257), (4, 640)])
def test_default_max_clusters_vs_oracle(lap, n):
  x = so.blobs(n, 24, 4, seed=77 + n, noise=0.4)
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=None)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(min_clusters=2, refinement_options=icassp_options(),
                                    laplacian_type=LAP[lap])
  got = clusterer.predict(x)
  w = clusterer.consumed_eigenvalues()
  ref = dump["eigenvalues"]
  assert w.shape[0] == n
  assert np.max(np.abs(w[1:] - ref[1:]) / np.maximum(np.abs(ref[1:]), 1e-12)) < 1e-5
  assert max(clusterer.last_diag.n_clusters_raw, 2) == dump["n_clusters"]
  assert so.adjusted_rand_index(got, want) == 1.0
  # _compute_eigenvectors_ncluster on the same affinity: same decision, usable vectors
  vecs, k, delta = clusterer._compute_eigenvectors_ncluster(so.affinity(x))
  assert max(k, 2) == dump["n_clusters"] and vecs.shape[0] == n
  np.testing.assert_allclose(delta, dump["max_delta"], rtol=1e-6)


def test_descending_without_max_clusters_many_values(handle):
  """laplacian_type=None, max_clusters=None: the loop reads every eigenvalue down to
  stop_eigenvalue (utils.py:117-128).  With a slowly decaying spectrum that is more than a
  Krylov basis holds -> dense path for the values."""
  n = 400
  x = so.blobs(n, 200, 3, seed=11)   # unrefined affinity: ~200 eigenvalues above 1e-2
  cfg = so.OracleConfig(sequence=(), stop_eigenvalue=1e-2)
  dump = {}
  want = so.predict(x, cfg, dump)
  ref = dump["eigenvalues"]
  idx = so.consumed_eigen_indices(n, None, True, ref, 1e-2)
  assert idx.size > 64
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=None)
  got = clusterer.predict(x)
  assert clusterer.last_diag.eig_path == 5
  w = clusterer.consumed_eigenvalues()
  assert np.max(np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-12)) < 1e-5
  assert max(clusterer.last_diag.n_clusters_raw, 2) == dump["n_clusters"]
  assert so.adjusted_rand_index(got, want) == 1.0


@pytest.mark.parametrize("lap,maxc", [(4, 100), (0, 150), (4, 500)])
def test_max_clusters_above_64_vs_oracle(lap, maxc):
  """The reference accepts any max_clusters (spectral_clusterer.py:29-46).  More than 64
  consumed eigenvalues do not fit a Krylov basis: they come from the dense path; what stays
  limited on the device is the SELECTED cluster count (<= 64 eigenvector columns)."""
  n = 900
  x = so.blobs(n, 32, 6, seed=900 + maxc, noise=0.4)
  cfg = so.icassp2018_config(laplacian_type=lap, max_clusters=maxc)
  dump = {}
  want = so.predict(x, cfg, dump)
  clusterer = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc,
                                    refinement_options=icassp_options(),
                                    laplacian_type=LAP[lap])
  got = clusterer.predict(x)
  ref = np.real(dump["eigenvalues"])
  idx = so.consumed_eigen_indices(n, maxc, lap == 0, ref, 1e-2)
  w = clusterer.consumed_eigenvalues()
  assert idx.max() < w.shape[0]
  rel = np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-9 * np.abs(ref).max())
  assert rel.max() < 1e-5
  assert max(clusterer.last_diag.n_clusters_raw, 2) == dump["n_clusters"]
  assert so.adjusted_rand_index(got, want) == 1.0
