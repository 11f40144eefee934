"""GPU: the matrix-free Diffuse (csrc/diffuse_free.hip, free_api.hip).

For a sequence whose Diffuse (reference refinement.py:229-234) is followed only by
RowWiseNormalize (:240-245) and the Laplacian (laplacian.py:41-58) the device never forms
S = A A^T: rowmax(S) comes from an exact 8-bit-digit integer MFMA product (a candidate
search within a proven slack) + an fp64 recheck, rowsum(S) = A (A 1), and the eigensolver
applies A twice per block.  Both routes must give what the reference gives:

  * stage level: rowmax / rowsum of a a^T through `sc_stage_diffuse_rowstats` -- matrix-free
    (mode 2) against the explicit fp64 product (mode 1) and against NumPy, on refined affinities
    and on inputs built to stress the search (near ties, exact ties, negative entries, a zero
    row, rows that are tiny against the rest, ragged sizes);
  * end to end: the reference goldens (labels, n_clusters, max_delta, consumed eigenvalues)
    with `diffuse_mode` 2 (matrix-free wherever the sequence allows) and 1 (explicit), and the
    default routing (matrix-free from n = 2048 on).
"""

import ctypes
import dataclasses

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden

import spectralcluster_amd as sca
from spectralcluster_amd import _lib

pytestmark = pytest.mark.gpu

LAP = {0: None, 2: sca.LaplacianType.Unnormalized, 3: sca.LaplacianType.RandomWalk,
       4: sca.LaplacianType.GraphCut}
EXPLICIT, FREE = 1, 2


def rowstats(a, mode, prune=None):
  """info: candidates evaluated, rows over the cap, largest candidate count, S formed after all,
  tiles of the digit product computed, tiles in its upper triangle.  prune: the tile skip list
  on (1) / off (0) for this call; None = the default (on)."""
  h = _lib.default_handle()
  a = np.ascontiguousarray(a, dtype=np.float64)
  n = a.shape[0]
  rmax, rsum = np.empty(n), np.empty(n)
  info = (ctypes.c_int32 * 6)()
  if prune is not None:
    h.check(h.lib.sc_set_free_prune(h.raw, int(prune)))
  try:
    h.check(h.lib.sc_stage_diffuse_rowstats(h.raw, _lib.as_double_p(a), n, mode,
                                            _lib.as_double_p(rmax), _lib.as_double_p(rsum), info))
  finally:
    if prune is not None:
      h.check(h.lib.sc_set_free_prune(h.raw, -1))
  return rmax, rsum, list(info)


def check_rowstats(a, name, expect_overflow=None, expect_formed=None):
  s = a @ a.T
  want_max, want_sum = s.max(axis=1), s.sum(axis=1)
  scale = np.abs(s).max()
  for mode in (EXPLICIT, FREE):
    rmax, rsum, info = rowstats(a, mode)
    # summation order only: 1e-13 of the row's own scale (|a_i| |a_j| bounds every term)
    tol = 1e-13 * np.maximum(np.abs(want_max), 1e-3 * scale)
    assert np.all(np.abs(rmax - want_max) <= tol), (name, mode, np.abs(rmax - want_max).max())
    tols = 1e-12 * np.maximum(np.abs(want_sum), np.abs(s).sum(axis=1))
    assert np.all(np.abs(rsum - want_sum) <= tols), (name, mode)
    if mode == FREE:
      if expect_overflow is not None:
        assert (info[1] > 0) == expect_overflow, (name, info)
      if expect_formed is not None:
        assert bool(info[3]) == expect_formed, (name, info)
  return info


def refined_before_diffuse(x, **over):
  cfg = so.icassp2018_config(**over)
  i = list(cfg.sequence).index(so.OP_DIFFUSE)
  return so.refine(so.affinity(x), dataclasses.replace(cfg, sequence=tuple(cfg.sequence[:i])))


# ------------------------------------------------------------------- stage level
@pytest.mark.parametrize("n,d,k", [(300, 32, 3), (1000, 64, 5), (1153, 48, 4), (2048, 128, 4)])
def test_rowstats_of_refined_affinities(n, d, k):
  a = refined_before_diffuse(so.blobs(n, d, k, seed=n))
  info = check_rowstats(a, "blobs%d" % n, expect_formed=False)
  assert info[0] <= 3 * n  # a few exact dot products per row, not n


@pytest.mark.parametrize("kind", so.HARD_KINDS)
def test_rowstats_of_unfriendly_inputs(kind):
  n = 1000
  g = golden("hard_%s_n%d_lap4.npz" % (kind, n))
  a = refined_before_diffuse(so.hard_inputs(kind, n, int(g["params"][1]), int(g["params"][2])))
  # "tiny": five samples whose rows of S are small against the slack -> evaluated in full
  check_rowstats(a, kind, expect_formed=False)


def test_rowstats_exact_and_near_ties():
  rng = np.random.default_rng(5)
  n = 640
  b = rng.random((n, n))
  a = 0.5 * (b + b.T)
  # exact ties: duplicated samples (identical rows AND columns)
  for dup in ((3, 77), (3, 200), (400, 401), (400, 402), (400, 403)):
    a[dup[1], :] = a[dup[0], :]
    a[:, dup[1]] = a[:, dup[0]]
  a = 0.5 * (a + a.T)
  # near ties: a row that differs from another in its last bits
  a[500, :] = a[10, :] * (1.0 + 3e-16)
  a[:, 500] = a[500, :]
  check_rowstats(a, "ties", expect_formed=False)


def test_rowstats_many_identical_rows_form_s_after_all():
  """More rows over the candidate cap than the exact-row route takes (64): S is formed
  explicitly, the result is still the reference's."""
  rng = np.random.default_rng(6)
  b = rng.random((32, 32))
  a = np.kron(0.5 * (b + b.T), np.ones((16, 16)))  # n = 512: 32 groups of 16 identical samples
  check_rowstats(a, "plateau", expect_overflow=True, expect_formed=True)


def test_rowstats_negative_entries_and_zero_row():
  rng = np.random.default_rng(7)
  n = 700
  b = rng.standard_normal((n, n))
  a = 0.5 * (b + b.T)
  a[123, :] = 0.0
  a[:, 123] = 0.0
  check_rowstats(a, "signed")


def test_rowstats_wide_dynamic_range():
  """Rows far below the matrix maximum: their digits carry few bits, the slack is large
  against their row of S and they are evaluated in full."""
  rng = np.random.default_rng(8)
  n = 900
  b = rng.random((n, n))
  a = 0.5 * (b + b.T)
  scale = np.ones(n)
  scale[[5, 6, 7, 450, 899]] = 1e-4
  a = a * scale[:, None] * scale[None, :]
  check_rowstats(a, "range", expect_formed=False)


# ------------------------------------------------------------------- tile skip list
def check_skip_list(a, name, expect_pruned):
  """The skip list must change NOTHING: rowmax / rowsum bit for bit, the candidate counts, the
  overflow record -- against the same search over every tile -- and rowmax against NumPy."""
  on = rowstats(a, FREE, prune=1)
  off = rowstats(a, FREE, prune=0)
  assert np.array_equal(on[0], off[0]) and np.array_equal(on[1], off[1]), name
  assert on[2][:4] == off[2][:4], (name, on[2], off[2])
  total = off[2][5]
  assert off[2][4] == total == on[2][5], (name, off[2])
  assert on[2][4] <= total
  if expect_pruned is True:
    assert on[2][4] < total, (name, on[2])
  elif expect_pruned is False:
    assert on[2][4] == total, (name, on[2])
  s = a @ a.T
  want = s.max(axis=1)
  tol = 1e-13 * np.maximum(np.abs(want), 1e-3 * np.abs(s).max())
  assert np.all(np.abs(on[0] - want) <= tol), name
  return on[2]


@pytest.mark.parametrize("n,d,k", [(1153, 48, 4), (2048, 128, 4), (3007, 64, 3), (4100, 64, 8)])
def test_skip_list_on_refined_affinities(n, d, k):
  """Blob-like input (clusters contiguous in the sample order, like speaker turns): most
  off-diagonal tiles hold nothing but entries far below their rows' diagonal -- skipped."""
  a = refined_before_diffuse(so.blobs(n, d, k, seed=n))
  info = check_skip_list(a, "blobs%d" % n, expect_pruned=True if n >= 2048 else None)
  if n >= 4096:
    assert info[4] < 0.5 * info[5], info


def test_skip_list_keeps_every_tile_of_unstructured_input():
  rng = np.random.default_rng(11)
  for n in (640, 1500):
    b = rng.random((n, n))
    check_skip_list(0.5 * (b + b.T), "uniform%d" % n, expect_pruned=False)
  # the same blobs in a random sample order: the block structure is gone from the tiles
  n = 2048
  x = so.blobs(n, 64, 4, seed=8)[rng.permutation(n)]
  check_skip_list(refined_before_diffuse(x), "shuffled", expect_pruned=None)


def test_skip_list_keeps_the_tile_of_a_distant_tie():
  """Duplicated samples 1500 rows apart: a row's product with its partner's row is (nearly) its
  own diagonal entry -- the partner's far-away tile holds a candidate and must stay: the candidate
  sets are the same with and without the list (check_skip_list), and larger than one per row."""
  n = 2304
  x = so.blobs(n, 64, 4, seed=12)
  for i in (5, 300, 700):
    x[i + 1500] = x[i]
  a = refined_before_diffuse(x)
  info = check_skip_list(a, "distant ties", expect_pruned=None)
  assert info[0] > n       # (candidates beyond the rows' own diagonal entries)


def test_skip_list_with_rows_tiny_against_the_rest_and_a_zero_row():
  """Rows whose diagonal-only threshold is negative (tiny rows: their slack exceeds their own
  T_ii) keep every tile of their tile row; a zero row keeps everything too."""
  n = 1792
  a = refined_before_diffuse(so.blobs(n, 64, 4, seed=13))
  scale = np.ones(n)
  scale[[7, 900, 1791]] = 1e-4
  a = a * scale[:, None] * scale[None, :]
  a[1000, :] = 0.0
  a[:, 1000] = 0.0
  check_skip_list(a, "tiny rows", expect_pruned=None)


# ------------------------------------------------------------------- end to end
def icassp_options():
  return sca.RefinementOptions(
      gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.RowMax,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


def rel_err(got, want):
  return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-12))


E2E = ["e2e_n200_lap4_max7.npz", "e2e_n1000_lap0_max7.npz", "e2e_n1000_lap4_max20.npz",
       "e2e_n1000_lap3_max20.npz", "e2e_n1000_lap2_max20.npz", "e2e_n2048_lap0_max7.npz",
       "e2e_n2048_lap4_max20.npz", "e2e_n8192_lap4_max20.npz", "e2e_n8192_lap0_max7.npz"]


@pytest.mark.parametrize("mode", [FREE, EXPLICIT])
@pytest.mark.parametrize("name", E2E)
def test_predict_vs_reference_golden_both_routes(name, mode):
  g = golden(name)
  n, d, k, seed, lap, max_clusters = [int(v) for v in g["params"]]
  x = so.blobs(n, d, k, seed)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=max_clusters,
                            refinement_options=icassp_options(), laplacian_type=LAP[lap])
  c.diffuse_mode = mode
  labels = c.predict(x)
  dg = c.last_diag
  assert dg.diffuse_path == (_lib.DIFFUSE_PATH_FREE if mode == FREE
                             else _lib.DIFFUSE_PATH_EXPLICIT), name
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-6)
  w = dg.eigenvalue_array()
  idx, ref = g["consumed_index"], g["consumed_eigenvalues"]
  if lap in (0, 1):
    keep = so.consumed_eigen_indices(n, max_clusters, True, ref, 1e-2)
    idx, ref = idx[keep], ref[keep]
  assert rel_err(w[idx], ref) < 1e-6, name
  if mode == FREE:
    assert dg.free_candidates >= n and dg.free_candidates <= 3 * n


@pytest.mark.parametrize("n", [1050, 2100, 2239, 3007])
@pytest.mark.parametrize("lap", [0, 4])
def test_routes_agree_at_ragged_sizes(n, lap):
  """Sizes whose 64-row tile count is odd or that end inside a tile: the digits written by the
  threshold + symmetrise pass (edge tiles, the 64 rows between the last tile and the product's
  128-row padding) against the explicit product on the same input."""
  x = so.blobs(n, 48, 5, seed=n + lap)
  out = {}
  for mode in (FREE, EXPLICIT):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=12, refinement_options=icassp_options(),
                              laplacian_type=LAP[lap])
    c.diffuse_mode = mode
    labels = c.predict(x)
    dg = c.last_diag
    assert dg.diffuse_path == (_lib.DIFFUSE_PATH_FREE if mode == FREE else _lib.DIFFUSE_PATH_EXPLICIT)
    out[mode] = (labels, dg.n_clusters_raw, dg.max_delta, c.consumed_eigenvalues())
  assert out[FREE][1] == out[EXPLICIT][1]
  np.testing.assert_allclose(out[FREE][2], out[EXPLICIT][2], rtol=1e-6)
  wf, wx = out[FREE][3], out[EXPLICIT][3]
  if lap == 0:  # the descending loop stops reading after the first value < 1e-2
    below = np.nonzero(wx < 1e-2)[0]
    stop = int(below[0]) + 1 if below.size else wx.size
    wf, wx = wf[:stop], wx[:stop]
  scale = np.abs(wx).max()
  assert np.max(np.abs(wf - wx)) < 2e-6 * scale
  assert so.adjusted_rand_index(out[FREE][0], out[EXPLICIT][0]) == 1.0


@pytest.mark.parametrize("lap", [0, 4])
def test_matrix_free_route_vs_oracle_at_a_ragged_size(lap):
  """the matrix-free route (skip list on) at a size that ends inside a tile,
  against the ORACLE (np.linalg.eig on the reference's own matrix), not against the other route:
  labels, cluster count, maximum gap, every consumed eigenvalue; and the skip list off gives the
  same eigenvalues bit for bit."""
  n, maxc = 2239, 12
  x = so.blobs(n, 48, 5, seed=n + lap)
  dump = {}
  want = so.predict(x, so.icassp2018_config(laplacian_type=lap, max_clusters=maxc), dump)
  got = {}
  for prune in (1, 0):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc,
                              refinement_options=icassp_options(), laplacian_type=LAP[lap])
    c.diffuse_mode = FREE
    h = c._handle()
    h.check(h.lib.sc_set_free_prune(h.raw, prune))
    try:
      labels = c.predict(x)
    finally:
      h.check(h.lib.sc_set_free_prune(h.raw, -1))
    dg = c.last_diag
    assert dg.diffuse_path == _lib.DIFFUSE_PATH_FREE
    total = -(-n // 128) * (-(-n // 128) + 1) // 2
    assert (dg.free_tiles_run < total) if prune else (dg.free_tiles_run == total)
    assert dg.n_clusters == dump["n_clusters"]
    np.testing.assert_allclose(dg.max_delta, dump["max_delta"], rtol=1e-6)
    assert so.adjusted_rand_index(labels, want) == 1.0
    w = c.consumed_eigenvalues()
    idx = so.consumed_eigen_indices(n, maxc, lap == 0, dump["eigenvalues"] if lap == 0 else None,
                                    1e-2 if lap == 0 else None)
    assert rel_err(w[idx], dump["eigenvalues"][idx]) < 1e-6
    got[prune] = w
  assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("binarize,preserve,sym", [(True, False, "Max"), (False, True, "Average"),
                                                   (True, True, "Average")])
def test_routes_agree_with_binarisation_and_preserved_diagonal(binarize, preserve, sym):
  """The digit-writing threshold pass with the options that put exact ones into the matrix
  (max|a| is then floored at 1) and with the Average symmetrisation: both routes, same input."""
  n = 1390
  x = so.blobs(n, 64, 4, seed=31)
  opts = sca.RefinementOptions(
      gaussian_blur_sigma=1, p_percentile=0.9, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.RowMax, thresholding_with_binarization=binarize,
      thresholding_preserve_diagonal=preserve, symmetrize_type=getattr(sca.SymmetrizeType, sym),
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  out = {}
  for mode in (FREE, EXPLICIT):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=10, refinement_options=opts,
                              laplacian_type=LAP[4])
    c.diffuse_mode = mode
    labels = c.predict(x)
    dg = c.last_diag
    assert dg.diffuse_path in ((_lib.DIFFUSE_PATH_FREE, _lib.DIFFUSE_PATH_FREE_THEN_EXPLICIT)
                               if mode == FREE else (_lib.DIFFUSE_PATH_EXPLICIT,))
    out[mode] = (labels, dg.n_clusters_raw, dg.max_delta, c.consumed_eigenvalues())
  assert out[FREE][1] == out[EXPLICIT][1]
  np.testing.assert_allclose(out[FREE][2], out[EXPLICIT][2], rtol=1e-6)
  wf, wx = out[FREE][3], out[EXPLICIT][3]
  assert np.max(np.abs(wf - wx)) < 2e-6 * np.abs(wx).max()
  assert so.adjusted_rand_index(out[FREE][0], out[EXPLICIT][0]) == 1.0


def test_stage_eig_after_a_matrix_free_predict():
  """The matrix-free operator is a property of ONE solve: a stage call on the same handle right
  after such a predict() solves the matrix it is given (the flag used to outlive the call)."""
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=icassp_options(),
                            laplacian_type=LAP[4])
  c.diffuse_mode = FREE
  c.predict(so.blobs(2048, 64, 4, seed=5))
  assert c.last_diag.diffuse_path == _lib.DIFFUSE_PATH_FREE
  rng = np.random.default_rng(9)
  n, count = 600, 6
  q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  lam = np.concatenate([np.linspace(3.0, 2.0, count), rng.uniform(-0.5, 0.5, n - count)])
  m = (q * lam) @ q.T
  m = 0.5 * (m + m.T)
  h = _lib.default_handle()
  values = np.empty(count)
  vectors = np.empty((n, count))
  h.check(h.lib.sc_stage_sym_eig(h.raw, _lib.as_double_p(np.ascontiguousarray(m)), n, count, 1,
                                 _lib.as_double_p(values), _lib.as_double_p(vectors), _lib.ScDiag()))
  want = np.sort(np.linalg.eigvalsh(m))[::-1][:count]
  np.testing.assert_allclose(values, want, rtol=1e-9)


def test_default_routing_by_size():
  opts = icassp_options()
  for n, want in ((1000, _lib.DIFFUSE_PATH_EXPLICIT), (2048, _lib.DIFFUSE_PATH_FREE)):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
    c.predict(so.blobs(n, 64, 4, seed=n))
    assert c.last_diag.diffuse_path == want, n


@pytest.mark.parametrize("kind", so.HARD_KINDS)
@pytest.mark.parametrize("n,lap", [(1000, 4), (2048, 0), (2048, 4), (4096, 4)])
def test_hard_inputs_matrix_free(kind, n, lap):
  """The unfriendly spectra (restarts, long bases, the "tiny" cluster whose rows overflow
  the candidate lists) through the two-pass operator."""
  name = "hard_%s_n%d_lap%d" % (kind, n, lap)
  g = golden(name + ".npz")
  nn, d, seed, lap_g, maxc = (int(v) for v in g["params"])
  x = so.hard_inputs(kind, n, d, seed)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, laplacian_type=LAP[lap],
                            refinement_options=icassp_options())
  c.diffuse_mode = FREE
  labels = c.predict(x)
  dg = c.last_diag
  assert dg.diffuse_path in (_lib.DIFFUSE_PATH_FREE, _lib.DIFFUSE_PATH_FREE_THEN_EXPLICIT)
  idx, ref = g["consumed_index"], g["consumed_eigenvalues"]
  scale = np.abs(g["head_eigenvalues"]).max()
  err = np.abs(dg.eigenvalue_array()[idx] - ref) / np.maximum(np.abs(ref), 1e-9 * scale)
  assert err.max() < 1e-5, (name, err.max())
  assert dg.n_clusters_raw == int(g["n_clusters_raw"]), name
  np.testing.assert_allclose(dg.max_delta, float(g["max_delta"]), rtol=1e-5, err_msg=name)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0, name
  if kind == "tiny":
    assert dg.free_overflow_rows > 0  # its five far-away samples were evaluated in full


def test_other_sequences_keep_the_explicit_product():
  """Diffuse in the middle of a sequence (its entries ARE read) stays explicit even when the
  matrix-free route is asked for."""
  x = so.blobs(600, 32, 3, seed=600)
  seq = [sca.RefinementName.CropDiagonal, sca.RefinementName.RowWiseThreshold,
         sca.RefinementName.Symmetrize, sca.RefinementName.Diffuse,
         sca.RefinementName.GaussianBlur, sca.RefinementName.RowWiseNormalize]
  opts = sca.RefinementOptions(gaussian_blur_sigma=1, p_percentile=0.9,
                               thresholding_soft_multiplier=0.01, refinement_sequence=seq)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
  c.diffuse_mode = FREE
  got = c.predict(x)
  assert c.last_diag.diffuse_path == _lib.DIFFUSE_PATH_EXPLICIT
  ocfg = so.OracleConfig(sequence=(so.OP_CROP_DIAGONAL, so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE,
                                   so.OP_DIFFUSE, so.OP_GAUSSIAN_BLUR, so.OP_ROW_WISE_NORMALIZE),
                         p_percentile=0.9, min_clusters=2, max_clusters=7)
  assert so.adjusted_rand_index(got, so.predict(x, ocfg)) == 1.0
