"""Grouped execution of a batch (sc_predict_batch_grouped: up to 16 utterances per launch,
eigensolver and k-means chains in lockstep, groups dealt to three lanes) gives every utterance the
result of its own predict() call -- against single calls, against the oracle and against the
real-reference golden of BASELINE config 5."""
import os

import numpy as np
import pytest

import spectralcluster_amd as sca
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def icassp(**kw):
  return sca.SpectralClusterer(
      min_clusters=kw.pop("min_clusters", 2), max_clusters=kw.pop("max_clusters", 7),
      refinement_options=sca.configs.icassp2018_refinement_options, **kw)


def mixed_utterances(count, seed, lo=130, hi=2600, d=48):
  rng = np.random.default_rng(seed)
  ns = rng.integers(lo, hi, count)
  ks = rng.integers(2, 7, count)
  return [so.blobs(int(n), d, int(k), seed=1000 * seed + i) for i, (n, k) in enumerate(zip(ns, ks))]


@pytest.mark.parametrize("group", [2, 5, 16])
@pytest.mark.parametrize("lap", [None, sca.LaplacianType.GraphCut])
def test_grouped_labels_equal_single_calls(group, lap):
  utts = mixed_utterances(37, seed=3 + group)
  c = icassp(laplacian_type=lap, max_clusters=7 if lap is None else 12)
  got = c.predict_batch(utts, group=group)
  diags = c.last_batch_diags
  for i, u in enumerate(utts):
    want = c.predict(u)
    assert np.array_equal(got[i], want), (i, u.shape)
    assert diags[i].n_clusters == c.last_diag.n_clusters
    assert diags[i].n_clusters_raw == c.last_diag.n_clusters_raw
    # the eigenvalues the eigengap rule reads (the solver holds only those to its tolerance)
    w, w1 = diags[i].eigenvalue_array(), c.last_diag.eigenvalue_array()
    idx = so.consumed_eigen_indices(u.shape[0], c.max_clusters, lap is None, w1, 1e-2)
    np.testing.assert_allclose(w[idx], w1[idx], rtol=2e-6)


@pytest.mark.parametrize("lap", [sca.LaplacianType.Unnormalized, sca.LaplacianType.RandomWalk])
def test_grouped_front_with_the_other_laplacians(lap):
  """the grouped scaling-vector launch carries the Laplacian type"""
  utts = mixed_utterances(18, seed=61, lo=520, hi=1500)
  c = icassp(laplacian_type=lap, max_clusters=10)
  got = c.predict_batch(utts, group=16)
  for u, lab in zip(utts, got):
    assert np.array_equal(lab, c.predict(u))


def test_other_refinement_sequences_take_the_member_front():
  """The grouped front covers the ICASSP2018 sequence with its fusions; any other sequence
  (here: percentile threshold without blur, and the full sequence with a diagonal-preserving
  threshold) runs the single-call stages member by member, then the lockstep chains."""
  utts = mixed_utterances(14, seed=71, lo=520, hi=1400, d=32)
  seqs = [
      sca.RefinementOptions(
          p_percentile=0.9, thresholding_soft_multiplier=0.01,
          thresholding_type=sca.ThresholdType.Percentile,
          refinement_sequence=[sca.RefinementName.RowWiseThreshold, sca.RefinementName.Symmetrize,
                               sca.RefinementName.Diffuse, sca.RefinementName.RowWiseNormalize]),
      sca.RefinementOptions(
          gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
          thresholding_preserve_diagonal=True,
          refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
  ]
  for opts in seqs:
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
    got = c.predict_batch(utts, group=8)
    for u, lab in zip(utts, got):
      assert np.array_equal(lab, c.predict(u))


def test_blur_radius_eight_in_the_grouped_front():
  opts = sca.RefinementOptions(gaussian_blur_sigma=2, p_percentile=0.95,
                               thresholding_soft_multiplier=0.01,
                               refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=7, refinement_options=opts)
  utts = mixed_utterances(10, seed=81, lo=600, hi=2300, d=32)
  got = c.predict_batch(utts, group=16)
  for u, lab in zip(utts, got):
    assert np.array_equal(lab, c.predict(u))


def test_grouped_batch_vs_oracle():
  utts = mixed_utterances(12, seed=11, lo=140, hi=900, d=24)
  c = icassp()
  got = c.predict_batch(utts, group=16)
  for u, lab in zip(utts, got):
    want = so.predict(u, so.icassp2018_config())
    assert so.adjusted_rand_index(lab, want) == 1.0


def test_members_outside_the_grouped_range_take_the_single_path():
  """n <= 128 (dense Jacobi) and n >= 4096 (symmetric-storage matvec) are not grouped; a batch
  may mix them freely, and a batch of one works."""
  sizes = [60, 128, 129, 4100, 300, 17, 2047]
  utts = [so.blobs(n, 32, 3, seed=n) for n in sizes]
  c = icassp()
  got = c.predict_batch(utts, group=8)
  for u, lab in zip(utts, got):
    assert np.array_equal(lab, c.predict(u))
  assert np.array_equal(c.predict_batch(utts[4:5], group=8)[0], c.predict(utts[4]))
  assert c.predict_batch([], group=8) == []


def test_member_that_leaves_the_common_path_is_recomputed():
  """A spectrum that needs more than 64 basis vectors (many near-equal leading eigenvalues)
  cannot finish inside the lockstep solve: that member goes through the single-call solver
  (restarts and all), the others stay grouped."""
  hard = so.blobs(400, 200, 3, seed=1)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=40,
                            refinement_options=sca.configs.icassp2018_refinement_options)
  utts = [so.blobs(500, 200, 4, seed=2), hard, so.blobs(350, 200, 2, seed=3)]
  got = c.predict_batch(utts, group=4)
  for u, lab in zip(utts, got):
    assert np.array_equal(lab, c.predict(u))


def test_row_wise_renorm_and_min_clusters_in_a_group():
  utts = mixed_utterances(9, seed=21, lo=150, hi=700, d=16)
  c = icassp(row_wise_renorm=True, min_clusters=4)
  got = c.predict_batch(utts, group=4)
  for u, lab in zip(utts, got):
    assert np.array_equal(lab, c.predict(u))
  assert all(d.n_clusters >= 4 for d in c.last_batch_diags)


def test_non_finite_member_raises_like_predict():
  utts = mixed_utterances(5, seed=31, lo=150, hi=400, d=16)
  utts[2] = utts[2].copy()
  utts[2][7] = 0.0  # a zero embedding row: NaN cosine, np.linalg.eig raises in the reference
  c = icassp()
  with pytest.raises(Exception) as single:
    c.predict(utts[2])
  with pytest.raises(type(single.value)):
    c.predict_batch(utts, group=4)
  # the handle (and its member arenas) stay usable
  ok = c.predict_batch(utts[:2], group=4)
  assert np.array_equal(ok[0], c.predict(utts[0]))


def test_grouped_is_deterministic_and_order_independent():
  utts = mixed_utterances(20, seed=41, lo=130, hi=800, d=24)
  c = icassp()
  a = c.predict_batch(utts, group=16)
  b = c.predict_batch(utts, group=16)
  rev = c.predict_batch(utts[::-1], group=3)[::-1]
  for x, y, z in zip(a, b, rev):
    assert np.array_equal(x, y) and np.array_equal(x, z)


def test_lanes_give_every_group_the_one_lane_result():
  """A batch of >= 4 groups is dealt to three lanes (three leads, three host threads): every
  utterance still gets the result of its own predict() call, two passes agree bit for bit
  (labels and reported eigenvalues), and so does a pass in a process that created the
  multi-stream form's eight streams first (the lanes pick their streams by measurement)."""
  utts = mixed_utterances(100, seed=77, lo=260, hi=1400, d=32)  # 7 groups of 16
  c = icassp(laplacian_type=sca.LaplacianType.GraphCut, max_clusters=10)
  a = c.predict_batch(utts, group=16)
  da = [d.eigenvalue_array().copy() for d in c.last_batch_diags]
  c.predict_batch(utts[:24], streams=8)
  b = c.predict_batch(utts, group=16)
  db = [d.eigenvalue_array().copy() for d in c.last_batch_diags]
  for i, u in enumerate(utts):
    assert np.array_equal(a[i], b[i]), i
    assert np.array_equal(da[i], db[i]), i
  for i in range(0, 100, 7):
    assert np.array_equal(a[i], c.predict(utts[i])), i
  # a fresh clusterer (new lead, new lanes) reproduces it
  c2 = icassp(laplacian_type=sca.LaplacianType.GraphCut, max_clusters=10)
  for x, y in zip(a, c2.predict_batch(utts, group=16)):
    assert np.array_equal(x, y)


def test_config5_batch512_grouped_vs_reference():
  g = np.load(os.path.join(GOLDEN, "batch512.npz"))
  ns, ks = g["ns"], g["ks"]
  utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
  clusterer = sca.configs.icassp2018_clusterer
  labels = clusterer.predict_batch(utts, group=16)
  diags = clusterer.last_batch_diags
  ref_labels, pos, bad = g["labels"], 0, []
  worst = 0.0
  for i, n in enumerate(ns):
    n = int(n)
    ref = ref_labels[pos:pos + n]
    pos += n
    d = diags[i]
    w_ref = g["consumed_eigenvalues"][i]
    idx = so.consumed_eigen_indices(n, 7, True, w_ref, 1e-2)
    w = d.eigenvalue_array()[idx]
    worst = max(worst, float(np.max(np.abs(w - w_ref[idx]) / np.maximum(np.abs(w_ref[idx]), 1e-12))))
    if (so.adjusted_rand_index(labels[i], ref) != 1.0 or
        d.n_clusters_raw != g["n_clusters_raw"][i] or
        abs(d.max_delta - g["max_delta"][i]) > 1e-6 * abs(g["max_delta"][i])):
      bad.append(i)
  assert not bad, bad
  assert worst < 1e-5, worst


def test_matrix_free_member_handed_back_resumes_on_its_two_pass_operator():
  """A large member of a batch group (n >= 1536) takes the matrix-free Diffuse: its front leaves
  A, not S = A A^T.  When such a member leaves the lockstep path AFTER its front (here: 40
  clusters wanted, beyond the 32 the k-means chain holds) the single-call solver resumes from
  that front and must apply A twice -- it once solved diag(p) + diag(c) A diag(c) with the
  scaling vectors of S, silently.  Labels, eigenvalues and eigengap must be those
  of the member's own predict() call."""
  from spectralcluster_amd import _lib
  utts = [so.blobs(n, 48, 5, seed=n) for n in (1700, 600, 2250, 1990)]
  c = icassp(min_clusters=40, max_clusters=12, laplacian_type=sca.LaplacianType.GraphCut)
  got = c.predict_batch(utts, group=8)
  diags = c.last_batch_diags
  for i, u in enumerate(utts):
    want = c.predict(u)
    one = c.last_diag
    assert diags[i].n_clusters == 40 and one.n_clusters == 40
    assert diags[i].n_clusters_raw == one.n_clusters_raw, i
    if u.shape[0] >= 1536:
      assert diags[i].diffuse_path in (_lib.DIFFUSE_PATH_FREE,
                                       _lib.DIFFUSE_PATH_FREE_THEN_EXPLICIT), i
    w, w1 = diags[i].eigenvalue_array(), one.eigenvalue_array()
    idx = so.consumed_eigen_indices(u.shape[0], 12, False, w1, 1e-2)
    np.testing.assert_allclose(w[idx], w1[idx], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(diags[i].max_delta, one.max_delta, rtol=1e-6)
    assert so.adjusted_rand_index(got[i], want) == 1.0, i
