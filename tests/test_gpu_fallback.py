"""GPU parity tests for the callers around the spectral path:
naive clusterer, fallback clusterer, single-cluster conditions, multi-stage streaming --
against the reference's own known answers (tests/naive_clusterer_test.py,
tests/fallback_clusterer_test.py, tests/spectral_clusterer_test.py:330-495,
tests/multi_stage_clusterer_test.py) and against outputs of the real reference
(tests/golden/fallback.npz, oracle/make_golden.py --fallback).
"""

import numpy as np
import pytest

import spectral_oracle as so
from conftest import golden

import spectralcluster_amd as sca
from spectralcluster_amd import fallback_clusterer as fb
from spectralcluster_amd import multi_stage_clusterer as ms
from spectralcluster_amd import naive_clusterer

pytestmark = pytest.mark.gpu

TOY = np.array([[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.0], [0.9, -0.1], [0.0, 1.2]])
ONE = np.array([[1.0, 0.0], [1.1, 0.1], [1.0, 0.0], [1.1, 0.0], [0.9, -0.1], [1.0, 0.2]])


def icassp_options(sigma=0, p=0.95):
  return sca.RefinementOptions(gaussian_blur_sigma=sigma, p_percentile=p,
                               refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


# --- naive clusterer (reference tests/naive_clusterer_test.py) ------------------------
def test_naive_6by2_and_online_state():
  clusterer = naive_clusterer.NaiveClusterer(threshold=0.5)
  labels = sca.utils.enforce_ordered_labels(clusterer.predict(TOY))
  np.testing.assert_equal(labels, [0, 0, 1, 1, 0, 1])
  assert clusterer.predict_next(np.array([1.2, -0.1])) == 0
  assert clusterer.predict_next(np.array([-0.1, 0.8])) == 1
  clusterer.reset()
  assert clusterer.predict_next(np.array([-0.1, 0.8])) == 0


def test_naive_adaptation():
  clusterer = naive_clusterer.NaiveClusterer(threshold=0.5, adaptation_threshold=1.0)
  assert clusterer.predict_next(np.array([1.2, -0.1])) == 0
  assert clusterer.centroids[0].count == 1
  assert clusterer.predict_next(np.array([1.3, 0.2])) == 0      # too strict: no merge
  assert clusterer.centroids[0].count == 1
  clusterer.adaptation_threshold = 0.5
  assert clusterer.predict_next(np.array([1.3, 0.2])) == 0
  assert clusterer.centroids[0].count == 2
  np.testing.assert_array_equal(clusterer.centroids[0].embedding,
                                (np.array([1.2, -0.1]) * 1 + np.array([1.3, 0.2])) / 2)
  with pytest.raises(ValueError):
    naive_clusterer.NaiveClusterer(threshold=0.5, adaptation_threshold=0.4)


@pytest.mark.parametrize("tag,thr,ad", [("t5", 0.5, None), ("t7a9", 0.7, 0.9)])
def test_naive_vs_reference_golden(tag, thr, ad):
  g = golden("fallback.npz")
  x = so.blobs(300, 16, 4, 101, noise=0.6)
  clusterer = naive_clusterer.NaiveClusterer(thr, ad)
  first = clusterer.predict(x[:100])       # state carries over between calls
  rest = clusterer.predict(x[100:])
  np.testing.assert_array_equal(np.concatenate([first, rest]), g["naive_" + tag])
  np.testing.assert_array_equal([c.count for c in clusterer.centroids],
                                g["naive_counts_" + tag])
  np.testing.assert_allclose(np.stack([c.embedding for c in clusterer.centroids]),
                             g["naive_cent_" + tag], rtol=1e-13)


# --- fallback clusterer (reference tests/fallback_clusterer_test.py) --------------------
def test_fallback_clusterer_6by2():
  for kind in (fb.FallbackClustererType.Naive, fb.FallbackClustererType.Agglomerative):
    options = fb.FallbackOptions(fallback_clusterer_type=kind, naive_threshold=0.5,
                                 agglomerative_threshold=0.5)
    labels = fb.FallbackClusterer(options).predict(TOY)
    np.testing.assert_equal(sca.utils.enforce_ordered_labels(labels), [0, 0, 1, 1, 0, 1])


def test_gmm_bic_reference_known_answers():
  options = fb.FallbackOptions(single_cluster_condition=fb.SingleClusterCondition.AffinityGmmBic)
  a = np.array([[1, 0.999, 1.001], [0.999, 1, 1], [1.001, 1, 1]])
  assert fb.check_single_cluster(options, None, a)
  a = np.array([[1.0, 2, 2], [2, 1, 1], [2, 1, 1]])
  assert not fb.check_single_cluster(options, None, a)
  with pytest.raises(ValueError, match="diagonal_offset"):
    fb.check_single_cluster(
        fb.FallbackOptions(single_cluster_affinity_diagonal_offset=2), None, a)


def test_gmm_bic_vs_sklearn():
  """Same BICs as sklearn's GaussianMixture where its seeded k-means start is unambiguous."""
  from sklearn.mixture import GaussianMixture
  from spectralcluster_amd import _lib
  rng = np.random.default_rng(5)
  for n, spread in ((40, 0.0), (120, 0.4)):
    x = so.blobs(n, 16, 2, seed=n, noise=0.2 + spread)
    a = so.affinity(x)
    vals = a[np.triu_indices(n, 1)][:, None]
    want1 = GaussianMixture(n_components=1).fit(vals).bic(vals)
    want2 = GaussianMixture(n_components=2, random_state=0).fit(vals).bic(vals)
    handle = _lib.default_handle()
    handle.check(handle.lib.sc_set_affinity(handle.raw, _lib.as_double_p(a), n))
    bic = np.empty(2)
    handle.check(handle.lib.sc_affinity_gmm_bic(handle.raw, 1, _lib.as_double_p(bic[0:1]),
                                                _lib.as_double_p(bic[1:2])))
    np.testing.assert_allclose(bic[0], want1, rtol=1e-9)
    np.testing.assert_allclose(bic[1], want2, rtol=1e-3)   # EM stops at tol=1e-3 per sample
  del rng


@pytest.mark.parametrize("name,k,seed", [("one", 1, 102), ("many", 3, 103)])
def test_single_cluster_conditions_vs_reference_golden(name, k, seed):
  g = golden("fallback.npz")
  x = so.blobs(80, 16, k, seed, noise=0.2)
  a = so.affinity(x)
  from spectralcluster_amd import _lib
  handle = _lib.default_handle()
  handle.check(handle.lib.sc_set_affinity(handle.raw, _lib.as_double_p(a), 80))
  stats = np.empty(4)
  handle.check(handle.lib.sc_affinity_stats(handle.raw, _lib.as_double_p(stats)))
  assert stats[0] == g["stats_" + name][0] and stats[1] == g["stats_" + name][1]
  np.testing.assert_allclose(stats[2:], g["stats_" + name][2:], rtol=1e-12)
  for cond in ("AllAffinity", "NeighborAffinity", "AffinityStd", "FallbackClusterer"):
    for thr in (0.5, 0.75, 0.9):
      options = fb.FallbackOptions(
          single_cluster_condition=getattr(fb.SingleClusterCondition, cond),
          single_cluster_affinity_threshold=thr,
          fallback_clusterer_type=fb.FallbackClustererType.Agglomerative)
      want = bool(g["single_%s_%s_%02d" % (name, cond, round(thr * 100))])
      assert fb.check_single_cluster(options, x, a) == want, (cond, thr)
  assert fb.check_single_cluster(fb.FallbackOptions(), x, a) == bool(g["single_%s_gmm" % name])
  with pytest.raises(TypeError):
    fb.check_single_cluster(fb.FallbackOptions(single_cluster_condition="nope"), x, a)


# --- predict() with min_clusters=1 (reference tests/spectral_clusterer_test.py:330-495) ---
def test_predict_single_cluster_known_answers():
  clusterer = sca.SpectralClusterer(min_clusters=1, refinement_options=icassp_options())
  np.testing.assert_equal(sca.utils.enforce_ordered_labels(clusterer.predict(ONE)), [0] * 6)
  m = ONE.copy()
  m[5] = [1.0, 0.5]
  for cond, hi, lo in ((fb.SingleClusterCondition.AllAffinity, 0.93, 0.91),):
    for thr, want in ((hi, [0, 0, 0, 0, 0, 1]), (lo, [0] * 6)):
      c = sca.SpectralClusterer(
          min_clusters=1, laplacian_type=sca.LaplacianType.GraphCut, refinement_options=None,
          fallback_options=fb.FallbackOptions(single_cluster_condition=cond,
                                              single_cluster_affinity_threshold=thr))
      np.testing.assert_equal(sca.utils.enforce_ordered_labels(c.predict(m)), want)
  # fallback clusterer as the condition (naive, threshold 0.5): one cluster
  c = sca.SpectralClusterer(
      min_clusters=1, laplacian_type=sca.LaplacianType.GraphCut,
      fallback_options=fb.FallbackOptions(
          single_cluster_condition=fb.SingleClusterCondition.FallbackClusterer,
          fallback_clusterer_type=fb.FallbackClustererType.Naive, naive_threshold=0.5))
  np.testing.assert_equal(c.predict(m), [0] * 6)


def test_predict_min1_and_few_embeddings_vs_reference_golden():
  g = golden("fallback.npz")
  for name, k, seed in (("one", 1, 102), ("many", 3, 103)):
    x = so.blobs(80, 16, k, seed, noise=0.2)
    c = sca.SpectralClusterer(
        min_clusters=1, max_clusters=6,
        refinement_options=icassp_options(sigma=1),
        fallback_options=fb.FallbackOptions(
            single_cluster_condition=fb.SingleClusterCondition.AffinityStd,
            single_cluster_affinity_threshold=0.05))
    got = c.predict(x)
    assert so.adjusted_rand_index(got, g["predict_min1_" + name]) == 1.0
  x = so.blobs(80, 16, 3, 103, noise=0.2)
  c = sca.SpectralClusterer(
      refinement_options=icassp_options(sigma=1),
      fallback_options=fb.FallbackOptions(
          spectral_min_embeddings=100,
          fallback_clusterer_type=fb.FallbackClustererType.Agglomerative,
          agglomerative_threshold=0.4))
  np.testing.assert_array_equal(c.predict(x), g["predict_few"])


# --- multi-stage streaming (reference tests/multi_stage_clusterer_test.py) --------------
def make_multi_stage(deflicker=ms.Deflicker.NoDeflicker):
  main = sca.SpectralClusterer(refinement_options=icassp_options())
  return ms.MultiStageClusterer(main_clusterer=main, fallback_threshold=0.5, L=3, U1=5, U2=7,
                                deflicker=deflicker)


STREAM = [[1, 2], [3, -1], [1, 1], [-2, -1], [0, 1], [-2, 0]]


@pytest.mark.parametrize("count,expected", [
    (1, [0]), (2, [0, 1]), (4, [0, 0, 0, 1]), (6, [0, 1, 0, 2, 3, 2]),
    (8, [0, 1, 0, 2, 3, 2, 0, 1]), (10, [0, 1, 0, 2, 3, 2, 0, 1, 0, 2]),
    (16, [0, 1, 0, 2, 3, 2, 0, 1, 0, 2, 3, 2, 0, 1, 0, 2])])
def test_multi_stage_reference_known_answers(count, expected):
  stream = (STREAM * 3)[:count]
  for deflicker in ((ms.Deflicker.NoDeflicker, ms.Deflicker.OrderBased, ms.Deflicker.Hungarian)
                    if count == 8 else (ms.Deflicker.NoDeflicker,)):
    clusterer = make_multi_stage(deflicker)
    for e in stream:
      row = np.array([e]) if count == 1 else np.array(e)
      labels = clusterer.streaming_predict(row)
    np.testing.assert_equal(sca.utils.enforce_ordered_labels(labels), expected)
  with pytest.raises(ValueError):
    ms.MultiStageClusterer(main_clusterer=sca.SpectralClusterer(max_spectral_size=50))


@pytest.mark.parametrize("tag,deflicker", [("none", ms.Deflicker.NoDeflicker),
                                           ("hungarian", ms.Deflicker.Hungarian)])
def test_multi_stage_stream_vs_reference_golden(tag, deflicker):
  g = golden("fallback.npz")
  main = sca.SpectralClusterer(refinement_options=icassp_options(p=0.2), stop_eigenvalue=0.01)
  clusterer = ms.MultiStageClusterer(main_clusterer=main, fallback_threshold=0.5, L=20, U1=60,
                                     U2=120, deflicker=deflicker)
  for step, e in enumerate(g["stream"], 1):
    labels = clusterer.streaming_predict(e)
    key = "ms_%s_%d" % (tag, step)
    if key in g:
      want = g[key]
      if deflicker == ms.Deflicker.Hungarian and step > 60:
        np.testing.assert_array_equal(labels, want)      # the matching pins the numbering
        assert labels.dtype == want.dtype
      else:
        assert so.adjusted_rand_index(np.asarray(labels).astype(int), want.astype(int)) == 1.0


def test_multi_stage_1000by6():
  # reference tests/multi_stage_clusterer_test.py:221-245 (seeded noise)
  rng = np.random.default_rng(3)
  base = np.array([[1.0, 0, 0, 0, 0, 0]] * 100 + [[0, 1.0, 0, 0, 0, 0]] * 200 +
                  [[0, 0, 2.0, 0, 0, 0]] * 300 + [[0, 0, 0, 1.0, 0, 0]] * 400)
  matrix = base + (rng.random((1000, 6)) * 2 - 1) * 0.02
  main = sca.SpectralClusterer(refinement_options=icassp_options(p=0.2), stop_eigenvalue=0.01)
  clusterer = ms.MultiStageClusterer(main_clusterer=main, fallback_threshold=0.5, L=50, U1=200,
                                     U2=400)
  for e in matrix:
    labels = clusterer.streaming_predict(e)
  np.testing.assert_equal(sca.utils.enforce_ordered_labels(labels),
                          [0] * 100 + [1] * 200 + [2] * 300 + [3] * 400)
