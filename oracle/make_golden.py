"""Generate tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--large | --float32 | --constraints | --general | --size-reduction | --fallback | --autotune4096 | --autotune4096-ttd | --batch512 | --dense | --hard]

`--autotune4096` and `--batch512` pin BASELINE.json configs 4 and 5 at their
full sizes (about 6 and 20 minutes of CPU).

`--large` additionally runs the two n=8192 configurations (about 150-160 s of
CPU each).  Inputs are regenerated from seeds by `spectral_oracle.blobs`; only
small outputs are stored (consumed eigenvalues, cluster counts, labels), plus
full per-stage matrices for the tiny cases.
"""

import copy
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

import spectral_oracle as so  # noqa: E402
from spectralcluster import autotune as ref_autotune  # noqa: E402
from spectralcluster import configs as ref_configs  # noqa: E402
from spectralcluster import constraint as ref_constraint  # noqa: E402
from spectralcluster import custom_distance_kmeans as ref_kmeans  # noqa: E402
from spectralcluster import laplacian as ref_laplacian  # noqa: E402
from spectralcluster import refinement as ref_refinement  # noqa: E402
from spectralcluster import spectral_clusterer as ref_sc  # noqa: E402
from spectralcluster import utils as ref_utils  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

LAP = {0: None, 1: ref_laplacian.LaplacianType.Affinity,
       2: ref_laplacian.LaplacianType.Unnormalized,
       3: ref_laplacian.LaplacianType.RandomWalk,
       4: ref_laplacian.LaplacianType.GraphCut}

TOY = np.array([[1.0, 0.0], [1.1, 0.1], [0.0, 1.0], [0.1, 1.0], [0.9, -0.1],
                [0.0, 1.2]])  # tests/spectral_clusterer_test.py:34-41


def icassp_options(sigma=1, p=0.95):
  return ref_refinement.RefinementOptions(
      gaussian_blur_sigma=sigma, p_percentile=p,
      thresholding_soft_multiplier=0.01,
      thresholding_type=ref_refinement.ThresholdType.RowMax,
      refinement_sequence=ref_configs.ICASSP2018_REFINEMENT_SEQUENCE)


def staged_run(x, sigma, p, lap, min_clusters, max_clusters):
  """Run the reference op by op, keeping every intermediate."""
  out = {}
  a = ref_utils.compute_affinity_matrix(x)
  out["affinity"] = a
  opts = icassp_options(sigma, p)
  for i, name in enumerate(opts.refinement_sequence):
    a = opts.get_refinement_operator(name).refine(a)
    out["stage%d" % i] = a
  if lap in (0, 1):
    w, v = ref_utils.compute_sorted_eigenvectors(a)
    k, delta = ref_utils.compute_number_of_clusters(
        w, max_clusters=max_clusters, stop_eigenvalue=1e-2, descend=True)
  else:
    lm = ref_laplacian.compute_laplacian(a, LAP[lap])
    out["laplacian"] = lm
    w, v = ref_utils.compute_sorted_eigenvectors(lm, descend=False)
    k, delta = ref_utils.compute_number_of_clusters(
        w, max_clusters=max_clusters, descend=False)
  out["eigenvalues"] = w
  out["eigenvectors"] = v
  out["n_clusters_raw"] = np.int64(k)
  out["max_delta"] = np.float64(delta)
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=min_clusters, max_clusters=max_clusters,
      refinement_options=opts, laplacian_type=LAP[lap])
  out["labels"] = clusterer.predict(x)
  return out


def e2e_run(n, d, k, seed, lap, max_clusters, p=0.95, dtype=None):
  """One whole reference predict(); eigenvalues / eigengap results are
  captured by wrapping the reference's own utils functions (called through
  the module attribute at spectral_clusterer.py:146-167).  dtype: the embeddings are cast to
  it first (float32: the reference then computes the affinity, the refinement and
  np.linalg.eig in single precision, utils.py:32-39, :59)."""
  x = so.blobs(n, d, k, seed)
  if dtype is not None:
    x = x.astype(dtype)
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=2, max_clusters=max_clusters,
      refinement_options=icassp_options(1, p), laplacian_type=LAP[lap])
  seen = {}
  real_eig = ref_utils.compute_sorted_eigenvectors
  real_gap = ref_utils.compute_number_of_clusters

  def spy_eig(*args, **kwargs):
    w, v = real_eig(*args, **kwargs)
    seen["w"] = w
    return w, v

  def spy_gap(*args, **kwargs):
    kk, delta = real_gap(*args, **kwargs)
    seen["k"], seen["delta"] = kk, delta
    return kk, delta

  ref_utils.compute_sorted_eigenvectors = spy_eig
  ref_utils.compute_number_of_clusters = spy_gap
  try:
    t0 = time.perf_counter()
    labels = clusterer.predict(x)
    secs = time.perf_counter() - t0
  finally:
    ref_utils.compute_sorted_eigenvectors = real_eig
    ref_utils.compute_number_of_clusters = real_gap
  w = seen["w"]
  idx = so.consumed_eigen_indices(n, max_clusters, lap in (0, 1))
  return dict(params=np.array([n, d, k, seed, lap, max_clusters]),
              p_percentile=np.float64(p), consumed_index=idx,
              consumed_eigenvalues=w[idx], head_eigenvalues=w[:max_clusters + 2],
              n_clusters_raw=np.int64(seen["k"]),
              max_delta=np.float64(seen["delta"]),
              labels=labels, ref_seconds=np.float64(secs))


def float32_goldens():
  """what the REFERENCE returns for float32 embeddings (it stays in float32 end
  to end; the device promotes to float64).  Stored with the dtype the reference returned, so the
  GPU test measures the promotion's deviation instead of asserting it away."""
  for c in ((200, 32, 4, 200, 4, 7), (1000, 64, 5, 1000, 4, 20), (1000, 64, 5, 1000, 0, 7),
            (2048, 128, 4, 2048, 4, 20)):
    r = e2e_run(*c, dtype=np.float32)
    r["eigenvalue_dtype"] = np.array(str(r["consumed_eigenvalues"].dtype))
    save("e2e_f32_n%d_lap%d_max%d.npz" % (c[0], c[4], c[5]), **r)
    print("   eigenvalues come back as", r["consumed_eigenvalues"].dtype, flush=True)


def save(name, **arrays):
  path = os.path.join(GOLDEN, name)
  np.savez_compressed(path, **arrays)
  print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def constraint_goldens():
  """7. Constraint operators (N3): per-op known answers + constrained predict()."""
  rng = np.random.default_rng(41)
  x40 = so.blobs(40, 8, 3, 41)
  a_sym = ref_utils.compute_affinity_matrix(x40)
  a_gen = rng.random((40, 40))                     # non-symmetric "affinity"
  q_sym = np.zeros((40, 40))
  for i, j, v in zip(rng.integers(0, 40, 60), rng.integers(0, 40, 60),
                     rng.choice([-1.0, 1.0], 60)):
    q_sym[i, j] = q_sym[j, i] = v
  q_gen = rng.choice([-1.0, 0.0, 0.0, 1.0], size=(40, 40))
  ops = {"x40": x40, "a_sym": a_sym, "a_gen": a_gen, "q_sym": q_sym, "q_gen": q_gen}
  for aname, a in (("sym", a_sym), ("gen", a_gen)):
    for qname, q in (("sym", q_sym), ("gen", q_gen)):
      tag = "a%s_q%s" % (aname, qname)
      ops["integ_max_" + tag] = ref_constraint.AffinityIntegration(
          ref_constraint.IntegrationType.Max).adjust_affinity(a, q)
      ops["integ_avg_" + tag] = ref_constraint.AffinityIntegration(
          ref_constraint.IntegrationType.Average).adjust_affinity(a, q)
      for alpha in (0.4, 0.6, 0.9):
        ops["cp_%02d_%s" % (round(alpha * 10), tag)] = (
            ref_constraint.ConstraintPropagation(alpha).adjust_affinity(a, q))
  scores = [0, 0, 14.308253288269043, 0.12095779925584793, 0, 3.5, 0.0, 1.0, 1.0001]
  ops["turn_scores"] = np.array(scores)
  ops["turn_matrix"] = ref_constraint.ConstraintMatrix(scores, 1).compute_diagonals()
  ops["turn_matrix_t3"] = ref_constraint.ConstraintMatrix(scores, 3).compute_diagonals()
  save("constraint_ops_n40.npz", **ops)

  # Turn-to-Diarize preset end to end (constraint propagation before refinement +
  # AutoTune), on synthetic conversations with speaker-turn scores.
  for n, d, k, seed in ((120, 16, 3, 7), (300, 32, 4, 11), (700, 64, 5, 13)):
    x, truth, sc = so.turn_blobs(n, d, k, seed)
    q = ref_constraint.ConstraintMatrix(list(sc), threshold=1).compute_diagonals()
    # the preset singleton narrows its own AutoTune range on every call
    # (autotune.py:126-131), so every run below starts from a pristine copy
    pristine = copy.deepcopy(ref_configs.turntodiarize_clusterer)
    clusterer = copy.deepcopy(pristine)
    a = ref_utils.compute_affinity_matrix(x)
    adj = clusterer.constraint_options.constraint_operator.adjust_affinity(a, q)
    grid = np.array(clusterer.autotune.get_percentile_range())
    ratios, ks = [], []
    for p in grid:
      clusterer.refinement_options.p_percentile = p
      _, kk, delta = clusterer._compute_eigenvectors_ncluster(adj, q)
      ratios.append(np.sqrt(1 - p) / delta)
      ks.append(kk)
    labels = copy.deepcopy(pristine).predict(x, q)
    unconstrained = copy.deepcopy(pristine).predict(x)
    save("turntodiarize_n%d.npz" % n, n=n, d=d, k=k, seed=seed, truth=truth, scores=sc,
         grid=grid, ratios=np.array(ratios), n_clusters=np.array(ks), labels=labels,
         labels_unconstrained=unconstrained,
         adjusted_checksum=np.array([adj.sum(), np.abs(adj).max(), adj[0, 1], adj[n // 2, n // 3]]))

  # AffinityIntegration after refinement (the reference's own 6x2 test shape,
  # tests/spectral_clusterer_test.py:243-286) on a larger conversation.
  x, truth, sc = so.turn_blobs(200, 16, 3, 17)
  q = ref_constraint.ConstraintMatrix(list(sc), threshold=1).compute_diagonals()
  q = np.maximum(q, 0) + np.eye(200)  # must-links only, ones on the diagonal
  out = {}
  for tag, kind in (("max", ref_constraint.IntegrationType.Max),
                    ("avg", ref_constraint.IntegrationType.Average)):
    opts = ref_refinement.RefinementOptions(
        p_percentile=0.9, thresholding_type=ref_refinement.ThresholdType.Percentile,
        thresholding_with_binarization=True, thresholding_preserve_diagonal=True,
        symmetrize_type=ref_refinement.SymmetrizeType.Average,
        refinement_sequence=ref_configs.TURNTODIARIZE_REFINEMENT_SEQUENCE)
    clusterer = ref_sc.SpectralClusterer(
        max_clusters=6, refinement_options=opts,
        constraint_options=ref_constraint.ConstraintOptions(
            constraint_name=ref_constraint.ConstraintName.AffinityIntegration,
            apply_before_refinement=False, integration_type=kind),
        laplacian_type=LAP[4], row_wise_renorm=True)
    out["labels_" + tag] = clusterer.predict(x, q)
    v, kk, delta = clusterer._compute_eigenvectors_ncluster(
        ref_utils.compute_affinity_matrix(x), q)
    out["n_clusters_" + tag] = np.int64(kk)
    out["max_delta_" + tag] = np.float64(delta)
  save("integration_n200.npz", truth=truth, scores=sc, q=q, **out)


def general_goldens():
  """8. Non-symmetric refined matrix (N2): the reference's AutoTune test shape
  (tests/spectral_clusterer_test.py:156-241: [RowWiseThreshold] + GraphCut), where
  np.linalg.eig works on a genuinely general matrix."""
  for n, d, k, seed in ((60, 8, 3, 5), (300, 16, 4, 6)):
    x = so.blobs(n, d, k, seed)
    opts = ref_refinement.RefinementOptions(
        thresholding_type=ref_refinement.ThresholdType.Percentile,
        refinement_sequence=[ref_refinement.RefinementName.RowWiseThreshold])
    tuner = ref_autotune.AutoTune(p_percentile_min=0.60, p_percentile_max=0.95,
                                  init_search_step=0.05, search_level=1)
    grid = np.array(tuner.get_percentile_range())
    clusterer = ref_sc.SpectralClusterer(
        min_clusters=2, max_clusters=6, refinement_options=opts, autotune=tuner,
        laplacian_type=LAP[4], row_wise_renorm=True)
    a = ref_utils.compute_affinity_matrix(x)
    ratios, ks, deltas, w2 = [], [], [], []
    for p in grid:
      clusterer.refinement_options.p_percentile = p
      refined = ref_refinement.RowWiseThreshold(
          p, 0.01, ref_refinement.ThresholdType.Percentile, False, False).refine(a)
      lap = ref_laplacian.compute_laplacian(refined, LAP[4])
      w, _ = ref_utils.compute_sorted_eigenvectors(lap, descend=False)
      _, kk, delta = clusterer._compute_eigenvectors_ncluster(a)
      ratios.append(np.sqrt(1 - p) / delta)
      ks.append(kk)
      deltas.append(delta)
      w2.append(w[:8])
    labels = clusterer.predict(x)
    save("general_n%d.npz" % n, n=n, d=d, k=k, seed=seed, grid=grid,
         ratios=np.array(ratios), n_clusters=np.array(ks), max_delta=np.array(deltas),
         eigenvalues=np.array(w2), labels=labels)


def size_reduction_goldens():
  """9. max_spectral_size (N4): AHC pre-clustering + spectral on the centroids."""
  out = {}
  # the reference's own test shape (tests/spectral_clusterer_test.py:71-89), seeded noise
  rng = np.random.default_rng(71)
  base = np.array([[1.0, 0, 0, 0, 0, 0]] * 400 + [[0, 1.0, 0, 0, 0, 0]] * 300 +
                  [[0, 0, 2.0, 0, 0, 0]] * 200 + [[0, 0, 0, 1.0, 0, 0]] * 100)
  x = base + (rng.random((1000, 6)) * 2 - 1) * 0.1
  clusterer = ref_sc.SpectralClusterer(
      refinement_options=icassp_options(sigma=0), max_spectral_size=100)
  from sklearn.cluster import AgglomerativeClustering
  out["x_1000by6"] = x
  out["ahc_1000by6"] = AgglomerativeClustering(
      n_clusters=100, metric="cosine", linkage="complete").fit_predict(x)
  out["labels_1000by6"] = clusterer.predict(x)
  for tag, (n, d, k, seed, mss, lap) in {"a": (1500, 32, 5, 91, 200, 0),
                                         "b": (2500, 64, 4, 92, 300, 4)}.items():
    xx = so.blobs(n, d, k, seed)
    c = ref_sc.SpectralClusterer(min_clusters=2, max_clusters=7,
                                 refinement_options=icassp_options(),
                                 laplacian_type=LAP[lap], max_spectral_size=mss)
    out["ahc_" + tag] = AgglomerativeClustering(
        n_clusters=mss, metric="cosine", linkage="complete").fit_predict(xx)
    out["labels_" + tag] = c.predict(xx)
  # average linkage with a distance threshold (the agglomerative fallback's form,
  # fallback_clusterer.py:108-113)
  xx = so.blobs(400, 16, 6, 93)
  for thr in (0.3, 0.5):
    out["avg_thr%02d" % round(thr * 10)] = AgglomerativeClustering(
        n_clusters=None, metric="cosine", linkage="average",
        distance_threshold=thr).fit_predict(xx)
  save("size_reduction.npz", **out)


def fallback_goldens():
  """10. Fallback decisions + multi-stage streaming (N4)."""
  from spectralcluster import fallback_clusterer as ref_fb
  from spectralcluster import multi_stage_clusterer as ref_ms
  from spectralcluster import naive_clusterer as ref_naive
  out = {}
  # naive clusterer on a stream of noisy speakers
  x = so.blobs(300, 16, 4, 101, noise=0.6)
  for tag, (thr, ad) in {"t5": (0.5, None), "t7a9": (0.7, 0.9)}.items():
    nc = ref_naive.NaiveClusterer(thr, ad)
    out["naive_" + tag] = nc.predict(x)
    out["naive_counts_" + tag] = np.array([c.count for c in nc.centroids])
    out["naive_cent_" + tag] = np.stack([c.embedding for c in nc.centroids])
  # single-cluster conditions on affinities of one / several speakers
  one = so.blobs(80, 16, 1, 102, noise=0.2)
  many = so.blobs(80, 16, 3, 103, noise=0.2)
  for name, xx in (("one", one), ("many", many)):
    a = ref_utils.compute_affinity_matrix(xx)
    out["stats_" + name] = np.array([a.min(), np.diag(a, k=1).min(), a.mean(), np.std(a)])
    for cond in ("AllAffinity", "NeighborAffinity", "AffinityStd", "FallbackClusterer"):
      for thr in (0.5, 0.75, 0.9):
        opts = ref_fb.FallbackOptions(
            single_cluster_condition=getattr(ref_fb.SingleClusterCondition, cond),
            single_cluster_affinity_threshold=thr,
            fallback_clusterer_type=ref_fb.FallbackClustererType.Agglomerative)
        out["single_%s_%s_%02d" % (name, cond, round(thr * 100))] = np.bool_(
            ref_fb.check_single_cluster(opts, xx, a))
    # the GMM start is randomly seeded in the reference: record a majority over seeds
    votes = [ref_fb.check_single_cluster(ref_fb.FallbackOptions(), xx, a) for _ in range(5)]
    out["single_%s_gmm" % name] = np.bool_(sum(votes) >= 3)
    out["single_%s_gmm_votes" % name] = np.array(votes)
  # predict() with min_clusters=1 and with too few embeddings
  for name, xx in (("one", one), ("many", many)):
    c = ref_sc.SpectralClusterer(min_clusters=1, max_clusters=6,
                                 refinement_options=icassp_options(),
                                 fallback_options=ref_fb.FallbackOptions(
                                     single_cluster_condition=ref_fb.SingleClusterCondition.AffinityStd,
                                     single_cluster_affinity_threshold=0.05))
    out["predict_min1_" + name] = c.predict(xx)
  c = ref_sc.SpectralClusterer(refinement_options=icassp_options(),
                               fallback_options=ref_fb.FallbackOptions(
                                   spectral_min_embeddings=100,
                                   fallback_clusterer_type=ref_fb.FallbackClustererType.Agglomerative,
                                   agglomerative_threshold=0.4))
  out["predict_few"] = c.predict(many)
  # multi-stage streaming: labels after selected steps of a seeded stream
  rng = np.random.default_rng(104)
  base = np.array([[1.0, 0, 0, 0, 0, 0]] * 100 + [[0, 1.0, 0, 0, 0, 0]] * 200 +
                  [[0, 0, 2.0, 0, 0, 0]] * 300 + [[0, 0, 0, 1.0, 0, 0]] * 400)
  stream = base + (rng.random((1000, 6)) * 2 - 1) * 0.02
  stream = stream[rng.permutation(1000)][:360]
  out["stream"] = stream
  for tag, defl in (("none", ref_ms.Deflicker.NoDeflicker),
                    ("hungarian", ref_ms.Deflicker.Hungarian)):
    opts = ref_refinement.RefinementOptions(
        gaussian_blur_sigma=0, p_percentile=0.2,
        refinement_sequence=ref_configs.ICASSP2018_REFINEMENT_SEQUENCE)
    main = ref_sc.SpectralClusterer(refinement_options=opts, stop_eigenvalue=0.01)
    ms = ref_ms.MultiStageClusterer(main_clusterer=main, fallback_threshold=0.5, L=20,
                                    U1=60, U2=120, deflicker=defl)
    for step, e in enumerate(stream, 1):
      labels = ms.streaming_predict(e)
      if step in (10, 30, 61, 119, 120, 121, 200, 360):
        out["ms_%s_%d" % (tag, step)] = np.asarray(labels)
  save("fallback.npz", **out)


def kmeans_metric_goldens():
  """4b. run_kmeans with the other custom_dist values and with the plain KMeans."""
  km = {}
  for tag, (n, k, seed) in {"a": (500, 4, 11), "b": (1500, 7, 12), "c": (300, 2, 13)}.items():
    r2 = np.random.default_rng(seed)
    cent = r2.standard_normal((k, k))
    e = cent[r2.integers(0, k, n)] * 0.6 + 0.35 * r2.standard_normal((n, k))
    km["e_" + tag] = e
    for metric in ("euclidean", "sqeuclidean", "cityblock", "chebyshev",
                   "correlation", "braycurtis", "canberra", "minkowski"):  # (last four: round 5)
      km["labels_%s_%s" % (tag, metric)] = ref_kmeans.run_kmeans(e, k, metric, 300)
  save("kmeans_metrics.npz", **km)


def kmeans_bigk_golden():
  """4c. ADVICE r5: the correlation metric with MORE than 128 clusters -- scipy centres every
  row by `mean(axis=1)`, and numpy's mean of a row longer than 128 splits it pairwise
  (loops_utils.h pairwise_sum); the other goldens stop at k = 12."""
  r2 = np.random.default_rng(15)
  n, k = 900, 150
  cent = r2.standard_normal((k, k))
  e = cent[r2.integers(0, k, n)] * 0.6 + 0.35 * r2.standard_normal((n, k))
  save("kmeans_correlation_k150.npz", e=e,
       labels=ref_kmeans.run_kmeans(e, k, "correlation", 300))


class _Spy:
  """Captures what the reference's own eigen / eigengap calls return (they are
  looked up through the module attribute at spectral_clusterer.py:146-167)."""

  def __enter__(self):
    self.calls = []
    self._eig = ref_utils.compute_sorted_eigenvectors
    self._gap = ref_utils.compute_number_of_clusters

    def spy_eig(*args, **kwargs):
      w, v = self._eig(*args, **kwargs)
      self.calls.append({"w": w})
      return w, v

    def spy_gap(*args, **kwargs):
      kk, delta = self._gap(*args, **kwargs)
      self.calls[-1]["k"], self.calls[-1]["delta"] = kk, delta
      return kk, delta

    ref_utils.compute_sorted_eigenvectors = spy_eig
    ref_utils.compute_number_of_clusters = spy_gap
    return self

  def __exit__(self, *exc):
    ref_utils.compute_sorted_eigenvectors = self._eig
    ref_utils.compute_number_of_clusters = self._gap


def autotune4096_golden():
  """11. BASELINE config 4 at full size: AutoTune over 16 p_percentile values,
  n=4096 d=256, ICASSP2018 + GraphCut, max_clusters=20 (autotune.py:76-132,
  spectral_clusterer.py:266-289)."""
  n, d, k, seed, max_clusters = 4096, 256, 8, 4096, 20
  x = so.blobs(n, d, k, seed)
  tuner = ref_autotune.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                init_search_step=0.025, search_level=1)
  grid = np.array(tuner.get_percentile_range())
  assert len(grid) == 16
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=2, max_clusters=max_clusters,
      refinement_options=icassp_options(), autotune=tuner,
      laplacian_type=LAP[4])
  idx = so.consumed_eigen_indices(n, max_clusters, False)
  t0 = time.perf_counter()
  with _Spy() as spy:
    labels = clusterer.predict(x)
  secs = time.perf_counter() - t0
  sweep = spy.calls[:16]
  assert len(spy.calls) == 16  # predict() re-uses the eigenvectors of the best p
  ratios = np.array([np.sqrt(1 - p) / c["delta"] for p, c in zip(grid, sweep)])
  save("autotune_n4096.npz",
       params=np.array([n, d, k, seed, 4, max_clusters]), grid=grid,
       ratios=ratios, n_clusters=np.array([c["k"] for c in sweep]),
       max_delta=np.array([c["delta"] for c in sweep]),
       consumed_index=idx,
       consumed_eigenvalues=np.stack([c["w"][idx] for c in sweep]),
       best_p=np.float64(clusterer.refinement_options.p_percentile),
       labels=labels.astype(np.int8), ref_seconds=np.float64(secs))


def ttd_options(p=0.95):
  """configs.py:53-59 (a fresh object: the preset singleton is mutated by AutoTune)."""
  return ref_refinement.RefinementOptions(
      p_percentile=p, thresholding_soft_multiplier=0.01,
      thresholding_type=ref_refinement.ThresholdType.Percentile,
      thresholding_with_binarization=True, thresholding_preserve_diagonal=True,
      symmetrize_type=ref_refinement.SymmetrizeType.Average,
      refinement_sequence=ref_configs.TURNTODIARIZE_REFINEMENT_SEQUENCE)


def autotune4096_ttd_golden():
  """11b. BASELINE config 4, secondary variant: the same 16-value AutoTune
  at n=4096 under the Turn-to-Diarize refinement (Percentile + binarise + preserve-diag +
  Average, configs.py:49-59), GraphCut, max_clusters=20, row_wise_renorm like the preset
  (configs.py:76-85); no constraint matrix."""
  n, d, k, seed, max_clusters = 4096, 256, 8, 4096, 20
  x = so.blobs(n, d, k, seed)
  tuner = ref_autotune.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                init_search_step=0.025, search_level=1)
  grid = np.array(tuner.get_percentile_range())
  assert len(grid) == 16
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=2, max_clusters=max_clusters, refinement_options=ttd_options(),
      autotune=tuner, laplacian_type=LAP[4], row_wise_renorm=True)
  idx = so.consumed_eigen_indices(n, max_clusters, False)
  t0 = time.perf_counter()
  with _Spy() as spy:
    labels = clusterer.predict(x)
  secs = time.perf_counter() - t0
  sweep = spy.calls[:16]
  assert len(spy.calls) == 16
  ratios = np.array([np.sqrt(1 - p) / c["delta"] for p, c in zip(grid, sweep)])
  print("ratios", ratios, "argmin", int(np.argmin(ratios)), flush=True)
  save("autotune_ttd_n4096.npz",
       params=np.array([n, d, k, seed, 4, max_clusters]), grid=grid,
       ratios=ratios, n_clusters=np.array([c["k"] for c in sweep]),
       max_delta=np.array([c["delta"] for c in sweep]),
       consumed_index=idx,
       consumed_eigenvalues=np.stack([np.real(c["w"])[idx] for c in sweep]),
       final_p=np.float64(clusterer.refinement_options.p_percentile),
       best_p=np.float64(grid[int(np.argmin(ratios))]),
       labels=labels.astype(np.int8), ref_seconds=np.float64(secs))


def batch512_inputs():
  """BASELINE config 5: sizes / cluster counts of the batch."""
  rng = np.random.default_rng(512)
  ns = rng.integers(300, 3001, 512)
  ks = rng.integers(2, 8, 512)
  return ns, ks


def batch512_golden(limit=None):
  """12. BASELINE config 5 at full size: 512 independent predict() calls of
  configs.icassp2018_clusterer (configs.py:37-43)."""
  ns, ks = batch512_inputs()
  if limit:
    ns, ks = ns[:limit], ks[:limit]
  labels, ncl, deltas, eig, secs = [], [], [], [], []
  t_all = time.perf_counter()
  for i, (n, k) in enumerate(zip(ns, ks)):
    x = so.blobs(int(n), 256, int(k), seed=i)
    clusterer = copy.deepcopy(ref_configs.icassp2018_clusterer)
    t0 = time.perf_counter()
    with _Spy() as spy:
      lab = clusterer.predict(x)
    secs.append(time.perf_counter() - t0)
    c = spy.calls[-1]
    idx = so.consumed_eigen_indices(int(n), 7, True)
    labels.append(lab.astype(np.int8))
    ncl.append(c["k"])
    deltas.append(c["delta"])
    eig.append(c["w"][idx])
    if i % 16 == 0:
      print("batch512 %d/%d  %.0f s" % (i, len(ns), time.perf_counter() - t_all),
            flush=True)
  save("batch512.npz", ns=ns, ks=ks, labels=np.concatenate(labels),
       n_clusters_raw=np.array(ncl), max_delta=np.array(deltas),
       consumed_eigenvalues=np.stack(eig), ref_seconds=np.array(secs))


def dense_goldens():
  """13. Every eigenvalue is consumed (E1 dense path): the reference's DEFAULT
  max_clusters=None with a Laplacian (spectral_clusterer.py:32, utils.py:100-115), and the
  ascending NormalizedDiff gap, which divides by np.max(eigenvalues) (utils.py:110)."""
  gap = {1: ref_utils.EigenGapType.Ratio, 2: ref_utils.EigenGapType.NormalizedDiff}
  # (n, d, k, seed, lap, max_clusters, eigengap_type)
  cases = [(1000, 64, 5, 1000, 4, None, 1), (2048, 128, 4, 2048, 4, None, 1),
           (1000, 64, 5, 1000, 2, None, 1), (1000, 64, 5, 1000, 3, None, 1),
           (700, 48, 6, 700, 4, None, 2), (1000, 64, 5, 1000, 4, 20, 2),
           (1000, 64, 5, 1000, 2, 12, 2), (600, 32, 3, 600, 0, None, 1)]
  for n, d, k, seed, lap, maxc, gt in cases:
    x = so.blobs(n, d, k, seed)
    clusterer = ref_sc.SpectralClusterer(
        min_clusters=2, max_clusters=maxc, refinement_options=icassp_options(),
        laplacian_type=LAP[lap], eigengap_type=gap[gt])
    t0 = time.perf_counter()
    with _Spy() as spy:
      labels = clusterer.predict(x)
    secs = time.perf_counter() - t0
    c = spy.calls[-1]
    save("dense_n%d_lap%d_max%s_gap%d.npz" % (n, lap, maxc or "None", gt),
         params=np.array([n, d, k, seed, lap, maxc or 0, gt]),
         eigenvalues=c["w"], n_clusters_raw=np.int64(c["k"]),
         max_delta=np.float64(c["delta"]), labels=labels.astype(np.int8),
         ref_seconds=np.float64(secs))


HARD_CASES = [(kind, n, lap) for kind in so.HARD_KINDS for n in (1000, 2048, 4096)
              for lap in (0, 4)]


def hard_goldens(only_n=None, only_kinds=None):
  """14. Unfriendly spectra: unstructured / overlapping / many-cluster /
  unbalanced / interleaved inputs at n in {1000, 2048, 4096}, laplacian None (max 7) and
  GraphCut (max 20), ICASSP2018 refinement, d=256.  The reference's eigensolver
  (np.linalg.eig, utils.py:59) always returns; so must the device path."""
  for kind, n, lap in HARD_CASES:
    if (only_n and n not in only_n) or (only_kinds and kind not in only_kinds):
      continue
    maxc = 7 if lap == 0 else 20
    seed = 7000 + n
    x = so.hard_inputs(kind, n, 256, seed)
    clusterer = ref_sc.SpectralClusterer(
        min_clusters=2, max_clusters=maxc, refinement_options=icassp_options(),
        laplacian_type=LAP[lap])
    t0 = time.perf_counter()
    with _Spy() as spy:
      labels = clusterer.predict(x)
    secs = time.perf_counter() - t0
    c = spy.calls[-1]
    w = np.real(c["w"])
    idx = so.consumed_eigen_indices(n, maxc, lap == 0, w, 1e-2)
    save("hard_%s_n%d_lap%d.npz" % (kind, n, lap),
         params=np.array([n, 256, seed, lap, maxc]), kind=np.array(kind),
         consumed_index=idx, consumed_eigenvalues=w[idx], head_eigenvalues=w[:maxc + 4],
         n_clusters_raw=np.int64(c["k"]), max_delta=np.float64(c["delta"]),
         labels=labels.astype(np.int8), ref_seconds=np.float64(secs))
    print("  %s n=%d lap=%d: k=%d delta=%.6g head=%s  %.1f s" % (
        kind, n, lap, c["k"], c["delta"], np.array2string(w[:6], precision=5), secs),
          flush=True)


def many_cluster_goldens():
  """15. Eigengap decisions beyond 64 clusters (`max_clusters` > 64 AND more than 64 selected):
  n = 1500 samples of 90 speakers, max_clusters = 120, ICASSP2018 refinement, laplacian None
  and GraphCut.  Groundwork: the device path of round 3 raises UnsupportedOnDeviceError when
  more than 64 clusters are SELECTED; these fixtures pin the oracle for
  the day it does not (reference spectral_clusterer.py:29-46, utils.py:100-128)."""
  for lap in (0, 4):
    r = e2e_run(1500, 256, 90, 1590, lap, 120)
    save("manyk_n1500_k90_lap%d_max120.npz" % lap, **r)
    print("  lap=%d: n_clusters_raw=%d, distinct labels %d, %.1f s" % (
        lap, int(r["n_clusters_raw"]), len(np.unique(r["labels"])), float(r["ref_seconds"])),
          flush=True)


def general_wide_golden():
  """16. More than 32 eigenpairs of a genuinely non-symmetric refined matrix: [RowWiseThreshold
  (Percentile)] + GraphCut as in section 8, n = 400 samples of 36 speakers, min_clusters = 40,
  max_clusters = 48.
  Groundwork: the device's general path of round 3 holds at most 32 eigenpairs for n > 64
  (DESIGN.md section 6)."""
  n, d, k, seed, maxc, p = 400, 32, 36, 1636, 48, 0.9
  x = so.blobs(n, d, k, seed)
  opts = ref_refinement.RefinementOptions(
      p_percentile=p, thresholding_type=ref_refinement.ThresholdType.Percentile,
      refinement_sequence=[ref_refinement.RefinementName.RowWiseThreshold])
  minc = 40  # (the eigengap alone says 2 here; min_clusters makes 40 eigenVECTORS the embedding)
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=minc, max_clusters=maxc, refinement_options=opts, laplacian_type=LAP[4],
      row_wise_renorm=True)
  with _Spy() as spy:
    labels = clusterer.predict(x)
  c = spy.calls[-1]
  w = np.real(c["w"])
  save("general_wide_n400.npz", params=np.array([n, d, k, seed, 4, maxc]),
       min_clusters=np.int64(minc),
       p_percentile=np.float64(p), head_eigenvalues=w[:maxc + 2],
       n_clusters_raw=np.int64(c["k"]), max_delta=np.float64(c["delta"]), labels=labels)
  print("  general wide: k=%d delta=%.6g distinct labels %d" % (
      c["k"], c["delta"], len(np.unique(labels))), flush=True)


def general_dense_goldens():
  """17. The general (non-symmetrisable) path where np.linalg.eig's WHOLE spectrum is read, or
  more of it than a Krylov basis holds (round 5: the device's dense Hessenberg route):
    a/b  [RowWiseThreshold] + GraphCut with max_clusters=None -- the ascending eigengap loop reads
         every eigenvalue (utils.py:100-115) -- at n = 300 and n = 1000;
    c    [RowWiseThreshold (Percentile)] + GraphCut, n = 500 samples of 70 speakers,
         max_clusters = 80: 81 values read, 70ish eigenvectors used;
    d    [RowWiseThreshold] without a Laplacian, max_clusters=None: the descending loop reads on
         until an eigenvalue falls below stop_eigenvalue = 1e-2 (utils.py:116-128).
  Every case stores the full real-part spectrum in the reference's order."""
  cases = [
      ("general_dense_n300_lap4", 300, 32, 5, 1701, 4, None, 2, 0.95, "rowmax"),
      ("general_dense_n1000_lap4", 1000, 48, 6, 1702, 4, None, 2, 0.95, "rowmax"),
      ("general_dense_n500_max80", 500, 32, 70, 1703, 4, 80, 66, 0.9, "pct"),
      ("general_dense_n300_lap0", 300, 32, 5, 1704, 0, None, 2, 0.95, "rowmax"),
  ]
  for name, n, d, k, seed, lap, maxc, minc, p, ttype in cases:
    x = so.blobs(n, d, k, seed)
    opts = ref_refinement.RefinementOptions(
        p_percentile=p, thresholding_soft_multiplier=0.01,
        thresholding_type=(ref_refinement.ThresholdType.Percentile if ttype == "pct"
                           else ref_refinement.ThresholdType.RowMax),
        refinement_sequence=[ref_refinement.RefinementName.RowWiseThreshold])
    clusterer = ref_sc.SpectralClusterer(
        min_clusters=minc, max_clusters=maxc, refinement_options=opts, laplacian_type=LAP[lap])
    t0 = time.perf_counter()
    with _Spy() as spy:
      labels = clusterer.predict(x)
    secs = time.perf_counter() - t0
    c = spy.calls[-1]
    w = np.real(c["w"])
    save(name + ".npz", params=np.array([n, d, k, seed, lap, -1 if maxc is None else maxc]),
         min_clusters=np.int64(minc), p_percentile=np.float64(p),
         percentile=np.int64(ttype == "pct"), eigenvalues=w,
         n_clusters_raw=np.int64(c["k"]), max_delta=np.float64(c["delta"]), labels=labels,
         ref_seconds=np.float64(secs))
    print("  %s: k=%d delta=%.6g distinct labels %d, %.1f s" % (
        name, c["k"], c["delta"], len(np.unique(labels)), secs), flush=True)


def main():
  os.makedirs(GOLDEN, exist_ok=True)
  large = "--large" in sys.argv
  if "--many-clusters" in sys.argv:  # only section 15
    many_cluster_goldens()
    return
  if "--kmeans-bigk" in sys.argv:  # only section 4c
    kmeans_bigk_golden()
    return
  if "--float32" in sys.argv:  # only section 18
    float32_goldens()
    return
  if "--general-wide" in sys.argv:  # only section 16
    general_wide_golden()
    return
  if "--general-dense" in sys.argv:  # only section 17
    general_dense_goldens()
    return
  if "--hard" in sys.argv:  # only section 14 (optionally: --hard kind [kind ...])
    kinds = [a for a in sys.argv[sys.argv.index("--hard") + 1:] if a in so.HARD_KINDS]
    hard_goldens(only_kinds=kinds or None)
    return
  if "--constraints" in sys.argv:  # only section 7
    constraint_goldens()
    return
  if "--general" in sys.argv:  # only section 8
    general_goldens()
    return
  if "--size-reduction" in sys.argv:  # only section 9
    size_reduction_goldens()
    return
  if "--fallback" in sys.argv:  # only section 10
    fallback_goldens()
    return
  if "--kmeans-metrics" in sys.argv:  # only section 4b
    kmeans_metric_goldens()
    return
  if "--autotune4096" in sys.argv:  # only section 11
    autotune4096_golden()
    return
  if "--autotune4096-ttd" in sys.argv:  # only section 11b
    autotune4096_ttd_golden()
    return
  if "--batch512" in sys.argv:  # only section 12
    batch512_golden()
    return
  if "--dense" in sys.argv:  # only section 13
    dense_goldens()
    return

  # 1. The 6x2 toy of the reference tests, sigma=0, full stage dump, all
  #    Laplacian types (max_clusters None: every eigenvalue is consumed).
  for lap in (0, 2, 3, 4):
    r = staged_run(TOY, 0, 0.95, lap, None, None)
    save("toy6x2_lap%d.npz" % lap, x=TOY, **r)

  # 2. n=64 blobs, sigma=1, full stage dump.
  x64 = so.blobs(64, 16, 3, 64)
  for lap in (0, 4):
    r = staged_run(x64, 1, 0.95, lap, 2, 7)
    save("stages_n64_lap%d.npz" % lap, x=x64, **r)

  # 3. Per-op known answers on a non-symmetric 40x40 matrix (every option).
  rng = np.random.default_rng(40)
  m = rng.random((40, 40))
  ops = {"input": m,
         "crop": ref_refinement.CropDiagonal().refine(m),
         "blur_s1": ref_refinement.GaussianBlur(1).refine(m),
         "blur_s2": ref_refinement.GaussianBlur(2).refine(m),
         "sym_max": ref_refinement.Symmetrize().refine(m),
         "sym_avg": ref_refinement.Symmetrize(
             ref_refinement.SymmetrizeType.Average).refine(m),
         "diffuse": ref_refinement.Diffuse().refine(m),
         "rownorm": ref_refinement.RowWiseNormalize().refine(m)}
  for tname, tt in (("rowmax", ref_refinement.ThresholdType.RowMax),
                    ("pct", ref_refinement.ThresholdType.Percentile)):
    for bz in (0, 1):
      for pd in (0, 1):
        ops["thr_%s_b%d_d%d" % (tname, bz, pd)] = ref_refinement.RowWiseThreshold(
            0.8, 0.01, tt, bool(bz), bool(pd)).refine(m)
  sm = ops["sym_max"]
  for lap in (2, 3, 4):
    ops["lap%d" % lap] = ref_laplacian.compute_laplacian(sm, LAP[lap])
  save("ops_n40.npz", **ops)

  # 4. k-means: seeds/centroids/labels from sklearn + the reference loop.
  from sklearn.cluster import KMeans
  km = {}
  for tag, (n, k, seed) in {"a": (500, 4, 1), "b": (1200, 8, 2),
                            "c": (300, 2, 3), "d": (900, 20, 4)}.items():
    r2 = np.random.default_rng(seed)
    cent = r2.standard_normal((k, k))
    e = cent[r2.integers(0, k, n)] * 0.05 + 0.02 * r2.standard_normal((n, k))
    est = KMeans(n_clusters=k, init="k-means++", max_iter=1, random_state=0,
                 n_init="auto").fit(e)
    km["e_" + tag] = e
    km["centers_" + tag] = est.cluster_centers_
    km["labels_" + tag] = ref_kmeans.run_kmeans(e, k, "cosine", 300)
  save("kmeans.npz", **km)
  kmeans_metric_goldens()

  # 5. End-to-end, seeds only.
  cases = [(200, 32, 4, 200, 0, 7), (200, 32, 4, 200, 4, 7),
           (1000, 64, 5, 1000, 0, 7), (1000, 64, 5, 1000, 4, 20),
           (1000, 64, 5, 1000, 3, 20), (1000, 64, 5, 1000, 2, 20),
           (2048, 128, 4, 2048, 0, 7), (2048, 128, 4, 2048, 4, 20)]
  for c in cases:
    r = e2e_run(*c)
    save("e2e_n%d_lap%d_max%d.npz" % (c[0], c[4], c[5]), **r)

  # 6. AutoTune sweep (config-4 shape at n=512): 16 p values.
  x = so.blobs(512, 64, 6, 512)
  tuner = ref_autotune.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                init_search_step=0.025, search_level=1)
  grid = np.array(tuner.get_percentile_range())
  clusterer = ref_sc.SpectralClusterer(
      min_clusters=2, max_clusters=20, refinement_options=icassp_options(),
      autotune=tuner, laplacian_type=LAP[4])
  a = ref_utils.compute_affinity_matrix(x)
  ratios, ks = [], []
  for p in grid:
    clusterer.refinement_options.p_percentile = p
    _, kk, delta = clusterer._compute_eigenvectors_ncluster(a)
    ratios.append(np.sqrt(1 - p) / delta)
    ks.append(kk)
  labels = clusterer.predict(x)
  save("autotune_n512.npz", grid=grid, ratios=np.array(ratios),
       n_clusters=np.array(ks), labels=labels,
       best_p=np.float64(grid[int(np.argmin(ratios))]))

  constraint_goldens()
  general_goldens()
  size_reduction_goldens()
  fallback_goldens()

  if large:
    for c in [(8192, 256, 8, 0, 4, 20), (8192, 256, 4, 1, 0, 7)]:
      r = e2e_run(*c)
      save("e2e_n%d_lap%d_max%d.npz" % (c[0], c[4], c[5]), **r)


if __name__ == "__main__":
  main()
