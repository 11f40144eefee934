"""Generate tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--large]

`--large` additionally runs the two n=8192 configurations (about 150-160 s of
CPU each).  Inputs are regenerated from seeds by `spectral_oracle.blobs`; only
small outputs are stored (consumed eigenvalues, cluster counts, labels), plus
full per-stage matrices for the tiny cases.
"""

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


def e2e_run(n, d, k, seed, lap, max_clusters, p=0.95):
  """One whole reference predict(); eigenvalues / eigengap results are
  captured by wrapping the reference's own utils functions (called through
  the module attribute at spectral_clusterer.py:146-167)."""
  x = so.blobs(n, d, k, seed)
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


def save(name, **arrays):
  path = os.path.join(GOLDEN, name)
  np.savez_compressed(path, **arrays)
  print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def main():
  os.makedirs(GOLDEN, exist_ok=True)
  large = "--large" in sys.argv

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

  if large:
    for c in [(8192, 256, 8, 0, 4, 20), (8192, 256, 4, 1, 0, 7)]:
      r = e2e_run(*c)
      save("e2e_n%d_lap%d_max%d.npz" % (c[0], c[4], c[5]), **r)


if __name__ == "__main__":
  main()
