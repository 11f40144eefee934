"""First GPU contact of the matrix-free Diffuse: stage-level check + timings at n=8192."""
import os, sys, time, ctypes, dataclasses
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so
import spectralcluster_amd as sca
from spectralcluster_amd import _lib

def rowstats(a, mode):
  h = _lib.default_handle()
  a = np.ascontiguousarray(a, dtype=np.float64)
  n = a.shape[0]
  rmax, rsum = np.empty(n), np.empty(n)
  info = (ctypes.c_int32 * 6)()
  h.check(h.lib.sc_stage_diffuse_rowstats(h.raw, _lib.as_double_p(a), n, mode,
                                          _lib.as_double_p(rmax), _lib.as_double_p(rsum), info))
  return rmax, rsum, list(info)

for n, d, k in ((300, 32, 3), (1024, 64, 4), (2048, 128, 4)):
  x = so.blobs(n, d, k, seed=n)
  cfg = so.icassp2018_config()
  a = so.refine(so.affinity(x), dataclasses.replace(cfg, sequence=tuple(cfg.sequence[:4])))
  s = a @ a.T
  for mode in (1, 2):
    rmax, rsum, info = rowstats(a, mode)
    e1 = np.max(np.abs(rmax - s.max(axis=1)) / s.max(axis=1))
    e2 = np.max(np.abs(rsum - s.sum(axis=1)) / s.sum(axis=1))
    print("n=%d mode=%d rowmax rel err %.2e rowsum rel err %.2e info %s" % (n, mode, e1, e2, info), flush=True)

opts = sca.RefinementOptions(gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
    thresholding_type=sca.ThresholdType.RowMax, refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
for name in ("e2e_n2048_lap4_max20", "e2e_n8192_lap4_max20"):
  g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  for mode in (1, 2):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                              laplacian_type=sca.LaplacianType.GraphCut)
    c.diffuse_mode = mode
    best = 1e9
    for rep in range(4):
      t0 = time.perf_counter(); labels = c.predict(x); best = min(best, time.perf_counter() - t0)
    dg = c.last_diag
    idx, ref = g["consumed_index"], g["consumed_eigenvalues"]
    w = dg.eigenvalue_array()[idx]
    print(name, "mode", mode, "path", dg.diffuse_path, "ARI", so.adjusted_rand_index(labels, g["labels"]),
          "eig err %.2e" % np.max(np.abs(w - ref) / np.maximum(np.abs(ref), 1e-12)),
          "k", dg.n_clusters_raw, int(g["n_clusters_raw"]), "cands", dg.free_candidates, "ovf", dg.free_overflow_rows,
          "passes", dg.eig_matvec_passes, "best ms %.3f" % (best * 1e3), flush=True)
    print("   stage_ms", {k2: round(v, 3) for k2, v in dg.stage_times_ms().items() if v}, flush=True)
