"""GPU: the paths that the default configuration does not take must stay correct --
they are the repair / large-problem routes of the default ones:

  SC_EIG_HOST_CHAIN=1   host-driven orthonormalisation chain (what the fused k_lz_rows chain
                        falls back to when a block is rank deficient)
  SC_EIG_DEVICE_RR=1    one-workgroup Jacobi for the Rayleigh-Ritz problem (used above 64
                        basis vectors; the host solves the smaller ones)
  SC_KMEANS_SINGLE=1    single-workgroup k-means (k > 32, other metrics, very large n)
  SC_MATVEC_SYM_MIN_N=129  upper-triangle block matvec on every Krylov solve, not only for
                        n >= 4096 (edge tiles, restarts, the repair chain all go through it)

  SC_SWEEP_ONE_BY_ONE=1 an AutoTune level as separate sc_eig_ncluster calls (what a level falls
                        back to when member arenas do not fit or a value leaves the group;
                        the winner is then evaluated, not adopted)
  SC_EIG_FORCE_DENSE=1  the landing pad of spectra block Lanczos gives up on: eigenvalues by
                        tridiagonalisation + bisection, eigenvectors by inverse iteration +
                        Householder back-transform (sc_diag.eig_path == 6)

  SC_DIFFUSE=free       (with SC_DIFFUSE_FREE_MIN_N unset: the size rule is bypassed) the
                        matrix-free Diffuse on every golden, n = 1000 included, where the
                        default takes the explicit product
  SC_DIFFUSE=explicit   the fp64 Diffuse product on every golden, n = 2048 included, where the
                        default is matrix-free
  SC_GEN_DEVICE_RR / SC_GEN_LOOSE_BULK (with SC_GEN_DENSE_MAX_N=64), SC_FREE_NO_PRUNE, SC_NO_PREFETCH,
  SC_GROUP_EQUAL_COUNT, SC_GROUP_QUANTIZE_PASS: round 6's switches -- the Rayleigh-Ritz kernel,
  stop rule, product, uploads, group formation and grouped quantiser of round 5
  SC_GEN_DENSE_MAX_N=64 block Arnoldi (narrow and wide) on the general-path goldens of n = 300 /
                        400, where the default since round 5 is the dense Hessenberg route

Each runs in a fresh interpreter (the switches are read once per process) over reference
goldens of both Laplacian branches."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so
import spectralcluster_amd as sca
LAP = {0: None, 2: sca.LaplacianType.Unnormalized, 3: sca.LaplacianType.RandomWalk,
       4: sca.LaplacianType.GraphCut}
opts = sca.configs.icassp2018_refinement_options
for name in ("e2e_n1000_lap0_max7", "e2e_n1000_lap4_max20", "e2e_n1000_lap3_max20",
             "e2e_n2048_lap0_max7", "e2e_n2048_lap4_max20"):
  g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                            laplacian_type=LAP[lap])
  for rep in range(2):   # twice: the second call runs with whatever state the first left
    labels = c.predict(x)
    dg = c.last_diag
    idx, ref = g["consumed_index"], g["consumed_eigenvalues"]
    if lap == 0:  # the descending loop stops reading after the first value < 1e-2
      keep = so.consumed_eigen_indices(n, maxc, True, ref, 1e-2)
      idx, ref = idx[keep], ref[keep]
    w = dg.eigenvalue_array()[idx]
    assert np.max(np.abs(w - ref) / np.maximum(np.abs(ref), 1e-12)) < 1e-6, name
    assert dg.n_clusters_raw == int(g["n_clusters_raw"]), name
    assert so.adjusted_rand_index(labels, g["labels"]) == 1.0, name
  if os.environ.get("SC_EIG_HOST_CHAIN"):
    assert dg.eig_host_chain == 1
  if os.environ.get("SC_EIG_FORCE_DENSE"):
    assert dg.eig_path == 6 and dg.eig_fallback == 4
  if os.environ.get("SC_DIFFUSE") == "free" and not os.environ.get("SC_EIG_FORCE_DENSE"):
    assert dg.diffuse_path == 2, name
  if os.environ.get("SC_DIFFUSE") == "explicit":
    assert dg.diffuse_path == 1, name
# one AutoTune search (16 values) against the reference golden: per-value proxies and labels
g = np.load(os.path.join(ROOT, "tests", "golden", "autotune_n512.npz"))
x = so.blobs(512, 64, 6, 512)
tuner = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95, init_search_step=0.025,
                     search_level=1)
c = sca.SpectralClusterer(min_clusters=2, max_clusters=20, autotune=tuner,
                          refinement_options=sca.RefinementOptions(
                              gaussian_blur_sigma=1, p_percentile=0.95,
                              thresholding_soft_multiplier=0.01,
                              refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
                          laplacian_type=sca.LaplacianType.GraphCut)
labels = c.predict(x)
assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
assert c.last_best_p == float(g["best_p"])
ratios = [tuner.ratio(p, d.max_delta) for p, d in zip(g["grid"], c.last_sweep_diags)]
assert np.allclose(ratios, g["ratios"], rtol=1e-6)
km = np.load(os.path.join(ROOT, "tests", "golden", "kmeans.npz"))
for tag, k in (("a", 4), ("b", 8), ("c", 2), ("d", 20)):
  got = sca.custom_distance_kmeans.run_kmeans(km["e_" + tag], k, "cosine", 300)
  assert np.array_equal(got, km["labels_" + tag]), tag
# a long run: 76 consumed eigenvalues, basis at its cap, thick restarts (every path above
# must survive it)
n = 400
x = so.blobs(n, 200, 3, seed=11)
cfg = so.OracleConfig(sequence=(so.OP_DIFFUSE,), stop_eigenvalue=1e-2)
dump = {}
want = so.predict(x, cfg, dump)
ref = dump["eigenvalues"]
idx = so.consumed_eigen_indices(n, None, True, ref, 1e-2)
assert idx.size > 64
c = sca.SpectralClusterer(min_clusters=2, refinement_options=sca.RefinementOptions(
    refinement_sequence=[sca.RefinementName.Diffuse]))
got = c.predict(x)
w = c.consumed_eigenvalues()
assert np.max(np.abs(w[idx] - ref[idx]) / np.maximum(np.abs(ref[idx]), 1e-12)) < 1e-5
assert so.adjusted_rand_index(got, want) == 1.0
if os.environ.get("SC_EIG_FORCE_DENSE") and not os.environ.get("SC_DIFFUSE"):
  # the general (non-symmetric) path has a landing pad too: Hessenberg reduction on the device,
  # QR + inverse iteration on the host (eig_path 7), here forced on a request block Arnoldi solves
  g = np.load(os.path.join(ROOT, "tests", "golden", "general_wide_n400.npz"))
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  c = sca.SpectralClusterer(
      min_clusters=int(g["min_clusters"]), max_clusters=maxc,
      refinement_options=sca.RefinementOptions(
          thresholding_type=sca.ThresholdType.Percentile, p_percentile=float(g["p_percentile"]),
          refinement_sequence=[sca.RefinementName.RowWiseThreshold]),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = c.predict(so.blobs(n, d, k, seed))
  dg = c.last_diag
  assert dg.eig_path == 7 and dg.eig_fallback == 4
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  assert abs(dg.max_delta - float(g["max_delta"])) <= 1e-6 * float(g["max_delta"])
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
if os.environ.get("SC_GEN_DENSE_MAX_N"):
  g = np.load(os.path.join(ROOT, "tests", "golden", "general_wide_n400.npz"))
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  c = sca.SpectralClusterer(
      min_clusters=int(g["min_clusters"]), max_clusters=maxc,
      refinement_options=sca.RefinementOptions(
          thresholding_type=sca.ThresholdType.Percentile, p_percentile=float(g["p_percentile"]),
          refinement_sequence=[sca.RefinementName.RowWiseThreshold]),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  labels = c.predict(so.blobs(n, d, k, seed))
  dg = c.last_diag
  assert dg.eig_path == 4, dg.eig_path  # wide block Arnoldi
  assert dg.n_clusters_raw == int(g["n_clusters_raw"])
  assert abs(dg.max_delta - float(g["max_delta"])) <= 1e-6 * float(g["max_delta"])
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  g = np.load(os.path.join(ROOT, "tests", "golden", "general_n300.npz"))
  x = so.blobs(int(g["n"]), int(g["d"]), int(g["k"]), int(g["seed"]))
  tuner = sca.AutoTune(p_percentile_min=0.60, p_percentile_max=0.95, init_search_step=0.05,
                       search_level=1)
  c = sca.SpectralClusterer(
      min_clusters=2, max_clusters=6, autotune=tuner, laplacian_type=sca.LaplacianType.GraphCut,
      row_wise_renorm=True, refinement_options=sca.RefinementOptions(
          thresholding_type=sca.ThresholdType.Percentile,
          refinement_sequence=[sca.RefinementName.RowWiseThreshold]))
  labels = c.predict(x)
  assert c.last_diag.eig_path == 4  # narrow block Arnoldi
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
if os.environ.get("SC_KMEANS_SINGLE"):
  # every member of a grouped batch is handed back after its front (the lockstep k-means chain
  # is switched off): the large ones took the matrix-free Diffuse and must be resumed on the
  # two-pass operator (FrontResult.free_op)
  from spectralcluster_amd import _lib
  utts = [so.blobs(m, 48, 4, seed=m) for m in (1650, 700, 2100)]
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=12, refinement_options=opts,
                            laplacian_type=sca.LaplacianType.GraphCut)
  got = c.predict_batch(utts, group=8)
  bd = c.last_batch_diags
  for i, u in enumerate(utts):
    want = c.predict(u)
    assert bd[i].n_clusters_raw == c.last_diag.n_clusters_raw, i
    assert abs(bd[i].max_delta - c.last_diag.max_delta) <= 1e-6 * abs(c.last_diag.max_delta), i
    assert np.array_equal(got[i], want), i
    if u.shape[0] >= 1536:
      assert bd[i].diffuse_path in (_lib.DIFFUSE_PATH_FREE, _lib.DIFFUSE_PATH_FREE_THEN_EXPLICIT)
if (os.environ.get("SC_NO_PREFETCH") or os.environ.get("SC_GROUP_EQUAL_COUNT") or
    os.environ.get("SC_GROUP_QUANTIZE_PASS")):
  # batches with the round-5 behaviours (every call uploads for itself; groups of equal count):
  # members against single calls, a plain sequence and a grouped batch
  utts = [so.blobs(m, 64, 4, seed=m) for m in (1300, 640, 1500, 380, 1700, 900, 2100, 450)]
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=12, refinement_options=opts,
                            laplacian_type=sca.LaplacianType.GraphCut)
  singles = [c.predict(u) for u in utts]
  for got in (c.predict_batch(utts, streams=1), c.predict_batch(utts, group=4)):
    for a, b in zip(got, singles):
      assert so.adjusted_rand_index(a, b) == 1.0
print("ALTERNATE_PATH_OK")
"""


@pytest.mark.parametrize("switch", ["SC_EIG_HOST_CHAIN", "SC_EIG_DEVICE_RR", "SC_KMEANS_SINGLE",
                                    "SC_MATVEC_SYM_MIN_N", "SC_SWEEP_ONE_BY_ONE",
                                    "SC_EIG_FORCE_DENSE", "SC_DIFFUSE=free", "SC_DIFFUSE=explicit",
                                    "SC_GEN_DENSE_MAX_N=64",
                                    "SC_GEN_DENSE_MAX_N=64+SC_GEN_DEVICE_RR",
                                    "SC_GEN_DENSE_MAX_N=64+SC_GEN_LOOSE_BULK",
                                    "SC_FREE_NO_PRUNE", "SC_DIFFUSE=free+SC_FREE_NO_PRUNE",
                                    "SC_NO_PREFETCH", "SC_GROUP_EQUAL_COUNT",
                                    "SC_GROUP_QUANTIZE_PASS",
                                    "SC_DIFFUSE=free+SC_EIG_HOST_CHAIN",
                                    "SC_DIFFUSE=free+SC_EIG_FORCE_DENSE",
                                    "SC_DIFFUSE=free+SC_MATVEC_SYM_MIN_N"])
def test_alternate_path(tmp_path, switch):
  script = tmp_path / "alt.py"
  script.write_text(_SCRIPT)
  env = dict(os.environ)
  for item in switch.split("+"):
    name, _, value = item.partition("=")
    env[name] = value or ("129" if name == "SC_MATVEC_SYM_MIN_N" else "1")
  r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True,
                     timeout=600, env=env)
  assert r.returncode == 0 and "ALTERNATE_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
