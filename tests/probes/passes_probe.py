"""A/B probe (round 5): block passes and throughput of config 5 (batch512) and the headline call,
under whatever environment switches the caller sets (SC_EIG_NO_CAP, SC_UPLOAD_PLAIN).
python tests/probes/passes_probe.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("SC_DIFFUSE", "SC_EIG_TRACE")
               if k in os.environ) or "default"
rng = np.random.default_rng(512)
ns = rng.integers(300, 3001, 512)
ks = rng.integers(2, 8, 512)
utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
c = sca.SpectralClusterer(min_clusters=2, max_clusters=7,
                          refinement_options=sca.configs.icassp2018_refinement_options)
c.predict_batch(utts, group=16)
best = 1e9
for _ in range(reps):
  t0 = time.perf_counter()
  c.predict_batch(utts, group=16)
  best = min(best, time.perf_counter() - t0)
passes = sum(d.eig_matvec_passes for d in c.last_batch_diags)
print("[%s] batch512: %.1f utt/s (best of %d), %d block passes = %.3f per utterance" % (
    tag, 512 / best, reps, passes, passes / 512.0), flush=True)

x = so.blobs(8192, 256, 8, seed=0)
c8 = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                           refinement_options=sca.configs.icassp2018_refinement_options,
                           laplacian_type=sca.LaplacianType.GraphCut)
for _ in range(3):
  c8.predict(x)
t0 = time.perf_counter()
for _ in range(20):
  c8.predict(x)
dt = (time.perf_counter() - t0) / 20
d = c8.last_diag
print("[%s] predict8192: %.3f ms/call (python loop), passes %d, stage_ms affinity %.3f diffuse %.3f "
      "eig %.3f kmeans %.3f total %.3f free_stats %.3f" % (
          tag, 1e3 * dt, d.eig_matvec_passes, d.stage_ms[0], d.stage_ms[2], d.stage_ms[4], d.stage_ms[5],
          d.stage_ms[6], d.stage_ms[list(sca._lib.STAGE_NAMES).index("free_stats")]), flush=True)

x4 = so.blobs(4096, 256, 8, seed=4096)
tuner = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95, init_search_step=0.025,
                     search_level=1)
c4 = sca.SpectralClusterer(min_clusters=2, max_clusters=20, autotune=tuner,
                           refinement_options=sca.RefinementOptions(
                               gaussian_blur_sigma=1, p_percentile=0.95,
                               thresholding_soft_multiplier=0.01,
                               refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
                           laplacian_type=sca.LaplacianType.GraphCut)
c4.predict(x4)
t0 = time.perf_counter()
for _ in range(3):
  c4.predict(x4)
dt = (time.perf_counter() - t0) / 3
print("[%s] autotune16 predict: %.2f ms, passes per value %s" % (
    tag, 1e3 * dt, [int(dg.eig_matvec_passes) for dg in c4.last_sweep_diags]), flush=True)

# ---- the workloads whose time is host Rayleigh-Ritz + chain latency (SC_HOST_RR_FULL A/B)
xh = np.random.default_rng(8192).standard_normal((8192, 256))
for _ in range(2):
  c8.predict(xh)
t0 = time.perf_counter()
for _ in range(5):
  c8.predict(xh)
dt = (time.perf_counter() - t0) / 5
d = c8.last_diag
print("[%s] hard8192: %.2f ms/call, passes %d, eig %.2f ms, basis %d cycles %d" % (
    tag, 1e3 * dt, d.eig_matvec_passes, d.stage_ms[4], d.eig_basis, d.eig_cycles), flush=True)

opts = sca.RefinementOptions(
    thresholding_soft_multiplier=0.01, thresholding_type=sca.ThresholdType.Percentile,
    thresholding_with_binarization=True, thresholding_preserve_diagonal=True,
    symmetrize_type=sca.SymmetrizeType.Average,
    refinement_sequence=[sca.RefinementName.RowWiseThreshold, sca.RefinementName.Symmetrize])
ct = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                           autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                                 init_search_step=0.025, search_level=1),
                           laplacian_type=sca.LaplacianType.GraphCut, refinement_options=opts,
                           row_wise_renorm=True)
ct.predict(x4)
t0 = time.perf_counter()
for _ in range(3):
  ct.predict(x4)
dt = (time.perf_counter() - t0) / 3
print("[%s] autotune16_ttd predict: %.2f ms, passes per value %s" % (
    tag, 1e3 * dt, [int(dg.eig_matvec_passes) for dg in ct.last_sweep_diags]), flush=True)
