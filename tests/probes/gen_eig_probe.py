"""Convergence probe of the general (non-symmetric) eigen path: SC_EIG_TRACE=1 python tests/probes/gen_eig_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import spectral_oracle as so
import spectralcluster_amd as sca

def run(n, lap, maxc=6, p=0.9, d=16, k=3):
  x = so.blobs(n, d, k, seed=3 * n + lap)
  opts = sca.RefinementOptions(thresholding_type=sca.ThresholdType.Percentile, p_percentile=p,
                               refinement_sequence=[sca.RefinementName.RowWiseThreshold])
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                            laplacian_type=sca.LaplacianType(lap) if lap else None)
  t = time.time()
  try:
    c.predict(x)
    dg = c.last_diag
    print("n=%d lap=%d ok: passes %d cycles %d basis %d k=%d  %.1f ms" % (
        n, lap, dg.eig_matvec_passes, dg.eig_cycles, dg.eig_basis, dg.n_clusters,
        1e3 * (time.time() - t)), flush=True)
  except Exception as e:
    print("n=%d lap=%d FAILED: %s" % (n, lap, e), flush=True)

import os
for n in [int(v) for v in os.environ.get("PROBE_N", "500,2000").split(",")]:
  for lap in [int(v) for v in os.environ.get("PROBE_LAP", "0,4").split(",")]:
    run(n, lap)

if os.environ.get("PROBE_SWEEP"):
  x = so.blobs(600, 32, 5, seed=21)
  for p in np.linspace(0.6, 0.95, 7):
    opts = sca.RefinementOptions(thresholding_type=sca.ThresholdType.Percentile, p_percentile=float(p),
                                 refinement_sequence=[sca.RefinementName.RowWiseThreshold])
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=8, refinement_options=opts,
                              laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
    t = time.time()
    try:
      c.predict(x)
      dg = c.last_diag
      print("p=%.3f ok: passes %d cycles %d basis %d k=%d  %.1f ms" % (
          p, dg.eig_matvec_passes, dg.eig_cycles, dg.eig_basis, dg.n_clusters,
          1e3 * (time.time() - t)), flush=True)
    except Exception as e:
      print("p=%.3f FAILED: %s" % (p, e), flush=True)
