#!/usr/bin/env python
"""Where a Turn-to-Diarize AutoTune value spends its time (n = 4096): stage times and solver
bookkeeping per p_percentile.   [SC_EIG_TRACE=1] python tests/probes/ttd_trace.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

x = so.blobs(4096, 256, 8, 4096)
opts = sca.RefinementOptions(
    p_percentile=0.95, thresholding_soft_multiplier=0.01,
    thresholding_type=sca.ThresholdType.Percentile, thresholding_with_binarization=True,
    thresholding_preserve_diagonal=True, symmetrize_type=sca.SymmetrizeType.Average,
    refinement_sequence=sca.TURNTODIARIZE_REFINEMENT_SEQUENCE)
c = sca.SpectralClusterer(min_clusters=2, max_clusters=20, refinement_options=opts,
                          laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
h = c._handle()
c._upload(h, x)
for p in (0.55, 0.75, 0.85, 0.9, 0.95):
  c._eig_resident(h, p)
  t = time.perf_counter()
  dg = c._eig_resident(h, p)
  ms = 1e3 * (time.perf_counter() - t)
  st = {k: round(v, 2) for k, v in dg.stage_times_ms().items() if v}
  print("p=%.2f: %.2f ms  passes=%d cycles=%d basis=%d k_raw=%d  %s" % (
      p, ms, dg.eig_matvec_passes, dg.eig_cycles, dg.eig_basis, dg.n_clusters_raw, st), flush=True)
