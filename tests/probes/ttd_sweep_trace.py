#!/usr/bin/env python
"""The Turn-to-Diarize AutoTune sweep of bench.py (16 values, n = 4096) under the solver's
trace:   SC_EIG_TRACE=1 SC_GROUP_TRACE=1 python tests/probes/ttd_sweep_trace.py [icassp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralcluster_amd as sca  # noqa: E402
from bench import blobs  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "ttd"
x, _ = blobs(4096, 256, 8, 4096)


def make():
  if variant == "ttd":
    opts = sca.RefinementOptions(
        p_percentile=0.95, thresholding_soft_multiplier=0.01,
        thresholding_type=sca.ThresholdType.Percentile, thresholding_with_binarization=True,
        thresholding_preserve_diagonal=True, symmetrize_type=sca.SymmetrizeType.Average,
        refinement_sequence=[sca.RefinementName.RowWiseThreshold, sca.RefinementName.Symmetrize])
  else:
    opts = sca.RefinementOptions(
        gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
        refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)
  return sca.SpectralClusterer(
      min_clusters=2, max_clusters=20, laplacian_type=sca.LaplacianType.GraphCut,
      refinement_options=opts, row_wise_renorm=variant == "ttd",
      autotune=sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                            init_search_step=0.025, search_level=1))


make().predict(x)
sys.stderr.write("==== second sweep\n")
sys.stderr.flush()
t = time.perf_counter()
c = make()
c.predict(x)
print("%s sweep: %.2f ms" % (variant, 1e3 * (time.perf_counter() - t)))
