"""Block Arnoldi (general eigen path) against the oracle with EVERY consumed eigenvalue on the
1e-5 bar: [RowWiseThreshold]-only sequences x every Laplacian x both eigengap rules at
n in {600, 1000, 2000}.  Prints per case: eigen path, block passes, restart
cycles, stage time, worst consumed-eigenvalue error, cluster counts, ARI.

  python tests/probes/general_strict_probe.py [sizes...]
  SC_GEN_LOOSE_BULK=1 python tests/probes/general_strict_probe.py     # rounds 3-5's stop rule
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [600, 1000, 2000]
tot_pass = 0
worst_all = 0.0
for n in sizes:
  for lap in (0, 1, 2, 3, 4):
    for gap in ("Ratio", "NormalizedDiff"):
      maxc = 8
      x = so.blobs(n, 32, 4, seed=7 * n + lap)
      cfg = so.OracleConfig(
          min_clusters=2, max_clusters=maxc, sequence=(so.OP_ROW_WISE_THRESHOLD,),
          threshold_type=so.THRESHOLD_PERCENTILE, p_percentile=0.9, laplacian_type=lap,
          eigengap_type=so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF)
      dump = {}
      want = so.predict(x, cfg, dump)
      c = sca.SpectralClusterer(
          min_clusters=2, max_clusters=maxc,
          refinement_options=sca.RefinementOptions(
              thresholding_type=sca.ThresholdType.Percentile, p_percentile=0.9,
              refinement_sequence=[sca.RefinementName.RowWiseThreshold]),
          laplacian_type=sca.LaplacianType(lap) if lap else None,
          eigengap_type=getattr(sca.EigenGapType, gap))
      c.predict(x)
      t0 = time.perf_counter()
      got = c.predict(x)
      dt = time.perf_counter() - t0
      dg = c.last_diag
      descend = lap in (0, 1)
      idx = so.consumed_eigen_indices(n, maxc, descend, dump["eigenvalues"] if descend else None,
                                      1e-2 if descend else None)
      w = dg.eigenvalue_array()
      ref = dump["eigenvalues"]
      idx = [i for i in idx if i < len(w)]
      err = max(abs(w[i] - ref[i]) / max(abs(ref[i]), 1e-12) for i in idx)
      worst_all = max(worst_all, err)
      tot_pass += dg.eig_matvec_passes
      print("n=%d lap=%d %-14s path %d fallback %d passes %3d cycles %2d basis %3d  %.1f ms (eig %.1f)  "
            "worst consumed err %.1e over %d values  k %d/%d  delta rel %.1e  ARI %.3f" % (
                n, lap, gap, dg.eig_path, dg.eig_fallback, dg.eig_matvec_passes, dg.eig_cycles,
                dg.eig_basis, 1e3 * dt, dg.stage_ms[4], err, len(idx), dg.n_clusters,
                dump["n_clusters"],
                abs(dg.max_delta - dump["max_delta"]) / max(abs(dump["max_delta"]), 1e-300),
                so.adjusted_rand_index(got, want)), flush=True)
print("total block passes %d, worst consumed-eigenvalue error %.2e" % (tot_pass, worst_all))
