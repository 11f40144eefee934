#!/usr/bin/env python
"""Randomised sweep over the GENERAL eigen path where block Arnoldi is the default (512 < n):
non-symmetrisable refinement sequences x every Laplacian x both eigengap rules x threshold options,
against the oracle (np.linalg.eig).  Round 6's bar: EVERY consumed eigenvalue (the far end of the
ascending NormalizedDiff rule through max_delta) within 1e-5, cluster counts and labels equal.
   python tests/probes/general_arnoldi_fuzz.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
SEQ = {
    "thr": (so.OP_ROW_WISE_THRESHOLD,),
    "crop_blur_thr": (so.OP_CROP_DIAGONAL, so.OP_GAUSSIAN_BLUR, so.OP_ROW_WISE_THRESHOLD),
    "thr_norm": (so.OP_ROW_WISE_THRESHOLD, so.OP_ROW_WISE_NORMALIZE),
    "sym_thr": (so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE, so.OP_ROW_WISE_THRESHOLD),
}
NAMES = {so.OP_CROP_DIAGONAL: "CropDiagonal", so.OP_GAUSSIAN_BLUR: "GaussianBlur",
         so.OP_ROW_WISE_THRESHOLD: "RowWiseThreshold", so.OP_SYMMETRIZE: "Symmetrize",
         so.OP_DIFFUSE: "Diffuse", so.OP_ROW_WISE_NORMALIZE: "RowWiseNormalize"}
bad = 0
worst = 0.0
t_all = time.perf_counter()
for case in range(cases):
  n = int(rng.integers(520, 2600))
  d = int(rng.choice([16, 32, 64]))
  k = int(rng.integers(2, 7))
  lap = int(rng.choice([0, 1, 2, 3, 4]))
  gap = str(rng.choice(["Ratio", "NormalizedDiff"]))
  seq = str(rng.choice(list(SEQ)))
  p = float(rng.choice([0.95, 0.9, 0.8, 0.6]))
  ttype = int(rng.choice([so.THRESHOLD_ROW_MAX, so.THRESHOLD_PERCENTILE]))
  maxc = int(rng.choice([5, 8, 12, 20]))
  noise = float(rng.choice([0.2, 0.4]))
  x = so.blobs(n, d, k, seed=7000 + case, noise=noise)
  gcode = so.EIGENGAP_RATIO if gap == "Ratio" else so.EIGENGAP_NORMALIZED_DIFF
  cfg = so.OracleConfig(min_clusters=2, max_clusters=maxc, sequence=SEQ[seq], gaussian_blur_sigma=1,
                        p_percentile=p, threshold_type=ttype, laplacian_type=lap, eigengap_type=gcode)
  dump = {}
  want = so.predict(x, cfg, dump)
  opts = sca.RefinementOptions(
      gaussian_blur_sigma=1, p_percentile=p, thresholding_type=sca.ThresholdType(ttype),
      refinement_sequence=[getattr(sca.RefinementName, NAMES[op]) for op in SEQ[seq]])
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                            laplacian_type=sca.LaplacianType(lap) if lap else None,
                            eigengap_type=getattr(sca.EigenGapType, gap))
  got = c.predict(x)
  dg = c.last_diag
  descend = lap in (0, 1)
  note = ""
  if dg.symmetry_state != 3:
    note = "(symmetrisable: not the general path)"
  w = c.consumed_eigenvalues()
  ref = dump["eigenvalues"]
  idx = [i for i in so.consumed_eigen_indices(n, maxc, descend, ref if descend else None,
                                              1e-2 if descend else None, gcode) if i < w.size]
  err = max(abs(w[i] - ref[i]) / max(abs(ref[i]), 1e-12) for i in idx)
  derr = abs(dg.max_delta - dump["max_delta"]) / max(abs(dump["max_delta"]), 1e-300)
  degenerate = dump["max_delta"] < 1e-9
  ari = so.adjusted_rand_index(got, want)
  ok = degenerate or (err <= 1e-5 and derr <= 1e-5 and dg.n_clusters == dump["n_clusters"] and ari == 1.0)
  bad += not ok
  if not degenerate:
    worst = max(worst, err, derr)
  print("case %2d n=%4d d=%2d k=%d lap=%d %-14s %-13s p=%.2f t=%d maxc=%2d: path=%d fb=%d passes=%3d "
        "err=%.1e delta_err=%.1e k %d/%d ari=%.3f %s%s" % (
            case, n, d, k, lap, gap, seq, p, ttype, maxc, dg.eig_path, dg.eig_fallback,
            dg.eig_matvec_passes, err, derr, dg.n_clusters, dump["n_clusters"], ari,
            "" if ok else "MISMATCH ", note), flush=True)
print("%d cases, %d mismatches, worst consumed-eigenvalue / max_delta error %.2e, %.0f s" % (
    cases, bad, worst, time.perf_counter() - t_all))
