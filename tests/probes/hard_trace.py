#!/usr/bin/env python
"""Unfriendly spectra on the device: per hard golden (tests/golden/hard_*.npz) and for the
n=8192 unstructured input -- ms per call, matvec passes, restart cycles, which eigen path
finished, parity against the golden.  `SC_EIG_TRACE=1` adds the solver's own log.
   python tests/probes/hard_trace.py [max_n]"""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402
import spectralcluster_amd as sca  # noqa: E402

max_n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
opts = sca.configs.icassp2018_refinement_options
LAP = {0: None, 4: sca.LaplacianType.GraphCut}


def run(x, lap, maxc, reps=2):
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=maxc, refinement_options=opts,
                            laplacian_type=LAP[lap])
  t = time.perf_counter()
  try:
    labels = c.predict(x)
  except Exception as e:  # pylint: disable=broad-except
    return None, None, "RAISED %s: %s" % (type(e).__name__, e)
  first = 1e3 * (time.perf_counter() - t)
  t = time.perf_counter()
  for _ in range(reps):
    labels = c.predict(x)
  ms = 1e3 * (time.perf_counter() - t) / reps
  dg = c.last_diag
  return labels, dg, ("%.2f ms/call (first %.1f)  path=%d fallback=%d passes=%d basis=%d cycles=%d "
                      "k_raw=%d eig=%.2f ms" % (ms, first, dg.eig_path, dg.eig_fallback,
                                                dg.eig_matvec_passes, dg.eig_basis, dg.eig_cycles,
                                                dg.n_clusters_raw,
                                                dg.stage_times_ms().get("eig", 0.0)))


for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "hard_*.npz"))):
  g = np.load(path)
  n, d, seed, lap, maxc = (int(v) for v in g["params"])
  if n > max_n:
    continue
  kind = str(g["kind"])
  labels, dg, line = run(so.hard_inputs(kind, n, d, seed), lap, maxc)
  verdict = ""
  if dg is not None:
    w = dg.eigenvalue_array()[g["consumed_index"]]
    ref = g["consumed_eigenvalues"]
    err = np.max(np.abs(w - ref) / np.maximum(np.abs(ref), 1e-9 * np.abs(g["head_eigenvalues"]).max()))
    verdict = "  eig_err=%.1e k_ok=%d ari=%.4f" % (
        err, dg.n_clusters_raw == int(g["n_clusters_raw"]),
        so.adjusted_rand_index(labels, g["labels"]))
  print("%-28s %s%s" % (os.path.basename(path)[:-4], line, verdict), flush=True)

if max_n >= 8192:
  for kind in ("iid", "turns"):
    for lap, maxc in ((4, 20), (0, 7)):
      _, _, line = run(so.hard_inputs(kind, 8192, 256, 8192), lap, maxc, reps=2)
      print("%-28s %s" % ("hard8192_%s_lap%d" % (kind, lap), line), flush=True)
