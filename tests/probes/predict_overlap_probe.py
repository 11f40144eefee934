"""Host-side timing of sc_set_embeddings / sc_predict / sc_run_resident at n = 8192 (is the upload
hidden behind the affinity product?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import _inputs as so
import spectralcluster_amd as sca
from spectralcluster_amd import _lib
n, d = 8192, 256
x = so.blobs(n, d, 8, seed=1)
c = sca.SpectralClusterer(min_clusters=2, max_clusters=20, refinement_options=sca.configs.icassp2018_refinement_options, laplacian_type=sca.LaplacianType.GraphCut)
h = c._handle()
cfg = c.build_config() if hasattr(c, "build_config") else None
lab = np.empty(n, dtype=np.int64)
diag = _lib.ScDiag()
lib = h.lib
def t(fn, reps=20):
  fn(); lib.sc_synchronize(h.raw)
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  lib.sc_synchronize(h.raw)
  return 1e3 * (time.perf_counter() - t0) / reps
xp = _lib.as_double_p(x)
print("set_embeddings        %.3f ms" % t(lambda: h.check(lib.sc_set_embeddings(h.raw, xp, n, d))))
print("run_resident          %.3f ms" % t(lambda: h.check(lib.sc_run_resident(h.raw, cfg, _lib.as_int64_p(lab), diag))))
print("predict               %.3f ms" % t(lambda: h.check(lib.sc_predict(h.raw, xp, n, d, cfg, _lib.as_int64_p(lab), diag))))
print("compute_affinity      %.3f ms" % t(lambda: h.check(lib.sc_compute_affinity(h.raw))))
