"""Probe: large-k k-means (k_kmeans<true>) against the oracle, by (k, n); SC_KMEANS_TRACE=1
prints the device's k-means++ seeds, compared here with the oracle's."""
import ctypes, os, sys, re, subprocess
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so
if len(sys.argv) > 1:
  import spectralcluster_amd as sca
  from spectralcluster_amd import _lib
  k, n = int(sys.argv[1]), int(sys.argv[2])
  h = _lib.default_handle()
  rng = np.random.default_rng(k)
  centers = rng.standard_normal((k, k))
  e = np.ascontiguousarray(centers[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, k)))
  labels = np.empty(n, dtype=np.int64)
  iters = ctypes.c_int(0)
  h.check(h.lib.sc_stage_kmeans_metric(h.raw, _lib.as_double_p(e), n, k, 300, 0, _lib.as_int64_p(labels),
                                       None, ctypes.byref(iters)))
  want = so.run_kmeans(e, k, 300)
  print("RESULT k=%d n=%d mismatches %d iterations %d" % (k, n, int((labels != want).sum()), iters.value))
  sys.exit(0)
for k, n in ((300, 1237), (257, 1100), (300, 1025)):
  env = dict(os.environ, SC_KMEANS_TRACE="1")
  r = subprocess.run([sys.executable, __file__, str(k), str(n)], capture_output=True, text=True, env=env)
  print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
  m = re.search(r"seeds:((?: \d+)+)", r.stderr)
  rng = np.random.default_rng(k)
  centers = rng.standard_normal((k, k))
  e = centers[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, k))
  xc = e - e.mean(axis=0)
  want = so.kmeanspp_seeds(xc, k, so.Mt19937(0))
  if m:
    got = np.array([int(v) for v in m.group(1).split()])
    diff = np.nonzero(got != want)[0]
    print("   seeds differ at", diff[:10], "of", diff.size, "; first: device", got[diff[:3]], "oracle", want[diff[:3]])
  else:
    print("   no seeds line", r.stderr[-300:])
