"""How the upper-triangle block matvec of a single call scales with the number of tiles in flight:
block Lanczos (stage API) on random symmetric matrices of n = 8192 / 12288 / 16384 under
rocprofv3 --kernel-trace --stats; bytes = upper-triangle tiles x 128 KB per launch.
  rocprofv3 --kernel-trace --stats -d out -o run -- python tests/probes/matvec_scale_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralcluster_amd as sca  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [8192, 12288, 16384]:
  rng = np.random.default_rng(n)
  x = rng.standard_normal((n, 64))
  m = x @ x.T / 64.0
  m = 0.5 * (m + m.T)
  w, _ = sca.utils.compute_sorted_eigenvectors(m, descend=True, count=8)
  nt = (n + 127) // 128
  print("n=%d tiles=%d bytes per launch %.1f MB; top eigenvalue %.6g" % (
      n, nt * (nt + 1) // 2, nt * (nt + 1) // 2 * 128 * 128 * 8 / 1e6, w[0]), flush=True)
