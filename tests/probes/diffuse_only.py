"""Diffuse (S S^T) at n=8192 through the stage API, for rocprofv3 kernel timing.
   python tests/probes/diffuse_only.py [n] [reps] [data: random|const|sparse|fewbit]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spectralcluster_amd import refinement as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
data = sys.argv[3] if len(sys.argv) > 3 else "random"
rng = np.random.default_rng(0)
m = rng.random((n, n)); m = (m + m.T) / 2
if data == "const":
  m[:] = 0.5
elif data == "sparse":      # like a thresholded affinity: 95 % tiny values
  m = np.where(m > 0.95, m, 0.01 * m)
elif data == "fewbit":      # 4 mantissa bits
  m = np.round(m * 16) / 16
for _ in range(reps):
  out = rf.Diffuse().refine(m)
print(data, out[0, :3])
