"""One Diffuse (S S^T) at n=8192 through the stage API, for rocprofv3 kernel timing."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectralcluster_amd import refinement as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
m = rng.random((n, n)); m = (m + m.T) / 2
for _ in range(3):
  out = rf.Diffuse().refine(m)
print(out[0, :3])
