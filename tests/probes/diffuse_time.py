"""Diffuse GEMM kernel time at n=8192 through the product library (HIP events inside bench's
roofline leg are per call; here: wall clock of back-to-back resident Diffuse stages)."""
import ctypes
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spectralcluster_amd import _lib
from spectralcluster_amd import refinement as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
m = rng.random((n, n)); m = (m + m.T) / 2
t0 = time.perf_counter(); out = rf.Diffuse().refine(m); t1 = time.perf_counter()
t2 = time.perf_counter(); out = rf.Diffuse().refine(m); t3 = time.perf_counter()
print("sched", os.environ.get("SC_GEMM_SCHED", "0"), "call ms", 1e3 * (t3 - t2), "finite", bool(np.isfinite(out).all()))
