#!/usr/bin/env python
"""SC_EIG_TRACE=1 python tests/probes/hard8192_trace.py: the solver log of one predict() on
the unstructured n=8192 input of bench.py's `hard8192` key (after a warm-up call)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spectralcluster_amd as sca  # noqa: E402

x = np.ascontiguousarray(np.random.default_rng(8192).standard_normal((8192, 256)))
c = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                          refinement_options=sca.configs.icassp2018_refinement_options,
                          laplacian_type=sca.LaplacianType.GraphCut)
c.predict(x)
sys.stderr.write("==== second call\n")
t = time.perf_counter()
c.predict(x)
print("%.2f ms" % (1e3 * (time.perf_counter() - t)), c.last_diag.stage_times_ms())
