#!/usr/bin/env python
"""Parity experiment behind DESIGN_HISTORY.md section 3.3 "matrix-free Diffuse" (CPU only, uses the
oracle): for a refinement sequence that ENDS in Diffuse (no RowWiseNormalize after it), the
eigen-stage never needs S = A A^T itself -- only S.V = A (A^T V) and rowsum(S) = A (A^T 1) --
so the n^3 product can be skipped.  This script checks that claim against the explicit
reference-shaped computation, and shows why it does NOT carry over to the ICASSP2018 sequence
(RowWiseNormalize needs rowmax(S), which is not a matrix-vector quantity).

    python tests/probes/matrix_free_note.py [n]
"""
import dataclasses
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spectral_oracle as so  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = so.blobs(n, 64, 6, seed=n)
seq = (so.OP_CROP_DIAGONAL, so.OP_GAUSSIAN_BLUR, so.OP_ROW_WISE_THRESHOLD, so.OP_SYMMETRIZE,
       so.OP_DIFFUSE)
cfg = so.icassp2018_config(laplacian_type=so.LAPLACIAN_GRAPH_CUT, max_clusters=20)
cfg = dataclasses.replace(cfg, sequence=seq)
dump = {}
labels_ref = so.predict(x, cfg, dump)          # explicit S = A A^T, np.linalg.eig on the Laplacian
w_ref = dump["eigenvalues"]

# matrix-free: A = the matrix BEFORE Diffuse; S is never formed
a = so.refine(so.affinity(x), dataclasses.replace(cfg, sequence=seq[:-1]))
deg = a @ (a.T @ np.ones(n))                   # rowsum(S)
h = 1.0 / (np.sqrt(deg) + 1e-10)               # GraphCut: L = h (D - S) h,  Op = -L
p = -h * h * deg
op = spla.LinearOperator((n, n), dtype=np.float64,
                         matvec=lambda v: p * v.ravel() + h * (a @ (a.T @ (h * v.ravel()))),
                         matmat=lambda v: p[:, None] * v + h[:, None] * (a @ (a.T @ (h[:, None] * v))))
theta, u = spla.eigsh(op, k=21, which="LA", tol=1e-12)
order = np.argsort(-theta)
w = -theta[order]
k, _ = so.eigengap(w, 20, eigengap_type=cfg.eigengap_type, descend=False)
k = max(k, 2)
v = u[:, order][:, :k]                         # back-transform t = 1: no RowWiseNormalize fold
v = v / np.linalg.norm(v, axis=0)[None, :]
labels = so.run_kmeans(v, k, cfg.max_iter)
idx = so.consumed_eigen_indices(n, 20, False)
rel = np.abs(w[idx] - w_ref[idx]) / np.maximum(np.abs(w_ref[idx]), 1e-12)
print("n=%d  sequence ending in Diffuse + GraphCut: matrix-free vs explicit" % n)
print("  max rel. error on the consumed eigenvalues: %.2e" % rel.max())
print("  n_clusters %d vs %d, ARI %.3f" % (k, dump["n_clusters"], so.adjusted_rand_index(labels, labels_ref)))
s = a @ a.T
print("ICASSP2018 (RowWiseNormalize after Diffuse) needs rowmax(S):")
print("  rows whose maximum is NOT on the diagonal: %d of %d" % (int((s.argmax(axis=1) != np.arange(n)).sum()), n))
bound = np.sqrt(np.diag(s)) * np.sqrt(np.diag(s)).max()
print("  Cauchy-Schwarz bound / true rowmax: median %.2f, max %.2f (no usable pruning)"
      % (np.median(bound / s.max(axis=1)), (bound / s.max(axis=1)).max()))
