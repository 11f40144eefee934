"""Accuracy of the dense general eigensolver on graded matrices (residuals vs numpy)."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import spectralcluster_amd as sca

rng = np.random.default_rng(0)
for n in (24, 40, 64):
  for noise in (0.0, 0.3):
    d = np.concatenate([[51.6, 49.2, 48.9], np.sort(rng.random(n - 3) * 19)[::-1]])
    u, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = u @ np.diag(d) @ u.T + noise * rng.standard_normal((n, n))
    a[0, 1] += 1e-3
    w, v = sca.utils.compute_sorted_eigenvectors(a)
    wr = np.sort(np.linalg.eigvals(a).real)[::-1]
    res = [np.linalg.norm(a @ v[:, i] - w[i] * v[:, i]) for i in range(4)]
    print("n=%d noise=%.1f  eigval err %.2e   residuals of first 4 (real parts only): %s" % (
        n, noise, np.abs(w - wr).max(), " ".join("%.1e" % r for r in res)))
