"""Timings of the N4 callers on the device (AHC, centroids, max_spectral_size, streaming)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import spectral_oracle as so
import spectralcluster_amd as sca
from spectralcluster_amd import utils

def timed(fn, reps=3):
  fn()
  t = time.time()
  for _ in range(reps):
    out = fn()
  return (time.time() - t) / reps * 1e3, out

for n, d, k in ((1000, 64, 100), (4000, 128, 500), (8192, 256, 1000)):
  x = so.blobs(n, d, 8, seed=n)
  ms, lab = timed(lambda: utils.cosine_agglomerative_clustering(x, n_clusters=k))
  t0 = time.time(); ref = so.agglomerative(x, k, "complete"); cpu = (time.time() - t0) * 1e3
  print("AHC n=%d -> %d: device %.1f ms, sklearn %.1f ms, equal %s" % (n, k, ms, cpu, np.array_equal(lab, ref)), flush=True)
  ms2, _ = timed(lambda: utils.get_cluster_centroids(x, lab))
  print("   centroids %.2f ms" % ms2)
  c = sca.SpectralClusterer(min_clusters=2, max_clusters=20,
                            refinement_options=sca.configs.icassp2018_refinement_options,
                            max_spectral_size=k)
  ms3, labels = timed(lambda: c.predict(x))
  print("   predict(max_spectral_size=%d) %.1f ms, ARI vs truth-free oracle skipped" % (k, ms3), flush=True)
