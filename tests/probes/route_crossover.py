import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np
import _inputs as so
import spectralcluster_amd as sca
opts = sca.configs.icassp2018_refinement_options
for n in (1024, 1280, 1536, 1792, 2048, 2560):
  x = so.blobs(n, 256, 5, n)
  row = []
  for mode in (1, 2):
    c = sca.SpectralClusterer(min_clusters=2, max_clusters=20, refinement_options=opts, laplacian_type=sca.LaplacianType.GraphCut)
    c.diffuse_mode = mode
    for _ in range(3): c.predict(x)
    best = 1e9
    for _ in range(5):
      t = time.perf_counter()
      for _ in range(10): c.predict(x)
      best = min(best, (time.perf_counter() - t) / 10)
    row.append(1e3 * best)
  print("n=%d explicit %.3f ms  free %.3f ms" % (n, row[0], row[1]), flush=True)
