"""GPU: the reference-side ctypes binding INTEGRATION.md section 2 shows is EXECUTED here.

The two Python code blocks of that section (what a maintainer of the reference would paste
into spectralcluster/spectral_clusterer.py to route predict() -- spectral_clusterer.py:201-314
-- through the C ABI) are extracted from the document, pointed at the in-tree library, and
run against a duck-typed object that carries the reference constructor's attributes
(/root/reference does not exist on the GPU box).  Results must be those of
spectralcluster_amd.SpectralClusterer, i.e. of the reference goldens.
"""

import os
import re
import types

import numpy as np
import pytest

import spectral_oracle as so
from conftest import ROOT, golden

import spectralcluster_amd as sca

pytestmark = pytest.mark.gpu

SO_PATH = os.path.join(ROOT, "spectralcluster_amd", "csrc", "libspectralcluster_amd.so")


def _stub_namespace():
  text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  section = text[text.index("## 2."):text.index("## 3.")]
  blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
  assert len(blocks) == 2, "INTEGRATION.md section 2 is expected to hold two code blocks"
  code = "\n".join(blocks)
  assert 'ctypes.CDLL("libspectralcluster_amd.so")' in code
  code = code.replace('ctypes.CDLL("libspectralcluster_amd.so")', "ctypes.CDLL(%r)" % SO_PATH)
  ns = {}
  exec(compile(code, "INTEGRATION.md#2", "exec"), ns)  # pylint: disable=exec-used
  return ns


def _duck(ns, **kw):
  """An object with the attributes SpectralClusterer.__init__ stores (spectral_clusterer.py:
  86-106) and the stub's functions bound as methods -- what the patched reference class is."""
  ref = sca.SpectralClusterer(**kw)  # only used as an attribute bag with the same enums
  obj = types.SimpleNamespace(**{k: getattr(ref, k) for k in (
      "min_clusters", "max_clusters", "refinement_options", "autotune", "laplacian_type",
      "stop_eigenvalue", "row_wise_renorm", "custom_dist", "max_iter", "constraint_options",
      "eigengap_type")})
  for name in ("predict", "predict_many", "evaluate_level", "cluster_winner", "_to_config"):
    setattr(obj, name, types.MethodType(ns[name], obj))
  return obj


@pytest.fixture(scope="module")
def stub():
  return _stub_namespace()


@pytest.mark.parametrize("name", ["e2e_n1000_lap0_max7", "e2e_n1000_lap4_max20",
                                  "e2e_n2048_lap4_max20"])
def test_stub_predict_matches_reference_golden(stub, name):
  g = golden(name + ".npz")
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  kw = dict(min_clusters=2, max_clusters=maxc,
            refinement_options=sca.configs.icassp2018_refinement_options,
            laplacian_type={0: None, 4: sca.LaplacianType.GraphCut}[lap])
  labels = _duck(stub, **kw).predict(x)
  assert labels.dtype == np.int64
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  assert np.array_equal(labels, sca.SpectralClusterer(**kw).predict(x))


def test_stub_turntodiarize_options_and_constraints(stub):
  """Every RefinementOptions field and the constraint options travel through the stub's
  _to_config (Percentile + binarisation + preserved diagonal + Average; E2CP)."""
  g = golden("turntodiarize_n300.npz")
  x, _, scores = so.turn_blobs(int(g["n"]), int(g["d"]), int(g["k"]), int(g["seed"]))
  q = so.constraint_matrix_diagonals(scores)
  ttd = sca.configs.turntodiarize_clusterer
  opts = sca.RefinementOptions(
      p_percentile=0.9, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.Percentile, thresholding_with_binarization=True,
      thresholding_preserve_diagonal=True, symmetrize_type=sca.SymmetrizeType.Average,
      refinement_sequence=sca.TURNTODIARIZE_REFINEMENT_SEQUENCE)
  kw = dict(min_clusters=2, max_clusters=7, refinement_options=opts,
            constraint_options=ttd.constraint_options,
            laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  got = _duck(stub, **kw).predict(x, q)
  want = sca.SpectralClusterer(**kw).predict(x, q)
  assert np.array_equal(got, want)


def test_stub_batch_and_autotune_level(stub):
  xs = [so.blobs(n, 64, 3, seed=n) for n in (600, 640, 700, 900)]
  kw = dict(min_clusters=2, max_clusters=7,
            refinement_options=sca.configs.icassp2018_refinement_options)
  duck = _duck(stub, **kw)
  many = duck.predict_many(xs)
  ref = sca.SpectralClusterer(**kw)
  for a, x in zip(many, xs):
    assert so.adjusted_rand_index(a, ref.predict(x)) == 1.0
  # one AutoTune level through the stub: the affinity is made resident first
  g = golden("autotune_n512.npz")
  x = so.blobs(512, 64, 6, 512)
  kw = dict(min_clusters=2, max_clusters=20,
            refinement_options=sca.RefinementOptions(
                gaussian_blur_sigma=1, p_percentile=0.95, thresholding_soft_multiplier=0.01,
                refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE),
            laplacian_type=sca.LaplacianType.GraphCut)
  duck = _duck(stub, **kw)
  import ctypes
  lib, handle = stub["_lib"], stub["_handle"]
  xc = np.ascontiguousarray(x)
  assert lib.sc_set_embeddings(handle, xc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               512, 64) == 0
  assert lib.sc_compute_affinity(handle) == 0
  grid = [float(p) for p in g["grid"]]
  out = duck.evaluate_level(grid)
  np.testing.assert_allclose([r for r, _ in out], g["ratios"], rtol=1e-6)
  assert [k for _, k in out] == [int(v) for v in g["n_clusters"]]
  # the search's winner: eigenvectors adopted from the sweep, then k-means
  winner = int(np.argmin([r for r, _ in out]))
  labels = duck.cluster_winner(grid, winner, 512)
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
