"""GPU: BASELINE.json configs 4 and 5 at their FULL sizes against goldens produced by the
real reference (oracle/make_golden.py --autotune4096 / --batch512).

  config 4  AutoTune sweep of 16 p_percentile values, n=4096 d=256, ICASSP2018 + GraphCut,
            max_clusters=20 (reference autotune.py:76-132, spectral_clusterer.py:266-289)
  config 4b the same sweep under the Turn-to-Diarize refinement (interior minimum)
  config 5  512 independent utterances, n in [300, 3000], d=256, configs.icassp2018_clusterer
"""

import os

import numpy as np
import pytest

import spectral_oracle as so
import spectralcluster_amd as sca
from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu


def icassp_options(sigma=1, p=0.95):
  return sca.RefinementOptions(
      gaussian_blur_sigma=sigma, p_percentile=p, thresholding_soft_multiplier=0.01,
      thresholding_type=sca.ThresholdType.RowMax,
      refinement_sequence=sca.ICASSP2018_REFINEMENT_SEQUENCE)


def test_config4_autotune_n4096_vs_reference():
  g = golden("autotune_n4096.npz")
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  assert (n, d, lap, maxc) == (4096, 256, 4, 20)
  x = so.blobs(n, d, k, seed)
  tuner = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                       init_search_step=0.025, search_level=1)
  grid = np.array(tuner.get_percentile_range())
  assert np.array_equal(grid, g["grid"]) and len(grid) == 16
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc, refinement_options=icassp_options(),
      autotune=tuner, laplacian_type=sca.LaplacianType.GraphCut)
  handle = clusterer._handle()
  clusterer._upload(handle, x)
  idx = g["consumed_index"]
  for i, p in enumerate(grid):
    diag = clusterer._eig_resident(handle, p)
    ratio = tuner.ratio(p, diag.max_delta)
    np.testing.assert_allclose(ratio, g["ratios"][i], rtol=1e-6, err_msg="p=%g" % p)
    assert diag.n_clusters_raw == g["n_clusters"][i], p
    w = diag.eigenvalue_array()[idx]
    ref = g["consumed_eigenvalues"][i]
    assert np.max(np.abs(w - ref) / np.maximum(np.abs(ref), 1e-12)) < 1e-5, p
  labels = clusterer.predict(x)
  # the closure leaves p_percentile at the last evaluated value (spectral_clusterer.py:277);
  # the winner is what the labels come from
  # (the golden's `best_p` is that final state of the reference)
  assert clusterer.refinement_options.p_percentile == float(g["best_p"]) == grid[-1]
  assert clusterer.last_best_p == grid[int(np.argmin(g["ratios"]))]
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0
  # the sharded form of the same sweep (what bench.py --gpus N runs), world of one
  from spectralcluster_amd import multigpu
  clusterer.autotune = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                                    init_search_step=0.025, search_level=1)
  sharded = multigpu.predict_autotune_distributed(multigpu.LocalComm(), clusterer, x)
  assert np.array_equal(sharded, labels)


def test_config4_turntodiarize_variant_n4096_vs_reference():
  """SURVEY.md 8(d) config 4, secondary variant: the same 16-value AutoTune at n=4096 under
  the Turn-to-Diarize refinement (Percentile + binarisation + preserved diagonal + Average,
  reference configs.py:49-59).  Unlike the ICASSP2018 sweep its proxy has an INTERIOR
  minimum (index 13 of 16), so first-strict-minimum selection (autotune.py:106-111) is
  distinguishable from `the last grid point` here."""
  g = golden("autotune_ttd_n4096.npz")
  n, d, k, seed, lap, maxc = (int(v) for v in g["params"])
  x = so.blobs(n, d, k, seed)
  tuner = sca.AutoTune(p_percentile_min=0.55, p_percentile_max=0.95,
                       init_search_step=0.025, search_level=1)
  grid = np.array(tuner.get_percentile_range())
  assert np.array_equal(grid, g["grid"])
  best_index = int(np.argmin(g["ratios"]))
  assert 0 < best_index < len(grid) - 1  # interior
  clusterer = sca.SpectralClusterer(
      min_clusters=2, max_clusters=maxc, autotune=tuner,
      refinement_options=sca.RefinementOptions(
          p_percentile=0.95, thresholding_soft_multiplier=0.01,
          thresholding_type=sca.ThresholdType.Percentile,
          thresholding_with_binarization=True, thresholding_preserve_diagonal=True,
          symmetrize_type=sca.SymmetrizeType.Average,
          refinement_sequence=sca.TURNTODIARIZE_REFINEMENT_SEQUENCE),
      laplacian_type=sca.LaplacianType.GraphCut, row_wise_renorm=True)
  handle = clusterer._handle()
  clusterer._upload(handle, x)
  idx = g["consumed_index"]
  for i, p in enumerate(grid):
    diag = clusterer._eig_resident(handle, p)
    np.testing.assert_allclose(tuner.ratio(p, diag.max_delta), g["ratios"][i], rtol=1e-6,
                               err_msg="p=%g" % p)
    assert diag.n_clusters_raw == g["n_clusters"][i], p
    w = diag.eigenvalue_array()[idx]
    ref = g["consumed_eigenvalues"][i]
    assert np.max(np.abs(w - ref) / np.maximum(np.abs(ref), 1e-9)) < 1e-5, p
  # the whole level as one grouped sweep reports the same
  sweep = clusterer._eig_sweep(handle, [float(p) for p in grid])
  np.testing.assert_allclose([tuner.ratio(p, dg.max_delta) for p, dg in zip(grid, sweep)],
                             g["ratios"], rtol=1e-6)
  labels = clusterer.predict(x)
  assert clusterer.last_best_p == float(g["best_p"]) == grid[best_index]
  assert clusterer.refinement_options.p_percentile == float(g["final_p"]) == grid[-1]
  assert so.adjusted_rand_index(labels, g["labels"]) == 1.0


def test_config5_batch512_vs_reference():
  g = golden("batch512.npz")
  ns, ks = g["ns"], g["ks"]
  rng = np.random.default_rng(512)
  assert np.array_equal(ns, rng.integers(300, 3001, 512))
  assert np.array_equal(ks, rng.integers(2, 8, 512))
  utts = [so.blobs(int(n), 256, int(k), seed=i) for i, (n, k) in enumerate(zip(ns, ks))]
  clusterer = sca.configs.icassp2018_clusterer
  labels = clusterer.predict_batch(utts, streams=8)
  diags = clusterer.last_batch_diags
  ref_labels, pos, bad = g["labels"], 0, []
  worst = 0.0
  for i, n in enumerate(ns):
    n = int(n)
    ref = ref_labels[pos:pos + n]
    pos += n
    d = diags[i]
    w_ref = g["consumed_eigenvalues"][i]
    idx = so.consumed_eigen_indices(n, 7, True, w_ref, 1e-2)
    w = d.eigenvalue_array()[idx]
    worst = max(worst, float(np.max(np.abs(w - w_ref[idx]) / np.maximum(np.abs(w_ref[idx]), 1e-12))))
    if (so.adjusted_rand_index(labels[i], ref) != 1.0 or
        d.n_clusters_raw != g["n_clusters_raw"][i] or
        abs(d.max_delta - g["max_delta"][i]) > 1e-6 * abs(g["max_delta"][i])):
      bad.append(i)
  assert not bad, bad
  assert worst < 1e-5, worst
  # one stream, one handle: identical labels (the batch is order-independent)
  again = clusterer.predict_batch(utts[:40], streams=1)
  for a, b in zip(again, labels[:40]):
    assert np.array_equal(a, b)
