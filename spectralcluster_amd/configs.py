"""Preset configurations (mirror of reference `spectralcluster/configs.py:21-43`).

The ICASSP 2018 ("Speaker Diarization with LSTM") preset is complete.  For the
Turn-to-Diarize system (reference configs.py:45-80) the refinement options and
the AutoTune object are provided and run on the device; its constraint
propagation (`constraint_options`) is outside the hot-path scope (SURVEY.md
section 8f-N3), so no `turntodiarize_clusterer` singleton is exported.
"""

from spectralcluster_amd import autotune

from spectralcluster_amd import refinement
from spectralcluster_amd import spectral_clusterer

RefinementName = refinement.RefinementName
RefinementOptions = refinement.RefinementOptions
ThresholdType = refinement.ThresholdType
SymmetrizeType = refinement.SymmetrizeType
SpectralClusterer = spectral_clusterer.SpectralClusterer
AutoTune = autotune.AutoTune

ICASSP2018_REFINEMENT_SEQUENCE = [
    RefinementName.CropDiagonal,
    RefinementName.GaussianBlur,
    RefinementName.RowWiseThreshold,
    RefinementName.Symmetrize,
    RefinementName.Diffuse,
    RefinementName.RowWiseNormalize,
]

TURNTODIARIZE_REFINEMENT_SEQUENCE = [
    RefinementName.RowWiseThreshold, RefinementName.Symmetrize
]

icassp2018_refinement_options = RefinementOptions(
    gaussian_blur_sigma=1,
    p_percentile=0.95,
    thresholding_soft_multiplier=0.01,
    thresholding_type=ThresholdType.RowMax,
    refinement_sequence=ICASSP2018_REFINEMENT_SEQUENCE)

icassp2018_clusterer = SpectralClusterer(
    min_clusters=2,
    max_clusters=7,
    autotune=None,
    laplacian_type=None,
    refinement_options=icassp2018_refinement_options,
    custom_dist="cosine")

turntodiarize_refinement_options = RefinementOptions(
    thresholding_soft_multiplier=0.01,
    thresholding_type=ThresholdType.Percentile,
    thresholding_with_binarization=True,
    thresholding_preserve_diagonal=True,
    symmetrize_type=SymmetrizeType.Average,
    refinement_sequence=TURNTODIARIZE_REFINEMENT_SEQUENCE)

turntodiarize_auto_tune = AutoTune(
    p_percentile_min=0.40,
    p_percentile_max=0.95,
    init_search_step=0.05,
    search_level=1)
