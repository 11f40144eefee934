"""Preset configurations (mirror of reference `spectralcluster/configs.py:21-43`).

Only the ICASSP 2018 ("Speaker Diarization with LSTM") preset is on the device
hot path.  The Turn-to-Diarize preset needs percentile thresholding and
constraint propagation, which are "next" rows of SURVEY.md section 8(f).
"""

from spectralcluster_amd import refinement
from spectralcluster_amd import spectral_clusterer

RefinementName = refinement.RefinementName
RefinementOptions = refinement.RefinementOptions
ThresholdType = refinement.ThresholdType
SymmetrizeType = refinement.SymmetrizeType
SpectralClusterer = spectral_clusterer.SpectralClusterer

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
