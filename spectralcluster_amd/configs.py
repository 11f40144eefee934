"""Preset configurations (values of reference `spectralcluster/configs.py:21-80`).

Both presets run entirely on the device: ICASSP 2018 ("Speaker Diarization with
LSTM") and Turn-to-Diarize (Percentile threshold + constraint propagation +
AutoTune, GraphCut Laplacian, row-wise re-norm).  Each clusterer is assembled from
keyword tables so the numbers of a preset sit in one place.
"""

from spectralcluster_amd import autotune
from spectralcluster_amd import constraint
from spectralcluster_amd import laplacian
from spectralcluster_amd import refinement
from spectralcluster_amd import spectral_clusterer

AutoTune = autotune.AutoTune
ConstraintName = constraint.ConstraintName
ConstraintOptions = constraint.ConstraintOptions
LaplacianType = laplacian.LaplacianType
RefinementName = refinement.RefinementName
RefinementOptions = refinement.RefinementOptions
SpectralClusterer = spectral_clusterer.SpectralClusterer
SymmetrizeType = refinement.SymmetrizeType
ThresholdType = refinement.ThresholdType

_OPS = RefinementName

# ---- ICASSP 2018 (reference configs.py:21-43) --------------------------------------
ICASSP2018_REFINEMENT_SEQUENCE = [_OPS.CropDiagonal, _OPS.GaussianBlur, _OPS.RowWiseThreshold,
                                  _OPS.Symmetrize, _OPS.Diffuse, _OPS.RowWiseNormalize]

_ICASSP2018_REFINEMENT = dict(gaussian_blur_sigma=1, p_percentile=0.95,
                              thresholding_soft_multiplier=0.01,
                              thresholding_type=ThresholdType.RowMax)
_ICASSP2018_CLUSTERER = dict(min_clusters=2, max_clusters=7, autotune=None, laplacian_type=None,
                             custom_dist="cosine")

icassp2018_refinement_options = RefinementOptions(
    refinement_sequence=ICASSP2018_REFINEMENT_SEQUENCE, **_ICASSP2018_REFINEMENT)
icassp2018_clusterer = SpectralClusterer(
    refinement_options=icassp2018_refinement_options, **_ICASSP2018_CLUSTERER)

# ---- Turn-to-Diarize (reference configs.py:45-80) ----------------------------------
TURNTODIARIZE_REFINEMENT_SEQUENCE = [_OPS.RowWiseThreshold, _OPS.Symmetrize]

_TURNTODIARIZE_REFINEMENT = dict(thresholding_soft_multiplier=0.01,
                                 thresholding_type=ThresholdType.Percentile,
                                 thresholding_with_binarization=True,
                                 thresholding_preserve_diagonal=True,
                                 symmetrize_type=SymmetrizeType.Average)
_TURNTODIARIZE_SWEEP = dict(p_percentile_min=0.40, p_percentile_max=0.95,
                            init_search_step=0.05, search_level=1)
_TURNTODIARIZE_CONSTRAINT = dict(constraint_name=ConstraintName.ConstraintPropagation,
                                 apply_before_refinement=True,
                                 constraint_propagation_alpha=0.4)
_TURNTODIARIZE_CLUSTERER = dict(min_clusters=2, max_clusters=7,
                                laplacian_type=LaplacianType.GraphCut, row_wise_renorm=True,
                                custom_dist="cosine")

turntodiarize_refinement_options = RefinementOptions(
    refinement_sequence=TURNTODIARIZE_REFINEMENT_SEQUENCE, **_TURNTODIARIZE_REFINEMENT)
turntodiarize_constraint_options = ConstraintOptions(**_TURNTODIARIZE_CONSTRAINT)
turntodiarize_auto_tune = AutoTune(**_TURNTODIARIZE_SWEEP)
turntodiarize_clusterer = SpectralClusterer(
    refinement_options=turntodiarize_refinement_options,
    constraint_options=turntodiarize_constraint_options,
    autotune=turntodiarize_auto_tune, **_TURNTODIARIZE_CLUSTERER)
