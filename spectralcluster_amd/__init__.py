"""MI355X-native dense hot path of SpectralClusterer.predict().

Public names mirror the reference package (`spectralcluster/__init__.py:14-43`)
for everything that touches the hot path.
"""

from spectralcluster_amd import _lib
from spectralcluster_amd import autotune
from spectralcluster_amd import configs
from spectralcluster_amd import constraint
from spectralcluster_amd import custom_distance_kmeans
from spectralcluster_amd import fallback_clusterer
from spectralcluster_amd import laplacian
from spectralcluster_amd import multi_stage_clusterer
from spectralcluster_amd import naive_clusterer
from spectralcluster_amd import refinement
from spectralcluster_amd import spectral_clusterer
from spectralcluster_amd import utils

AutoTune = autotune.AutoTune
AutoTuneProxy = autotune.AutoTuneProxy
ConstraintOptions = constraint.ConstraintOptions
ConstraintName = constraint.ConstraintName
IntegrationType = constraint.IntegrationType
ConstraintMatrix = constraint.ConstraintMatrix
FallbackOptions = fallback_clusterer.FallbackOptions
SingleClusterCondition = fallback_clusterer.SingleClusterCondition
FallbackClustererType = fallback_clusterer.FallbackClustererType
MultiStageClusterer = multi_stage_clusterer.MultiStageClusterer
Deflicker = multi_stage_clusterer.Deflicker
LaplacianType = laplacian.LaplacianType
RefinementName = refinement.RefinementName
RefinementOptions = refinement.RefinementOptions
ThresholdType = refinement.ThresholdType
SymmetrizeType = refinement.SymmetrizeType
SpectralClusterer = spectral_clusterer.SpectralClusterer
EigenGapType = utils.EigenGapType
ICASSP2018_REFINEMENT_SEQUENCE = configs.ICASSP2018_REFINEMENT_SEQUENCE
TURNTODIARIZE_REFINEMENT_SEQUENCE = configs.TURNTODIARIZE_REFINEMENT_SEQUENCE

DeviceLibraryError = _lib.DeviceLibraryError
UnsupportedOnDeviceError = _lib.UnsupportedOnDeviceError
EigenSolverNotConverged = _lib.EigenSolverNotConverged

__version__ = "0.1.0"
