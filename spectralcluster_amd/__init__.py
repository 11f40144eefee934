"""MI355X-native dense hot path of SpectralClusterer.predict().

The public names are those of the reference package (`spectralcluster/__init__.py:14-43`):
every submodule plus the classes / enums / presets it lifts to the top level, and the three
exception types of the device boundary.
"""

import importlib as _importlib

# submodule -> names lifted to the package level
_PUBLIC = {
    "autotune": ("AutoTune", "AutoTuneProxy"),
    "configs": ("ICASSP2018_REFINEMENT_SEQUENCE", "TURNTODIARIZE_REFINEMENT_SEQUENCE"),
    "constraint": ("ConstraintOptions", "ConstraintName", "ConstraintMatrix", "IntegrationType"),
    "custom_distance_kmeans": (),
    "fallback_clusterer": ("FallbackOptions", "SingleClusterCondition",
                           "FallbackClustererType"),
    "laplacian": ("LaplacianType",),
    "multi_stage_clusterer": ("Deflicker", "MultiStageClusterer"),
    "naive_clusterer": ("NaiveClusterer",),
    "refinement": ("RefinementName", "RefinementOptions", "ThresholdType", "SymmetrizeType"),
    "spectral_clusterer": ("SpectralClusterer",),
    "utils": ("EigenGapType",),
    "_lib": ("DeviceLibraryError", "UnsupportedOnDeviceError", "EigenSolverNotConverged"),
}

__all__ = []
for _module_name, _lifted in _PUBLIC.items():
  _module = _importlib.import_module(__name__ + "." + _module_name)
  globals()[_module_name] = _module
  if not _module_name.startswith("_"):
    __all__.append(_module_name)
  for _name in _lifted:
    globals()[_name] = getattr(_module, _name)
    __all__.append(_name)
del _module_name, _lifted, _module, _name

__version__ = "0.1.0"
