"""MI355X-native hot path of the stereo-depth + point-cloud fusion pipeline.

The directory name starts with a digit (it mirrors the upstream project name), so import it with
    import importlib; r3d = importlib.import_module("3d_reconstruction_project_amd")
or through the `r3d` alias module at the repository root (`import r3d`).
"""
from . import _lib, stereo_sgbm, synth  # noqa: F401
from ._lib import Context, R3DError, default_context  # noqa: F401
from .stereo_sgbm import (STEREO_SGBM_MODE_SGBM_3WAY, StereoSGBM, StereoSGBM_create, depth,  # noqa: F401
                          reference_matcher)
