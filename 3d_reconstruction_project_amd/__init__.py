"""MI355X-native hot path of the stereo-depth + point-cloud fusion pipeline.

The directory name starts with a digit (it mirrors the upstream project name), so import it with
    import importlib; r3d = importlib.import_module("3d_reconstruction_project_amd")
or through the `r3d` alias module at the repository root (`import r3d`).
"""
from . import (_lib, cloud_ops, distributed, io_formats, pipeline, normal_estimation, orientation, pointcloud, pointcloud_alignment,  # noqa: F401
               pointcloud_processing, stereo_prepost, stereo_sgbm, synth)
from ._lib import Context, R3DError, default_context  # noqa: F401
from .stereo_sgbm import (STEREO_SGBM_MODE_SGBM_3WAY, StereoSGBM, StereoSGBM_create, createRightMatcher, depth,  # noqa: F401
                          filterSpeckles, reference_matcher)
from .normal_estimation import NormalEstimation, estimate  # noqa: F401,E402
from .pointcloud import PointCloud  # noqa: F401,E402
from .pointcloud_alignment import GeneralizedICPAlignment, PointCloudAlignment, align, multi_scale_icp  # noqa: F401,E402
from .pointcloud_processing import PointCloudProcessingWithCUDA  # noqa: F401,E402
from .stereo_prepost import (COLOR_BGR2GRAY, CV_16SC2, INTER_LINEAR, NORM_MINMAX, DisparityWLSFilter, createDisparityWLSFilter,  # noqa: F401,E402
                             cvtColor, initUndistortRectifyMap, normalize, remap)
