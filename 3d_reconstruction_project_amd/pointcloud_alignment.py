"""Drop-in for the reference's pointcloud_alignment.py (class PointCloudAlignment, method align_point_clouds) and
for the GICP flavour of the same step in test/GICP1.py:81-104.  Same signature, same prints, same return value
(the DOWN-SAMPLED, transformed source); the Open3D calls are replaced by the HIP kernels behind include/r3d.h."""
import numpy as np

from . import cloud_ops
from .pointcloud import as_arrays, like


class PointCloudAlignment:
    def __init__(self, verbose=True, method="point_to_point"):
        self.verbose = verbose
        self.method = method
        self.last_result = None

    def align_point_clouds(self, source, target, threshold=0.02, voxel_size=0.01, max_iter=100):
        """pointcloud_alignment.py:6-43: voxel_down_sample both clouds, estimate_normals(Hybrid(2*voxel, 30)) on both,
        registration_icp(PointToPoint, criteria(1e-6, 1e-6, max_iter)) from identity, transform the source."""
        sp, sc, _ = as_arrays(source)
        tp, _, _ = as_arrays(target)
        if self.verbose:                                    # the reference's three progress lines, kept verbatim
            print("Downsampling point clouds using voxel size:", voxel_size)
            print("Estimating normals on CPU...")
            print("Performing ICP alignment using CUDA...")
        mode = {"point_to_point": cloud_ops.P2P, "point_to_plane": cloud_ops.P2PLANE, "gicp": cloud_ops.GICP}[self.method]
        # one device-resident call: both clouds go up once, the down-sampled transformed source comes back
        res = cloud_ops.align_point_clouds(sp, tp, threshold, voxel_size, max_iter, mode, voxel_size * 2, 30, source_colors=sc)
        self.last_result = res
        return like(source, res["points"], res["colors"], res["normals"])

    # north_star alias
    align = align_point_clouds


class GeneralizedICPAlignment:
    """test/GICP1.py:81-104 align_point_clouds(source, target, threshold=0.02): normals Hybrid(0.05, 30) if missing,
    registration_generalized_icp with default criteria (30 iterations), source.transform."""

    def __init__(self):
        self.last_result = None

    def align_point_clouds(self, source, target, threshold=0.02):
        sp, sc, sn = as_arrays(source)
        tp, _, tn = as_arrays(target)
        if sn is None:
            sn = cloud_ops.estimate_normals(sp, 0.05, 30)
        if tn is None:
            tn = cloud_ops.estimate_normals(tp, 0.05, 30)
        res = cloud_ops.registration(sp, tp, threshold, np.eye(4), cloud_ops.GICP, 30, 1e-6, 1e-6, sn, tn)
        self.last_result = res
        T = res["T"]
        return like(source, cloud_ops.transform_points(sp, T), sc, cloud_ops.transform_points(sn, T, rotate_only=True))

    align = align_point_clouds


def align(source, target, **kw):
    return PointCloudAlignment(verbose=False).align_point_clouds(source, target, **kw)


def multi_scale_icp(source, target, voxel_size, init=None, scales=(15.0, 5.0, 1.5), iterations=(30, 20, 10),
                    source_normals=None, target_normals=None, verbose=False):
    """Multi-scale point-to-plane refinement of test/check2.py:143-156 (same loop in check_lama1.py:287-290,
    mini1.py:293-296): registration_icp(PointToPlane) at max_correspondence_distance = voxel_size * scale, each scale
    starting from the previous result; ICPConvergenceCriteria(max_iteration=n) keeps the 1e-6 relative defaults.
    Returns (T, per-scale result dicts)."""
    sp, _, sn = as_arrays(source)
    tp, _, tn = as_arrays(target)
    tn = target_normals if target_normals is not None else tn
    if tn is None:
        raise ValueError("point-to-plane ICP needs target normals (the reference estimates them in preprocess_point_cloud)")
    T = np.eye(4) if init is None else np.asarray(init, dtype=np.float64)
    results = []
    for scale, n_it in zip(scales, iterations):
        dist = voxel_size * scale
        if verbose:
            print(f"ICP at scale {len(results)}, max correspondence distance: {dist}")
        res = cloud_ops.registration(sp, tp, dist, T, cloud_ops.P2PLANE, n_it, 1e-6, 1e-6, None, tn)
        T = res["T"]
        results.append(res)
    return T, results
