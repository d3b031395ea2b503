"""Drop-in for the reference's normal_estimation.py: NormalEstimation(device).estimate_normals(pcd)
(normal_estimation.py:4-22; called from main.py:80).  The reference converts the cloud to a Float32 tensor cloud,
runs estimate_normals(max_nn=50, radius=0.05) and orient_normals_consistent_tangent_plane(100)."""
import numpy as np

from . import _lib, cloud_ops
from .pointcloud import as_arrays, like


class NormalEstimation:
    def __init__(self, device="CUDA:0"):
        self.device = _lib.parse_device(device)

    def estimate_normals(self, pcd, max_nn=50, radius=0.05, orient_k=100):
        p, c, _ = as_arrays(pcd)
        ctx = _lib.default_context(self.device)
        p32 = p.astype(np.float32).astype(np.float64)       # from_legacy(pcd, Float32): coordinates round to fp32
        n = cloud_ops.estimate_normals(p32, radius, max_nn, ctx=ctx)
        if orient_k:
            from .orientation import orient_normals_consistent_tangent_plane
            n = orient_normals_consistent_tangent_plane(p32, n, orient_k, ctx=ctx)
        return like(pcd, p32, c, n)

    estimate = estimate_normals


def estimate(pcd, device="CUDA:0", **kw):
    return NormalEstimation(device).estimate_normals(pcd, **kw)
