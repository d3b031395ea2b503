"""Drop-in for the clean-up stage of the reference's pointcloud_processing.py:23-44
(PointCloudProcessingWithCUDA.process_point_cloud): voxel_down_sample(0.0025) -> remove_statistical_outlier(30, 1.2)
-> remove_radius_outlier(16, 0.01).  The reference reads the cloud from a PLY file name; a cloud object is accepted
as well (the file path goes through io_formats.read_ply)."""
import numpy as np

from . import _lib, cloud_ops
from .pointcloud import PointCloud, as_arrays, like


class PointCloudProcessingWithCUDA:
    def __init__(self, device="CUDA:0", downsample_voxel_size=0.0025):
        self.device = _lib.parse_device(device)
        self.downsample_voxel_size = downsample_voxel_size

    def process_point_cloud(self, filename_or_cloud, nb_neighbors=30, std_ratio=1.2, nb_points=16, radius=0.01):
        if isinstance(filename_or_cloud, str):
            from .io_formats import read_ply
            d = read_ply(filename_or_cloud)
            pcd = PointCloud(d["points"], d.get("colors_f"), d.get("normals"))
        else:
            pcd = filename_or_cloud
        ctx = _lib.default_context(self.device)
        p, c, n = as_arrays(pcd)
        # from_legacy(pcd, Float32) -> o3d.t voxel_down_sample (float32 keys from origin 0, float32 means) -> to_legacy
        p, c, n = cloud_ops.voxel_down_sample(p, self.downsample_voxel_size, c, n, ctx=ctx, tensor=True)
        keep = cloud_ops.statistical_outlier_mask(p, nb_neighbors, std_ratio, ctx=ctx)
        p, c, n = p[keep], (c[keep] if c is not None else None), (n[keep] if n is not None else None)
        keep = cloud_ops.radius_outlier_mask(p, nb_points, radius, ctx=ctx)
        p, c, n = p[keep], (c[keep] if c is not None else None), (n[keep] if n is not None else None)
        return like(pcd if not isinstance(filename_or_cloud, str) else None, p, c, n)
