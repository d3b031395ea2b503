"""Minimal stand-in for open3d.geometry.PointCloud (Open3D is not installed here or on the GPU box).

The drop-in classes accept ANY object exposing .points / .normals / .colors convertible by np.asarray (a real legacy
Open3D cloud works: its Vector3dVector attributes convert and can be assigned back) or this small class.
Mirrors the members the reference uses: `+=` (main.py:49), transform (pointcloud_alignment.py:42), has_normals
(test/GICP1.py:94-97), len(pcd.points) (main.py:39,42).
"""
import numpy as np


def _arr(a):
    if a is None:
        return np.zeros((0, 3))
    a = np.asarray(a, dtype=np.float64)
    return a.reshape(-1, 3) if a.size else np.zeros((0, 3))


class PointCloud:
    def __init__(self, points=None, colors=None, normals=None):
        self.points = _arr(points)
        self.colors = _arr(colors)
        self.normals = _arr(normals)

    def has_points(self):
        return len(self.points) > 0

    def has_normals(self):
        return len(self.normals) > 0 and len(self.normals) == len(self.points)

    def has_colors(self):
        return len(self.colors) > 0 and len(self.colors) == len(self.points)

    def __len__(self):
        return len(self.points)

    def __iadd__(self, other):
        """Legacy operator+=: attributes survive only if BOTH clouds carry them (or self was empty)."""
        n0 = len(self.points)
        op, on, oc = _arr(other.points), _arr(getattr(other, "normals", None)), _arr(getattr(other, "colors", None))
        keep_n = (n0 == 0 or self.has_normals()) and len(on) == len(op) and len(op) > 0
        keep_c = (n0 == 0 or self.has_colors()) and len(oc) == len(op) and len(op) > 0
        self.normals = np.concatenate([_arr(self.normals), on]) if keep_n else np.zeros((0, 3))
        self.colors = np.concatenate([_arr(self.colors), oc]) if keep_c else np.zeros((0, 3))
        self.points = np.concatenate([_arr(self.points), op])
        return self

    def __add__(self, other):
        out = PointCloud(self.points.copy(), self.colors.copy(), self.normals.copy())
        out += other
        return out

    def transform(self, T):
        from . import cloud_ops
        T = np.asarray(T, dtype=np.float64)
        if len(self.points):
            self.points = cloud_ops.transform_points(self.points, T)
        if self.has_normals():
            self.normals = cloud_ops.transform_points(self.normals, T, rotate_only=True)
        return self

    def voxel_down_sample(self, voxel_size):
        from . import cloud_ops
        p, c, n = cloud_ops.voxel_down_sample(self.points, voxel_size, self.colors if self.has_colors() else None,
                                              self.normals if self.has_normals() else None)
        return PointCloud(p, c, n)

    def estimate_normals(self, radius=None, max_nn=30, search_param=None):
        """estimate_normals(search_param=KDTreeSearchParamHybrid(radius, max_nn)) -- pass the two numbers, or any
        object with .radius / .max_nn (or .knn) attributes."""
        from . import cloud_ops
        if search_param is not None:
            radius = getattr(search_param, "radius", None)
            max_nn = getattr(search_param, "max_nn", getattr(search_param, "knn", max_nn))
        self.normals = cloud_ops.estimate_normals(self.points, radius, max_nn,
                                                  self.normals if self.has_normals() else None)
        return self


def as_arrays(pcd):
    """(points, colors|None, normals|None) float64 arrays from a PointCloud-like object or an (N,3) array."""
    if isinstance(pcd, np.ndarray):
        return _arr(pcd), None, None
    p = _arr(pcd.points)
    c = _arr(getattr(pcd, "colors", None))
    n = _arr(getattr(pcd, "normals", None))
    return p, (c if len(c) == len(p) and len(p) else None), (n if len(n) == len(p) and len(p) else None)


def like(template, points, colors=None, normals=None):
    """Builds the result in the caller's own cloud type when it is constructible and assignable, else PointCloud."""
    if isinstance(template, (PointCloud, np.ndarray)) or template is None:
        return PointCloud(points, colors, normals)
    try:
        out = type(template)()
        vec = type(template.points)
        out.points = vec(points)
        if colors is not None:
            out.colors = vec(colors)
        if normals is not None:
            out.normals = vec(normals)
        return out
    except Exception:  # noqa: BLE001
        return PointCloud(points, colors, normals)
