"""numpy-facing wrappers of the point-cloud entry points of the C ABI (include/r3d.h).  No CPU fallback."""
import ctypes

import numpy as np

from . import _lib

_vp = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)

_lib.register({
    "r3d_voxel_downsample": ([_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_double, _vp, _vp, _vp,
                              ctypes.POINTER(ctypes.c_int64)], ctypes.c_int),
    "r3d_voxel_downsample_tensor": ([_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_double, _vp, _vp, _vp,
                                     ctypes.POINTER(ctypes.c_int64)], ctypes.c_int),
    "r3d_estimate_normals": ([_vp, _vp, ctypes.c_int64, ctypes.c_double, ctypes.c_int32, _vp, _vp], ctypes.c_int),
    "r3d_neighbor_score": ([_vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, _vp], ctypes.c_int),
    "r3d_reproject_disparity": ([_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp,
                                 ctypes.POINTER(ctypes.c_int64)], ctypes.c_int),
    "r3d_disparity_to_cloud_dev": ([_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, ctypes.c_double, _vp,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int64, _vp, _vp,
                                    ctypes.POINTER(ctypes.c_int64)], ctypes.c_int),
    "r3d_knn_graph": ([_vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, _vp, _vp], ctypes.c_int),
    "r3d_orient_normals": ([_vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp], ctypes.c_int),
    "r3d_orient_normals_graph": ([_vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp, ctypes.c_int64, _vp], ctypes.c_int),
    "r3d_transform_points": ([_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int32, _vp], ctypes.c_int),
    "r3d_icp": ([_vp, ctypes.POINTER(_lib.IcpParams), _vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp,
                 ctypes.POINTER(_lib.IcpStats)], ctypes.c_int),
})

P2P, P2PLANE, GICP = 0, 1, 2

_lib.register({
    "r3d_debug_sort_by_cell": ([_vp, _vp, ctypes.c_int64, _vp, ctypes.c_double, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp], ctypes.c_int),
    "r3d_debug_exclusive_scan": ([_vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp], ctypes.c_int),
})


def debug_exclusive_scan(values, maximum=False, ctx=None):
    """r3d_debug_exclusive_scan: exclusive prefix sum (or running maximum) of int32 values on the device."""
    ctx = ctx or _lib.default_context()
    v = np.ascontiguousarray(values, dtype=np.int32)
    out = np.empty_like(v)
    ctx.call("r3d_debug_exclusive_scan", v.ctypes.data_as(_vp), len(v), int(bool(maximum)), out.ctypes.data_as(_vp))
    return out


def debug_sort_by_cell(points, origin, cell, dims, key_order, impl=0, ctx=None):
    """r3d_debug_sort_by_cell: (idx, keys) of the stable (cell key, index) sort; impl 0 = counting sort, 1 = radix fallback."""
    ctx = ctx or _lib.default_context()
    p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    org = np.ascontiguousarray(origin, dtype=np.float64)
    d = np.ascontiguousarray(dims, dtype=np.int32)
    idx = np.empty(len(p), np.int32)
    keys = np.empty(len(p), np.uint64)
    ctx.call("r3d_debug_sort_by_cell", p.ctypes.data_as(_vp), len(p), org.ctypes.data_as(_vp), float(cell), d.ctypes.data_as(_vp), int(key_order),
             int(impl), idx.ctypes.data_as(_vp), keys.ctypes.data_as(_vp))
    return idx, keys


class AlignParams(ctypes.Structure):
    _fields_ = [("icp", _lib.IcpParams), ("voxel_size", ctypes.c_double), ("normal_radius", ctypes.c_double),
                ("normal_max_nn", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib.register({
    "r3d_align_point_clouds": ([_vp, ctypes.POINTER(AlignParams), _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp, _vp,
                                _vp, ctypes.POINTER(ctypes.c_int64), _vp, ctypes.POINTER(_lib.IcpStats)], ctypes.c_int),
})


_i64p = ctypes.POINTER(ctypes.c_int64)


class DepthCamera(ctypes.Structure):
    """r3d_depth_camera: pinhole intrinsics (camera_intrinsic.json) + the depth conversion of create_from_rgbd_image."""
    _fields_ = [("fx", ctypes.c_double), ("fy", ctypes.c_double), ("ppx", ctypes.c_double), ("ppy", ctypes.c_double),
                ("depth_scale", ctypes.c_double), ("depth_trunc", ctypes.c_double), ("flip_yz", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


# test/check84.py:158 passes depth_scale = 1.0 / sensor depth scale, the sensor value being the float32 0.001
DEPTH_SCALE_REALSENSE = float(np.float32(1.0) / np.float32(0.001))


def depth_camera(intrinsics, depth_scale=DEPTH_SCALE_REALSENSE, depth_trunc=3.0, flip=True):
    """intrinsics: dict with fx / fy / ppx / ppy (the reference's camera_intrinsic.json) or a 3x3 matrix."""
    if isinstance(intrinsics, dict):
        fx, fy, ppx, ppy = (float(intrinsics[k]) for k in ("fx", "fy", "ppx", "ppy"))
    else:
        K = np.asarray(intrinsics, dtype=np.float64).reshape(3, 3)
        fx, fy, ppx, ppy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    return DepthCamera(fx, fy, ppx, ppy, float(depth_scale), float(depth_trunc), int(bool(flip)), 0)


def _depth_args(depth, color):
    d = np.ascontiguousarray(depth, dtype=np.uint16)
    if d.ndim != 2:
        raise ValueError("depth image must be 2-D uint16")
    c = None
    if color is not None:
        c = np.ascontiguousarray(color, dtype=np.uint8)
        if c.shape != d.shape + (3,):
            raise ValueError(f"colour image {c.shape} does not match the depth image {d.shape} (3 channels expected)")
    return d, c


_lib.register({
    "r3d_backproject_depth": ([_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(DepthCamera), _vp, ctypes.c_int32,
                               _vp, _vp, _vp, _i64p], ctypes.c_int),
    "r3d_model_append_depth": ([_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(DepthCamera), _vp, ctypes.c_int32,
                                _i64p], ctypes.c_int),
    "r3d_model_align_append_depth": ([_vp, ctypes.POINTER(AlignParams), _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.POINTER(DepthCamera), _vp, ctypes.c_int32, _vp, ctypes.POINTER(_lib.IcpStats), _i64p, _i64p],
                                     ctypes.c_int),
    "r3d_model_create": ([_vp, ctypes.POINTER(_vp)], ctypes.c_int),
    "r3d_model_destroy": ([_vp], None),
    "r3d_model_clear": ([_vp], ctypes.c_int),
    "r3d_model_size": ([_vp, _i64p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)], ctypes.c_int),
    "r3d_model_voxel_table_stats": ([_vp, _i64p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)], ctypes.c_int),
    "r3d_model_append": ([_vp, _vp, _vp, _vp, ctypes.c_int64], ctypes.c_int),
    "r3d_model_align_append": ([_vp, ctypes.POINTER(AlignParams), _vp, _vp, ctypes.c_int64, _vp, ctypes.POINTER(_lib.IcpStats), _i64p],
                               ctypes.c_int),
    "r3d_model_register_append": ([_vp, ctypes.POINTER(_lib.IcpParams), _vp, _vp, _vp, ctypes.c_int64, _vp,
                                   ctypes.POINTER(_lib.IcpStats)], ctypes.c_int),
    "r3d_model_estimate_normals": ([_vp, ctypes.c_double, ctypes.c_int32], ctypes.c_int),
    "r3d_model_download": ([_vp, _vp, _vp, _vp], ctypes.c_int),
    "r3d_disparity_to_cloud_resident": ([_vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, ctypes.c_double, _vp,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int64, _vp, _vp, _i64p], ctypes.c_int),
    "r3d_icp_dev": ([_vp, ctypes.POINTER(_lib.IcpParams), _vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp,
                     ctypes.POINTER(_lib.IcpStats)], ctypes.c_int),
    "r3d_transform_points_dev": ([_vp, _vp, ctypes.c_int64, _vp, ctypes.c_int32, _vp], ctypes.c_int),
    "r3d_transform_blocks_dev": ([_vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp], ctypes.c_int),
})


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def voxel_down_sample(points, voxel, colors=None, normals=None, ctx=None, tensor=False):
    """Legacy voxel_down_sample (grid origin min_bound - voxel/2, float64 means); tensor=True: the o3d.t method on a Float32
    cloud (origin 0, float32 keys and means; pointcloud_processing.py:27).  Voxels come out in lexicographic key order."""
    ctx = ctx or _lib.default_context()
    p, c, n = _c(points), _c(colors), _c(normals)
    N = len(p)
    if N == 0:
        return p, c, n
    op = np.empty((N, 3))
    oc = np.empty((N, 3)) if c is not None else None
    on = np.empty((N, 3)) if n is not None else None
    m = ctypes.c_int64()
    ctx.call("r3d_voxel_downsample_tensor" if tensor else "r3d_voxel_downsample", _ptr(p), _ptr(c), _ptr(n), N, float(voxel),
             _ptr(op), _ptr(oc), _ptr(on), ctypes.byref(m))
    m = m.value
    return op[:m].copy(), (oc[:m].copy() if oc is not None else None), (on[:m].copy() if on is not None else None)


def estimate_normals(points, radius, max_nn, prev_normals=None, ctx=None):
    """radius None or <= 0 -> pure kNN (KDTreeSearchParamKNN)."""
    ctx = ctx or _lib.default_context()
    p, pn = _c(points), _c(prev_normals)
    out = np.empty_like(p)
    if len(p) == 0:                      # Open3D returns the empty cloud unchanged
        return out
    ctx.call("r3d_estimate_normals", _ptr(p), len(p), float(radius) if radius else -1.0, int(max_nn), _ptr(pn), _ptr(out))
    return out


def neighbor_score(points, k=0, count_radius=0.0, ctx=None):
    ctx = ctx or _lib.default_context()
    p = _c(points)
    out = np.empty(len(p))
    ctx.call("r3d_neighbor_score", _ptr(p), len(p), int(k), float(count_radius), out.ctypes.data_as(_vp))
    return out


def reproject_disparity(disp, Q, min_disparity=0, want_pixels=False, ctx=None):
    """disp: int16 [H,W] (x16, as StereoSGBM.compute returns it); Q: 4x4.  Returns points [M,3] (and pixel indices)."""
    ctx = ctx or _lib.default_context()
    d = np.ascontiguousarray(disp, dtype=np.int16)
    H, W = d.shape
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4, 4)
    out = np.empty((H * W, 3))
    pix = np.empty(H * W, np.int32) if want_pixels else None
    m = ctypes.c_int64()
    ctx.call("r3d_reproject_disparity", d.ctypes.data_as(_vp), W, H, _ptr(Q), int(min_disparity) * 16, _ptr(out),
             pix.ctypes.data_as(_vp) if want_pixels else None, ctypes.byref(m))
    out = out[:m.value].copy()
    return (out, pix[:m.value].copy()) if want_pixels else out


def disparity_to_cloud_device(d_disp, width, height, Q, min_disparity=0, max_depth=None, pose=None, voxel=0.01,
                              normal_radius=None, max_nn=30, capacity=None, ctx=None):
    """Device-resident disparity -> cloud chain (r3d_disparity_to_cloud_dev): d_disp is the device pointer (int) of an
    int16 [height,width] map as written by StereoSGBM.compute_device.  Returns (points [M,3], normals [M,3] or None)."""
    ctx = ctx or _lib.default_context()
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4, 4)
    P = None if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(4, 4)
    cap = int(capacity) if capacity is not None else int(width) * int(height)
    pts = np.empty((cap, 3))
    nrm = np.empty((cap, 3)) if max_nn and max_nn > 0 else None
    m = ctypes.c_int64()
    ctx.call("r3d_disparity_to_cloud_dev", _vp(d_disp), int(width), int(height), _ptr(Q), int(min_disparity) * 16,
             float(max_depth) if max_depth else -1.0, _ptr(P), float(voxel) if voxel else -1.0,
             float(normal_radius) if normal_radius else -1.0, int(max_nn) if max_nn else 0, cap, _ptr(pts), _ptr(nrm),
             ctypes.byref(m))
    return pts[:m.value].copy(), (nrm[:m.value].copy() if nrm is not None else None)


def knn_graph(points, k, radius=0.0, want_d2=True, ctx=None):
    ctx = ctx or _lib.default_context()
    p = _c(points)
    k = int(min(k, len(p)))
    nbr = np.empty((len(p), k), np.int32)
    d2 = np.empty((len(p), k)) if want_d2 else None
    ctx.call("r3d_knn_graph", _ptr(p), len(p), k, float(radius), nbr.ctypes.data_as(_vp), _ptr(d2))
    return nbr, d2


def orient_normals(points, normals, k=100, delaunay_edges=None, ctx=None):
    """orient_normals_consistent_tangent_plane(k): returns the re-oriented copy of `normals`.  delaunay_edges ([m,2] int):
    edges of the cloud's Delaunay tetrahedralisation -> the exact Open3D graph (r3d_orient_normals_graph); None -> the
    k-NN-graph-only variant (r3d_orient_normals, a documented deviation)."""
    ctx = ctx or _lib.default_context()
    p = _c(points)
    n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3).copy()
    if len(p) == 0:
        return n
    if delaunay_edges is None:
        ctx.call("r3d_orient_normals", _ptr(p), len(p), int(k), _ptr(n))
    else:
        e = np.ascontiguousarray(delaunay_edges, dtype=np.int32).reshape(-1, 2)
        ctx.call("r3d_orient_normals_graph", _ptr(p), len(p), int(k), e.ctypes.data_as(_vp), len(e), _ptr(n))
    return n


def statistical_outlier_mask(points, nb_neighbors, std_ratio, ctx=None):
    """remove_statistical_outlier(nb_neighbors, std_ratio) (pointcloud_processing.py:35) [recalled RemoveStatisticalOutliers]:
    score a_i = mean distance to the k nearest (self included); the cloud mean sums the scores > 0 and divides by the number
    of points, the deviation sums over scores > 0 and divides by n - 1; keep iff 0 < a_i < mean + ratio * std.  Scores of 0
    need >= k coincident points; without them this is the plain mean / std(ddof=1) rule the recorded PLY files verify."""
    n = len(_c(points))
    if n == 0:
        return np.zeros(0, bool)
    a = neighbor_score(points, k=min(int(nb_neighbors), n), ctx=ctx)
    pos = a > 0
    mean = a[pos].sum() / n
    std = np.sqrt(((a[pos] - mean) ** 2).sum() / (n - 1)) if n > 1 else 0.0
    return pos & (a < mean + std_ratio * std)


def radius_outlier_mask(points, nb_points, radius, ctx=None):
    if len(_c(points)) == 0:
        return np.zeros(0, bool)
    return neighbor_score(points, count_radius=radius, ctx=ctx) > nb_points


def transform_points(points, T, rotate_only=False, ctx=None):
    ctx = ctx or _lib.default_context()
    p = _c(points)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    out = np.empty_like(p)
    ctx.call("r3d_transform_points", _ptr(p), len(p), _ptr(T), int(bool(rotate_only)), _ptr(out))
    return out


def registration(source, target, max_correspondence_distance, init=None, mode=P2P, max_iteration=30,
                 relative_fitness=1e-6, relative_rmse=1e-6, source_normals=None, target_normals=None,
                 gicp_epsilon=1e-3, ctx=None):
    """registration_icp / registration_generalized_icp.  Returns dict(T, fitness, inlier_rmse, iterations, ...)."""
    ctx = ctx or _lib.default_context()
    s, t, sn, tn = _c(source), _c(target), _c(source_normals), _c(target_normals)
    T0 = None if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
    T = np.empty((4, 4))
    prm = _lib.IcpParams(int(mode), int(max_iteration), float(max_correspondence_distance), float(relative_fitness),
                         float(relative_rmse), float(gicp_epsilon))
    st = _lib.IcpStats()
    ctx.call("r3d_icp", ctypes.byref(prm), _ptr(s), len(s), _ptr(sn), _ptr(t), len(t), _ptr(tn), _ptr(T0), _ptr(T),
             ctypes.byref(st))
    return _stats_dict(T, st)


def align_point_clouds(source, target, threshold=0.02, voxel_size=0.01, max_iteration=100, mode=P2P, normal_radius=None,
                       normal_max_nn=30, source_colors=None, init=None, relative_fitness=1e-6, relative_rmse=1e-6,
                       gicp_epsilon=1e-3, ctx=None):
    """The body of PointCloudAlignment.align_point_clouds (pointcloud_alignment.py:6-43) as one device-resident call
    (r3d_align_point_clouds): voxel_down_sample both -> normals on both -> registration -> transform the source.
    Returns dict(points, colors, normals, T, fitness, ...): the down-sampled, transformed source."""
    ctx = ctx or _lib.default_context()
    s, t, sc = _c(source), _c(target), _c(source_colors)
    T0 = None if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
    prm = AlignParams(_lib.IcpParams(int(mode), int(max_iteration), float(threshold), float(relative_fitness),
                                     float(relative_rmse), float(gicp_epsilon)),
                      float(voxel_size) if voxel_size else -1.0,
                      float(normal_radius) if normal_radius else (2.0 * voxel_size if voxel_size else -1.0),
                      int(normal_max_nn) if normal_max_nn else 0, 0)
    op = np.empty((len(s), 3))
    oc = np.empty((len(s), 3)) if sc is not None else None
    on = np.empty((len(s), 3)) if prm.normal_max_nn > 0 else None
    T = np.empty((4, 4))
    m = ctypes.c_int64()
    st = _lib.IcpStats()
    ctx.call("r3d_align_point_clouds", ctypes.byref(prm), _ptr(s), _ptr(sc), len(s), _ptr(t), len(t), _ptr(T0), _ptr(op),
             _ptr(oc), _ptr(on), ctypes.byref(m), _ptr(T), ctypes.byref(st))
    m = m.value
    return dict(points=op[:m].copy(), colors=None if oc is None else oc[:m].copy(), normals=None if on is None else on[:m].copy(),
                T=T, fitness=st.fitness, inlier_rmse=st.inlier_rmse, iterations=st.iterations, converged=bool(st.converged),
                correspondences=st.correspondences, setup_ms=st.setup_ms, loop_ms=st.loop_ms)


def _stats_dict(T, st):
    return dict(T=T, fitness=st.fitness, inlier_rmse=st.inlier_rmse, iterations=st.iterations, converged=bool(st.converged),
                correspondences=st.correspondences, setup_ms=st.setup_ms, loop_ms=st.loop_ms)


class ResidentModel:
    """The growing `combined_pcd` of the scanning loops (main.py:28,34-54; test/GICP1.py:134-155) kept in HBM (r3d_model_*):
    per frame only the frame is uploaded and the 4x4 + statistics come back; download() fetches the model once at the end."""

    def __init__(self, ctx=None):
        self.ctx = ctx or _lib.default_context()
        h = _vp()
        self.ctx.check(self.ctx._lib.r3d_model_create(self.ctx._h, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.r3d_model_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _call(self, name, *args):
        self.ctx.check(getattr(self.ctx._lib, name)(self._h, *args))

    def size(self):
        n, hc, hn = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        self._call("r3d_model_size", ctypes.byref(n), ctypes.byref(hc), ctypes.byref(hn))
        return n.value, bool(hc.value), bool(hn.value)

    def __len__(self):
        return self.size()[0]

    def voxel_table_stats(self):
        """(voxels in the resident legacy voxel table, full rebuilds, incremental updates) -- r3d_model_voxel_table_stats."""
        v, r, u = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        self._call("r3d_model_voxel_table_stats", ctypes.byref(v), ctypes.byref(r), ctypes.byref(u))
        return v.value, r.value, u.value

    def clear(self):
        self._call("r3d_model_clear")

    def append(self, points, colors=None, normals=None):
        p, c, n = _c(points), _c(colors), _c(normals)
        self._call("r3d_model_append", _ptr(p), _ptr(c), _ptr(n), len(p))

    def align_append(self, source, threshold=0.02, voxel_size=0.01, max_iteration=100, mode=P2P, normal_radius=None,
                     normal_max_nn=30, source_colors=None, relative_fitness=1e-6, relative_rmse=1e-6, gicp_epsilon=1e-3):
        """align_point_clouds(frame, model, threshold, voxel_size, max_iter) followed by model += aligned (main.py:48-49)."""
        s, sc = _c(source), _c(source_colors)
        prm = AlignParams(_lib.IcpParams(int(mode), int(max_iteration), float(threshold), float(relative_fitness),
                                         float(relative_rmse), float(gicp_epsilon)),
                          float(voxel_size) if voxel_size else -1.0,
                          float(normal_radius) if normal_radius else (2.0 * voxel_size if voxel_size else -1.0),
                          int(normal_max_nn) if normal_max_nn else 0, 0)
        T = np.empty((4, 4))
        st = _lib.IcpStats()
        m = ctypes.c_int64()
        self._call("r3d_model_align_append", ctypes.byref(prm), _ptr(s), _ptr(sc), len(s), _ptr(T), ctypes.byref(st), ctypes.byref(m))
        return dict(_stats_dict(T, st), appended=m.value)

    def register_append(self, source, threshold=0.02, mode=GICP, max_iteration=30, source_colors=None, source_normals=None,
                        relative_fitness=1e-6, relative_rmse=1e-6, gicp_epsilon=1e-3):
        """registration of the frame against the whole model, then model += transformed frame (test/GICP1.py:145-146)."""
        s, sc, sn = _c(source), _c(source_colors), _c(source_normals)
        prm = _lib.IcpParams(int(mode), int(max_iteration), float(threshold), float(relative_fitness), float(relative_rmse),
                             float(gicp_epsilon))
        T = np.empty((4, 4))
        st = _lib.IcpStats()
        self._call("r3d_model_register_append", ctypes.byref(prm), _ptr(s), _ptr(sc), _ptr(sn), len(s), _ptr(T), ctypes.byref(st))
        return _stats_dict(T, st)

    def append_depth(self, depth, camera, color=None):
        """First frame given as a depth image (uint16 [H,W]; color uint8 [H,W,3] optional): back-projected on the device."""
        d, c = _depth_args(depth, color)
        m = ctypes.c_int64()
        self._call("r3d_model_append_depth", d.ctypes.data_as(_vp), d.shape[1], d.shape[0], d.shape[1], ctypes.byref(camera),
                   c.ctypes.data_as(_vp) if c is not None else None, 3 * d.shape[1] if c is not None else 0, ctypes.byref(m))
        return m.value

    def align_append_depth(self, depth, camera, color=None, threshold=0.02, voxel_size=0.01, max_iteration=100, mode=P2P,
                           normal_radius=None, normal_max_nn=30, relative_fitness=1e-6, relative_rmse=1e-6, gicp_epsilon=1e-3):
        """main.py:48-49 for a frame that is still a depth image: create_from_rgbd_image + flip on the device, then align_append."""
        d, c = _depth_args(depth, color)
        prm = AlignParams(_lib.IcpParams(int(mode), int(max_iteration), float(threshold), float(relative_fitness),
                                         float(relative_rmse), float(gicp_epsilon)),
                          float(voxel_size) if voxel_size else -1.0,
                          float(normal_radius) if normal_radius else (2.0 * voxel_size if voxel_size else -1.0),
                          int(normal_max_nn) if normal_max_nn else 0, 0)
        T = np.empty((4, 4))
        st = _lib.IcpStats()
        npts, m = ctypes.c_int64(), ctypes.c_int64()
        self._call("r3d_model_align_append_depth", ctypes.byref(prm), d.ctypes.data_as(_vp), d.shape[1], d.shape[0], d.shape[1],
                   ctypes.byref(camera), c.ctypes.data_as(_vp) if c is not None else None, 3 * d.shape[1] if c is not None else 0,
                   _ptr(T), ctypes.byref(st), ctypes.byref(npts), ctypes.byref(m))
        return dict(_stats_dict(T, st), frame_points=npts.value, appended=m.value)

    def estimate_normals(self, radius=0.05, max_nn=30):
        self._call("r3d_model_estimate_normals", float(radius) if radius else -1.0, int(max_nn))

    def download(self):
        n, hc, hn = self.size()
        p = np.empty((n, 3))
        c = np.empty((n, 3)) if hc else None
        nr = np.empty((n, 3)) if hn else None
        if n:
            self._call("r3d_model_download", _ptr(p), _ptr(c), _ptr(nr))
        return p, c, nr


def disparity_to_cloud_resident(d_disp, width, height, Q, d_out_points, d_out_normals, capacity, min_disparity=0, max_depth=None,
                                pose=None, voxel=0.01, normal_radius=None, max_nn=30, ctx=None):
    """r3d_disparity_to_cloud_resident: like disparity_to_cloud_device, but the cloud is written to the caller's DEVICE buffers
    (pointers as ints, `capacity` triplets each; e.g. torch tensors' data_ptr()).  Returns the number of points."""
    ctx = ctx or _lib.default_context()
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(4, 4)
    P = None if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(4, 4)
    m = ctypes.c_int64()
    ctx.call("r3d_disparity_to_cloud_resident", _vp(d_disp), int(width), int(height), _ptr(Q), int(min_disparity) * 16,
             float(max_depth) if max_depth else -1.0, _ptr(P), float(voxel) if voxel else -1.0,
             float(normal_radius) if normal_radius else -1.0, int(max_nn) if max_nn else 0, int(capacity), _vp(d_out_points),
             _vp(d_out_normals) if d_out_normals else None, ctypes.byref(m))
    return m.value


def registration_device(d_source, ns, d_target, nt, max_correspondence_distance, init=None, mode=P2P, max_iteration=30,
                        relative_fitness=1e-6, relative_rmse=1e-6, d_source_normals=None, d_target_normals=None, gicp_epsilon=1e-3,
                        ctx=None):
    """r3d_icp_dev: registration of two clouds that are already in HBM (device pointers as ints)."""
    ctx = ctx or _lib.default_context()
    T0 = None if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
    T = np.empty((4, 4))
    prm = _lib.IcpParams(int(mode), int(max_iteration), float(max_correspondence_distance), float(relative_fitness),
                         float(relative_rmse), float(gicp_epsilon))
    st = _lib.IcpStats()
    ctx.call("r3d_icp_dev", ctypes.byref(prm), _vp(d_source), int(ns), _vp(d_source_normals) if d_source_normals else None,
             _vp(d_target), int(nt), _vp(d_target_normals) if d_target_normals else None, _ptr(T0), _ptr(T), ctypes.byref(st))
    return _stats_dict(T, st)


def transform_points_device(d_points, n, T, d_out, rotate_only=False, ctx=None):
    ctx = ctx or _lib.default_context()
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    ctx.call("r3d_transform_points_dev", _vp(d_points), int(n), _ptr(T), int(bool(rotate_only)), _vp(d_out))


def transform_blocks_device(blocks, ctx=None):
    """blocks: sequence of (d_in, n, T 4x4, d_out, rotate_only) -- every block moved by its own rigid transform in one launch per
    16 blocks (r3d_transform_blocks_dev); device pointers as ints, asynchronous on the context stream."""
    ctx = ctx or _lib.default_context()
    nb = len(blocks)
    if nb == 0:
        return
    pin = (ctypes.c_void_p * nb)(*[int(b[0]) for b in blocks])
    pout = (ctypes.c_void_p * nb)(*[int(b[3]) for b in blocks])
    cnt = np.ascontiguousarray([int(b[1]) for b in blocks], dtype=np.int64)
    Ts = np.ascontiguousarray(np.stack([np.asarray(b[2], dtype=np.float64).reshape(4, 4) for b in blocks]))
    ro = np.ascontiguousarray([1 if b[4] else 0 for b in blocks], dtype=np.int32)
    ctx.call("r3d_transform_blocks_dev", nb, ctypes.cast(pin, _vp), cnt.ctypes.data_as(_vp), _ptr(Ts), ro.ctypes.data_as(_vp), ctypes.cast(pout, _vp))


def backproject_depth(depth, camera, color=None, want_pixels=False, ctx=None):
    """create_from_rgbd_image(+ flip) of test/check84.py:155-159,172-178 (r3d_backproject_depth).  depth: uint16 [H,W];
    camera: depth_camera(...); color: uint8 [H,W,3] or None.  Returns (points [M,3], colors [M,3] or None[, pixel index [M]])."""
    ctx = ctx or _lib.default_context()
    d, c = _depth_args(depth, color)
    H, W = d.shape
    pts = np.empty((H * W, 3))
    col = np.empty((H * W, 3)) if c is not None else None
    pix = np.empty(H * W, np.int32) if want_pixels else None
    m = ctypes.c_int64()
    ctx.call("r3d_backproject_depth", d.ctypes.data_as(_vp), W, H, W, ctypes.byref(camera), c.ctypes.data_as(_vp) if c is not None else None,
             3 * W if c is not None else 0, _ptr(pts), _ptr(col), pix.ctypes.data_as(_vp) if want_pixels else None, ctypes.byref(m))
    n = m.value
    out = (pts[:n].copy(), col[:n].copy() if col is not None else None)
    return out + (pix[:n].copy(),) if want_pixels else out
