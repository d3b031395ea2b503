"""Orchestration counterparts of the reference's per-frame loops, on top of the HIP-backed stage classes.

fuse()             main.py:34-54 simple_scanning_loop (and test/GICP1.py:134-155): skip empty frames, first valid frame
                   initialises the model, every later frame is aligned to the model and appended.
view_to_cloud()    the join the reference never wrote (Q is loaded at Calib_depth/depth2.py:66 and then unused):
                   StereoSGBM disparity -> reprojectImageTo3D -> voxel_down_sample -> normals.
multi_view_fuse()  BASELINE config C5: one view per rank, all-gather-v of the per-view clouds, every view registered
                   to view 0, fused cloud available on every rank (rank 0 hands it to the mesher).
"""
import numpy as np

from . import cloud_ops, distributed
from .pointcloud import PointCloud, as_arrays
from .pointcloud_alignment import GeneralizedICPAlignment, PointCloudAlignment


def fuse(frames, flavour="icp", verbose=False, **align_kw):
    """frames: iterable of PointCloud-likes (or None / empty for a failed capture).  Returns the accumulated model."""
    model = PointCloud()
    aligner = PointCloudAlignment(verbose=verbose) if flavour == "icp" else GeneralizedICPAlignment()
    for frame in frames:
        if frame is None or len(as_arrays(frame)[0]) == 0:
            if verbose:
                print("No valid point cloud captured, skipping frame.")      # main.py:53-54
            continue
        p, c, n = as_arrays(frame)
        if len(model.points) == 0:
            model.points = p.copy()                                          # main.py:42-45: points + colors only
            model.colors = c.copy() if c is not None else np.zeros((0, 3))
            if flavour == "gicp" and n is not None:
                model.normals = n.copy()
            continue
        aligned = aligner.align_point_clouds(frame, model, **align_kw)
        model += aligned                                                     # main.py:49
        if flavour == "gicp":                                                # GICP1.py:148: re-estimate on the model
            model.estimate_normals(radius=0.05, max_nn=30)
    return model


def scaled_Q(Q, scale, unit=1.0):
    """Q of the 960x540 rig (tests/golden/jetson_stereo_8MP_stereo.npz) re-expressed for images `scale` times larger.
    The rig was calibrated in millimetres (baseline 31.5): unit=1e-3 yields metres, the unit of the scanner half."""
    Q = np.array(Q, dtype=np.float64)
    Q[0, 3] *= scale
    Q[1, 3] *= scale
    Q[2, 3] *= scale
    Q[:3] *= unit
    return Q


def view_to_cloud(left, right, Q, matcher, voxel=0.01, normal_radius=None, max_nn=30, pose=None, max_depth=None,
                  device_resident=True):
    """One stereo view -> down-sampled cloud with normals.  pose (4x4, optional) is applied to the cloud.
    device_resident (default): the disparity map never leaves HBM; the matcher writes it with compute_device and
    r3d_disparity_to_cloud_dev chains reprojection, depth filter, pose, voxel grid and normals on the device, so only the
    two input images go up and the final cloud comes down.  False: the same stages through the host-buffer entry points
    (identical results; kept for callers that want the intermediate arrays)."""
    if device_resident:
        ctx = matcher.context
        L = np.ascontiguousarray(left, dtype=np.uint8)
        R = np.ascontiguousarray(right, dtype=np.uint8)
        H, W = L.shape
        d_l, d_r, d_d = ctx.to_device(L), ctx.to_device(R), ctx.alloc(W * H * 2)
        try:
            matcher.compute_device(d_l, d_r, W, H, W, d_d)
            pts, nrm = cloud_ops.disparity_to_cloud_device(d_d, W, H, Q, matcher.getMinDisparity(), max_depth, pose, voxel,
                                                           normal_radius or 2 * voxel, max_nn, ctx=ctx)
        finally:
            for p in (d_l, d_r, d_d):
                ctx.free(p)
        return PointCloud(pts, normals=nrm) if len(pts) else PointCloud()
    disp = matcher.compute(left, right)
    pts = cloud_ops.reproject_disparity(disp, Q, matcher.getMinDisparity())
    if len(pts):                                       # zero disparities reproject to infinity (W = 0): never part of a cloud
        pts = pts[np.abs(pts[:, 2]) <= (max_depth if max_depth is not None else 1.0e300)]
    if len(pts) == 0:
        return PointCloud()
    if pose is not None:
        pts = cloud_ops.transform_points(pts, pose)
    pts, _, _ = cloud_ops.voxel_down_sample(pts, voxel)
    nrm = cloud_ops.estimate_normals(pts, normal_radius or 2 * voxel, max_nn)
    return PointCloud(pts, normals=nrm)


def multi_view_fuse(local_clouds, n_views, threshold=0.02, mode=cloud_ops.GICP, max_iteration=30, register=None):
    """local_clouds: {view_id: PointCloud with normals} owned by this rank (distributed.shard_views).
    Exchange once, register every owned view to view 0, exchange the 4x4 transforms, return
    (fused PointCloud, {view_id: T}).  `register` may replace the HIP registration (the CPU tests inject a stub)."""
    payload = {}
    for v, pc in local_clouds.items():
        p, _, n = as_arrays(pc)
        payload[v] = np.concatenate([p, n if n is not None else np.zeros_like(p)], 1)
    everyone = distributed.gather_rows_by_view(payload, n_views)
    ref = everyone[0]
    if register is None:
        def register(src, tgt):
            return cloud_ops.registration(src[:, :3], tgt[:, :3], threshold, mode=mode, max_iteration=max_iteration,
                                          source_normals=src[:, 3:], target_normals=tgt[:, 3:])["T"]
    mine = {}
    for v in local_clouds:
        T = np.eye(4) if v == 0 else np.asarray(register(everyone[v], ref), dtype=np.float64)
        mine[v] = T.reshape(1, 16)
    Ts = distributed.gather_rows_by_view(mine, n_views)
    fused = PointCloud()
    out_T = {}
    for v in range(n_views):
        T = Ts[v].reshape(4, 4)
        out_T[v] = T
        blk = everyone[v]
        pts = blk[:, :3] @ T[:3, :3].T + T[:3, 3]
        nrm = blk[:, 3:] @ T[:3, :3].T
        fused += PointCloud(pts, normals=nrm)
    return fused, out_T
