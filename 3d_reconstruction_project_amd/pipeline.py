"""Orchestration counterparts of the reference's per-frame loops, on top of the HIP-backed stage classes.

fuse()             main.py:34-54 simple_scanning_loop (and test/GICP1.py:134-155): skip empty frames, first valid frame
                   initialises the model, every later frame is aligned to the model and appended.
view_to_cloud()    the join the reference never wrote (Q is loaded at Calib_depth/depth2.py:66 and then unused):
                   StereoSGBM disparity -> reprojectImageTo3D -> voxel_down_sample -> normals.
multi_view_fuse_tensors()  BASELINE config C5: one view per rank, ONE all-gather-v of the per-view clouds (RCCL, clouds stay in
                   HBM), every view registered to view 0, fused cloud on every rank (rank 0 hands it to the mesher,
                   mesh_reconstruction.py:22-37); multi_view_fuse() is its host-cloud front end.
"""
import ctypes

import numpy as np

from . import cloud_ops, distributed
from .pointcloud import PointCloud, as_arrays
from .pointcloud_alignment import GeneralizedICPAlignment, PointCloudAlignment


def fuse(frames, flavour="icp", verbose=False, resident=True, ctx=None, log=None, **align_kw):
    """frames: iterable of PointCloud-likes (or None / empty for a failed capture).  Returns the accumulated model.
    flavour "icp": main.py:34-54 (align_point_clouds(frame, model, threshold=0.02, voxel_size=0.01, max_iter=100), model += aligned);
    flavour "gicp": test/GICP1.py:134-155 (registration_generalized_icp of the frame against the whole model, model += aligned,
    model.estimate_normals(Hybrid(0.05, 30))).  align_kw: threshold / voxel_size / max_iter (icp), threshold (gicp).
    resident (default): the model stays in HBM across frames (cloud_ops.ResidentModel): per frame only the frame goes up and the
    4x4 comes back, the model is downloaded once at the end.  resident=False drives the drop-in classes with a host-side model
    exactly as the reference's loop does (identical results; every frame then re-uploads the whole model).
    `log` (list) receives the registration result of every aligned frame."""
    if not resident:
        return _fuse_host(frames, flavour, verbose, log, **align_kw)
    model = first = None
    for frame in frames:
        if frame is None or len(as_arrays(frame)[0]) == 0:
            if verbose:
                print("No valid point cloud captured, skipping frame.")      # main.py:53-54
            continue
        p, c, n = as_arrays(frame)
        if first is None:
            # main.py:42-45: points + colors only; GICP1.py:139-141: normals as well.  Stays on the host until a second
            # frame needs a model to register against (a one-frame scan never touches the GPU)
            first = (p.copy(), None if c is None else c.copy(), n.copy() if (flavour == "gicp" and n is not None) else None)
            continue
        if model is None:
            model = cloud_ops.ResidentModel(ctx)
            model.append(*first)
        if flavour == "icp":
            if verbose:                                                      # the three progress lines of align_point_clouds
                print("Downsampling point clouds using voxel size:", align_kw.get("voxel_size", 0.01))
                print("Estimating normals on CPU...")
                print("Performing ICP alignment using CUDA...")
            vs = align_kw.get("voxel_size", 0.01)
            res = model.align_append(p, align_kw.get("threshold", 0.02), vs, align_kw.get("max_iter", 100), cloud_ops.P2P, 2 * vs, 30,
                                     source_colors=c)
        else:
            if n is None:                                                    # GICP1.py:94-95
                n = cloud_ops.estimate_normals(p, 0.05, 30, ctx=ctx)
            if not model.size()[2]:                                          # GICP1.py:96-97
                model.estimate_normals(0.05, 30)
            res = model.register_append(p, align_kw.get("threshold", 0.02), cloud_ops.GICP, 30, source_colors=c, source_normals=n)
            model.estimate_normals(0.05, 30)                                 # GICP1.py:148
        if log is not None:
            log.append(res)
    if model is None:
        return PointCloud(*((first[0], first[1], first[2]) if first is not None else ()))
    pts, cols, nrm = model.download()
    model.close()
    return PointCloud(pts, cols, nrm)


def fuse_depth_frames(depth_images, camera, colors=None, verbose=False, ctx=None, log=None, threshold=0.02, voxel_size=0.01, max_iter=100):
    """main.py:34-54 for frames that arrive as depth images (the recorded test/output84 frames; a RealSense z16 frame):
    create_from_rgbd_image + flip (test/check84.py:155-159,172-178) runs on the device, the model stays in HBM, the frame that
    crosses PCIe is the 0.6 MB image.  depth_images: iterable of uint16 [H,W] arrays (None / all-invalid = failed capture,
    skipped); camera: cloud_ops.depth_camera(intrinsics); colors: optional iterable of uint8 [H,W,3] images.
    Identical to fuse() over the back-projected clouds."""
    model = None
    colors = iter(colors) if colors is not None else None
    for depth in depth_images:
        col = next(colors) if colors is not None else None
        if depth is None or not np.any(depth):
            if verbose:
                print("No valid point cloud captured, skipping frame.")      # main.py:53-54
            continue
        if model is None:
            model = cloud_ops.ResidentModel(ctx)
            if model.append_depth(depth, camera, col) == 0:
                model.close()
                model = None
            continue
        res = model.align_append_depth(depth, camera, col, threshold, voxel_size, max_iter, cloud_ops.P2P, 2 * voxel_size, 30)
        if res["frame_points"] == 0:                                          # nothing within depth_trunc: a failed capture
            continue
        if log is not None:
            log.append(res)
    if model is None:
        return PointCloud()
    pts, cols, nrm = model.download()
    if log is not None:
        v, r, u = model.voxel_table_stats()
        log_stats = {"voxel_table": {"voxels": v, "rebuilds": r, "updates": u}}
        if log:
            log[-1].update(log_stats)
    model.close()
    return PointCloud(pts, cols, nrm)


def _fuse_host(frames, flavour, verbose, log, **align_kw):
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
        if log is not None:
            log.append(aligner.last_result)
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


def view_to_cloud_tensors(d_left, d_right, d_disp, width, height, Q, matcher, out, voxel=0.01, normal_radius=None, max_nn=30,
                          pose=None, max_depth=None):
    """One stereo view whose images are already in HBM -> cloud left in HBM.  d_left / d_right / d_disp: device pointers
    (uint8 images, int16 disparity scratch); out: float64 torch tensor [2, capacity, 3] on the matcher's device that receives
    points (plane 0) and normals (plane 1).  Returns the tensor view out[:, :n] ([2, n, 3]).  Nothing crosses PCIe but the
    point count.  The library kernels are ordered with torch's current stream (distributed.shared_stream): whatever torch
    queued on the inputs before the call is seen, and torch ops / collectives queued on `out` afterwards see the cloud."""
    ctx = matcher.context
    with distributed.shared_stream(ctx):
        matcher.compute_device(d_left, d_right, width, height, width, d_disp)
        cap = out.shape[1]
        n = cloud_ops.disparity_to_cloud_resident(d_disp, width, height, Q, out[0].data_ptr(), out[1].data_ptr(), cap,
                                                  matcher.getMinDisparity(), max_depth, pose, voxel, normal_radius or 2 * voxel,
                                                  max_nn, ctx=ctx)
    return out[:, :n]


def views_to_cloud_tensors(pairs, d_disps, width, height, Q, matcher, outs, cloud_ctx, voxel=0.01, normal_radius=None, max_nn=30,
                           poses=None, max_depth=None):
    """Several views owned by ONE GPU (a rank that holds more than one view of the C5 batch): the same results as one
    view_to_cloud_tensors call per view, but the SGM maps go through the matcher's batch entry point (three maps in flight on
    the library's lanes, r3d_sgbm_compute_batch_events_dev) and the cloud stages of view i (reprojection -> voxel grid ->
    normals) start on map i's completion event, underneath the SGM kernels of the later views, on `cloud_ctx`: a second Context
    with its own stream and arena, or a LIST of them -- the cloud stages are a chain of small kernels with host round trips in
    between, so each extra context gets its own host thread (ctypes releases the GIL) and the chains of different views overlap.
    pairs: [(d_left, d_right)] device pointers; d_disps: one int16 device buffer PER view; outs: one [2, capacity, 3] float64
    tensor per view.  Returns [out_i[:, :n_i]]."""
    import torch
    ctx = matcher.context
    n = len(pairs)
    cctxs = list(cloud_ctx) if isinstance(cloud_ctx, (list, tuple)) else [cloud_ctx]
    assert len(d_disps) == n and len(outs) == n and all(c is not ctx for c in cctxs) and len({id(c) for c in cctxs}) == len(cctxs)
    dev = outs[0].device
    evs = [ctx.event() for _ in range(n)]
    res = [None] * n
    try:
        with distributed.shared_stream(ctx) as sa:
            import os
            prio = -1 if os.environ.get("R3D_CLOUD_PRIO") == "1" else 0
            sbs = [torch.cuda.Stream(device=dev, priority=prio) for _ in cctxs]
            prev = [c.get_stream() for c in cctxs]
            for c, sb, pv in zip(cctxs, sbs, prev):
                sb.wait_stream(sa)
                if pv:        # work the cloud context queued earlier on its previous stream (it reuses the same arena) stays ahead
                    sb.wait_stream(torch.cuda.ExternalStream(pv, device=dev))
                c.set_stream(sb.cuda_stream)
            try:
                matcher.compute_batch_device([p[0] for p in pairs], [p[1] for p in pairs], width, height, width, list(d_disps),
                                             done_events=evs)

                def chain(w):
                    c = cctxs[w]
                    for i in range(w, n, len(cctxs)):
                        c.wait_event(evs[i])
                        k = cloud_ops.disparity_to_cloud_resident(d_disps[i], width, height, Q, outs[i][0].data_ptr(), outs[i][1].data_ptr(),
                                                                  outs[i].shape[1], matcher.getMinDisparity(), max_depth,
                                                                  None if poses is None else poses[i], voxel, normal_radius or 2 * voxel,
                                                                  max_nn, ctx=c)
                        res[i] = outs[i][:, :k]
                if len(cctxs) == 1:
                    chain(0)
                else:
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(len(cctxs)) as pool:
                        for f in [pool.submit(chain, w) for w in range(len(cctxs))]:
                            f.result()                       # re-raises a worker's exception here
            finally:
                # on EVERY path (a worker may have raised with the other chains' kernels still queued): the matcher's stream joins
                # the side streams, so that the ctx.sync() below also covers kernels that write outs[i] / read d_disps[i]
                for sb in sbs:
                    sa.wait_stream(sb)
                for c, p in zip(cctxs, prev):
                    c.set_stream(p)
    finally:
        ctx.sync()
        for e in evs:
            ctx.call("r3d_event_destroy", ctypes.c_void_p(e))
    return res


def multi_view_fuse_tensors(local, n_views, threshold=0.02, mode=cloud_ops.GICP, max_iteration=30, register=None, transform=None,
                            ctx=None, timings=None):
    """BASELINE config C5 on clouds that live where the exchange runs (HBM under RCCL).  local: {view_id: float64 tensor
    [2, n, 3]} (points, normals) of the views this rank owns (distributed.shard_views).  Steps: ONE all-gather-v of the clouds
    (distributed.gather_views) -> every owned view is registered to view 0 (r3d_icp_dev on the gathered blocks, in place) ->
    the 4x4s are all-gathered -> every view is transformed into the frame of view 0 (r3d_transform_blocks_dev: one launch) into one fused
    [2, N, 3] tensor, which every rank ends up holding (rank 0 hands it to the mesher, mesh_reconstruction.py:22-37).
    Returns (fused tensor, {view_id: T}).  `register(src [2,n,3], tgt [2,m,3]) -> 4x4` and `transform(block, T) -> block`
    replace the HIP calls in the CPU (gloo) tests; with device tensors and no stubs the HIP library does the work.
    timings (dict, optional) receives exchange_ms / register_ms / fuse_ms measured with events on the shared stream.
    (A rank that owns several views registers them one after the other: running them on several contexts / host threads at
    once was measured -- 7 registrations of 109 k-point clouds: 2.8 ms sequential, 3.3-3.8 ms on two contexts, 2.5-3.0 ms on
    four -- and dropped.)"""
    import torch
    dev = distributed.exchange_device()
    on_gpu = dev.type == "cuda"
    if not on_gpu and (register is None or transform is None):
        raise RuntimeError("multi_view_fuse_tensors: host tensors need injected register / transform stubs; the product path "
                           "runs on device tensors through libr3d_hip.so (no CPU fallback)")
    if on_gpu and ctx is None:
        from . import _lib
        ctx = _lib.default_context(dev.index or 0)
    if on_gpu:
        # ONE stream for torch ops, RCCL collectives and library kernels for the whole exchange (the context's own
        # stream is not ordered against torch's)
        outer = torch.cuda.current_stream(dev)
        with distributed.shared_stream(ctx):
            fused, Ts = _multi_view_fuse_on_stream(local, n_views, threshold, mode, max_iteration, register, transform, ctx, timings, dev)
        fused.record_stream(outer)
        return fused, Ts
    return _multi_view_fuse_on_stream(local, n_views, threshold, mode, max_iteration, register, transform, ctx, timings, dev)


def _multi_view_fuse_on_stream(local, n_views, threshold, mode, max_iteration, register, transform, ctx, timings, dev):
    import torch
    on_gpu = dev.type == "cuda"

    def mark():
        if not on_gpu:
            import time
            return time.perf_counter()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def lap(a, b):
        if not on_gpu:
            return 1e3 * (b - a)
        b.synchronize()
        return a.elapsed_time(b)

    t0 = mark()
    everyone = distributed.gather_views(local, n_views)
    t1 = mark()
    ref = everyone[0]
    mine = {}

    def register_view(v, c):
        # a registration that fails on this rank must not strand the others in the next collective: the exception travels as
        # a flag through gather_transforms, which then raises on every rank together
        try:
            if v == 0:
                return np.eye(4)
            if register is not None:
                return np.asarray(register(everyone[v], ref), dtype=np.float64)
            src = everyone[v]
            return cloud_ops.registration_device(src[0].data_ptr(), src.shape[1], ref[0].data_ptr(), ref.shape[1], threshold,
                                                 mode=mode, max_iteration=max_iteration, d_source_normals=src[1].data_ptr(),
                                                 d_target_normals=ref[1].data_ptr(), ctx=c)["T"]
        except Exception as e:  # noqa: BLE001
            return e
    for v in local:
        mine[v] = register_view(v, ctx)
    t2 = mark()
    Ts = distributed.gather_transforms(mine, n_views)
    total = sum(everyone[v].shape[1] for v in range(n_views))
    fused = torch.empty((2, total, 3), dtype=torch.float64, device=dev)
    o = 0
    blocks = []
    for v in range(n_views):
        blk, n = everyone[v], everyone[v].shape[1]
        if transform is not None:
            fused[:, o:o + n] = transform(blk, Ts[v])
        elif n:
            blocks.append((blk[0].data_ptr(), n, Ts[v], fused[0, o:o + n].data_ptr(), False))
            blocks.append((blk[1].data_ptr(), n, Ts[v], fused[1, o:o + n].data_ptr(), True))
        o += n
    if blocks:       # points and normals of all views in ONE launch (sixteen calls cost 0.3 ms of host overhead, profiles/r03_bench.json)
        cloud_ops.transform_blocks_device(blocks, ctx=ctx)
    t3 = mark()
    if timings is not None:
        timings.update(exchange_ms=lap(t0, t1), register_ms=lap(t1, t2), fuse_ms=lap(t2, t3))
    return fused, Ts


def multi_view_fuse(local_clouds, n_views, threshold=0.02, mode=cloud_ops.GICP, max_iteration=30, register=None, ctx=None):
    """Host-cloud front end of multi_view_fuse_tensors.  local_clouds: {view_id: PointCloud with normals} owned by this rank.
    The clouds go up once (to the device the exchange runs on), everything else stays there; returns
    (fused PointCloud, {view_id: T}).  `register(src [n,6], tgt [m,6]) -> 4x4` may replace the HIP registration (the CPU tests
    inject a stub; rows are xyz | normal)."""
    import torch
    dev = distributed.exchange_device()
    local = {}
    for v, pc in local_clouds.items():
        p, _, n = as_arrays(pc)
        local[v] = torch.from_numpy(np.stack([p, n if n is not None else np.zeros_like(p)], 0)).to(dev)
    reg = tr = None
    if register is not None:
        def reg(src, tgt):
            return register(torch.cat([src[0], src[1]], 1).cpu().numpy(), torch.cat([tgt[0], tgt[1]], 1).cpu().numpy())
    if dev.type != "cuda" or register is not None:
        def tr(blk, T):
            R = torch.from_numpy(np.ascontiguousarray(T[:3, :3])).to(blk.device)
            t = torch.from_numpy(np.ascontiguousarray(T[:3, 3])).to(blk.device)
            return torch.stack([blk[0] @ R.T + t, blk[1] @ R.T], 0)
    fused, Ts = multi_view_fuse_tensors(local, n_views, threshold, mode, max_iteration, reg, tr, ctx)
    arr = fused.cpu().numpy()
    return PointCloud(arr[0], normals=arr[1]), Ts
