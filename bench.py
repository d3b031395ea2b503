#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: disparity-maps/s at 8 MP (3264x2448), 128 disparities.

One "step" = one full StereoSGBM(3-way).compute over one synthetic rectified pair that is already resident in
HBM (one pair per rank; weak scaling: every rank owns one view, no data-path collective inside the SGM step).
Prints ONE JSON line on rank 0.  Launch for N>1:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, D = 3264, 2448, 128                     # BASELINE config C2
C2_KW = dict(minDisparity=0, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, uniquenessRatio=15,
             speckleWindowSize=0, speckleRange=2, preFilterCap=63)      # Calib_depth/depth2.py:139-158
HBM_PEAK = 8.0e12                             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
W1 = W - D
# algorithmic bytes: whole map = 4*W*H + 4*W*H*D with int16 costs; per kernel of the current split:
ALG_BYTES = {
    "map": 4 * W * H + 4 * W1 * H * D,
    "hscan": 2 * W1 * H * D,                  # the one compulsory volume WRITE (L_left + L_right)
    "hscan_bwd": 2 * W1 * H * D,              # same write, issued by the backward-phase launch when the phases are separate
    "vscan_wta": 2 * W1 * H * D + 4 * W * H,  # the one compulsory volume READ + disparity/cost out
    "cost": 2 * W * H,                        # reads both images; the cost volume itself is not algorithmic
}
KERNEL_OF = {"cost": "k_cost2", "hscan": "k_hscan2", "hscan_bwd": "k_hscan2", "vscan_wta": "k_vscan2", "prefilter": "k_prefilter",
             "lrcheck": "k_lrcheck", "median3": "k_median3"}


def bench_gicp(r3d, ctx, n=1_000_000, iters=20, cpu=True):
    """Secondary metric of BASELINE.json (config C3): GICP iterations/s on a 1M-point cloud pair, normals from
    kNN(20) PCA (the 'no normals' GICP branch), exactly `iters` iterations (criteria set so they never trigger),
    one-off setup (uploads, grid build, normals) reported separately.  SURVEY.md 8d: 80 B/source point + cell table
    = 88 MB algorithmic per iteration."""
    import numpy as np
    co = r3d.cloud_ops
    src, tgt, T_star = r3d.synth.cloud_pair(n)
    src, tgt = src.astype(np.float64), tgt.astype(np.float64)
    t0 = time.perf_counter()
    sn = co.estimate_normals(src, None, 20, ctx=ctx)
    tn = co.estimate_normals(tgt, None, 20, ctx=ctx)
    normals_cold_s = time.perf_counter() - t0          # includes the one-off growth of the device arena
    t0 = time.perf_counter()
    sn = co.estimate_normals(src, None, 20, ctx=ctx)
    tn = co.estimate_normals(tgt, None, 20, ctx=ctx)
    normals_s = time.perf_counter() - t0
    co.registration(src, tgt, 0.02, mode=co.GICP, max_iteration=2, relative_fitness=-1, relative_rmse=-1,
                    source_normals=sn, target_normals=tn, ctx=ctx)                        # warm-up (allocations)
    runs = [co.registration(src, tgt, 0.02, mode=co.GICP, max_iteration=iters, relative_fitness=-1, relative_rmse=-1,
                            source_normals=sn, target_normals=tn, ctx=ctx) for _ in range(3)]
    res = sorted(runs, key=lambda r: r["loop_ms"])[1]        # median of 3 repetitions
    # the same registration on clouds that are already resident in HBM (r3d_icp_dev): set-up without the four 24 MB uploads
    d_bufs = [ctx.to_device(a) for a in (src, sn, tgt, tn)]
    resident = sorted((co.registration_device(d_bufs[0], len(src), d_bufs[2], len(tgt), 0.02, mode=co.GICP, max_iteration=iters,
                                              relative_fitness=-1, relative_rmse=-1, d_source_normals=d_bufs[1],
                                              d_target_normals=d_bufs[3], ctx=ctx) for _ in range(3)), key=lambda r: r["setup_ms"])[1]
    for b in d_bufs:
        ctx.free(b)
    if abs(resident["inlier_rmse"] - res["inlier_rmse"]) > 1e-15 or np.abs(resident["T"] - res["T"]).max() > 0:
        raise SystemExit("GICP bench: device-resident registration differs from the host-array one")
    per_iter_ms = res["loop_ms"] / (iters + 1)               # iters+1 evaluate launches, iters solves
    err = float(np.linalg.norm(res["T"] - T_star))
    alg = 88e6
    traffic = traffic_src = None
    tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tp):                                   # last committed counter pass (tools/gpu_profile_round.sh), per launch
        with open(tp) as f:
            t = json.load(f).get("k_icp_eval")
        if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            traffic = round((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / 1e9, 4)
            traffic_src = "profiles/traffic_latest.json (median of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over k_icp_eval, file dated %s)" % time.strftime(
                "%Y-%m-%d", time.gmtime(os.path.getmtime(tp)))
    if err > 1e-3:                                           # SURVEY 8d pass rule for C3
        raise SystemExit(f"GICP bench: ||T - T*||_F = {err:.3e} > 1e-3")
    out = {"metric": "GICP iterations/s @1M pts", "value": round(1e3 / per_iter_ms, 2), "unit": "iterations/s",
           "iterations": res["iterations"], "ms_per_iteration": round(per_iter_ms, 4),
           "setup_ms": {"normals_knn20_both_clouds": round(1e3 * normals_s, 1), "normals_knn20_both_clouds_first_call": round(1e3 * normals_cold_s, 1),
                        "grid_sort_upload": round(res["setup_ms"], 1), "grid_sort_resident": round(resident["setup_ms"], 2)},
           "fitness": round(res["fitness"], 5), "inlier_rmse": res["inlier_rmse"], "T_error_frobenius": err,
           "roofline": {"bound": "hbm", "achieved": round(alg / (per_iter_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(alg / (per_iter_ms * 1e-3) / HBM_PEAK, 5), "traffic": traffic,
                        "traffic_unit": "GB/launch (k_icp_eval)", "traffic_source": traffic_src}}
    if cpu:
        from oracle import cloud_oracle as oc               # checker timed as the CPU baseline ("port")
        k = 2
        tc = time.perf_counter()
        oc.registration(src, tgt, 0.02, mode="gicp", max_iteration=k, relative_fitness=-1, relative_rmse=-1,
                        target_normals=tn, target_cov=oc.covariances_from_normals(tn),
                        source_cov=oc.covariances_from_normals(sn))
        dt = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": round((k + 1) / dt, 3), "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{k} iterations ({k + 1} evaluations) at 1M points, numpy/scipy restatement of Open3D "
                                         "registration_generalized_icp (oracle/cloud_oracle.py): neighbour search with "
                                         f"cKDTree.query(workers=-1) on all {os.cpu_count()} host cpus, per-pair 3x3 algebra in numpy (one thread)"}
    return out


def bench_frame_loop(r3d, ctx, dL, dR, reps=5):
    """Informational: one iteration of the Calib_depth/depth2.py:243-257 frame loop at C2 size, everything resident in
    HBM: remap + grey (x2), left and right matcher, WLS filter, normalize.  Fixed-point identity-plus-fraction maps keep the
    rectified pair a valid stereo pair."""
    import ctypes
    import numpy as np
    vp = ctypes.c_void_p
    xs, ys = np.meshgrid(np.arange(W, dtype=np.int16), np.arange(H, dtype=np.int16))
    d_m1 = ctx.to_device(np.ascontiguousarray(np.stack([xs, ys], -1)))
    d_m2 = ctx.to_device(np.full((H, W), 5 * 32 + 7, np.uint16))
    bufs = [d_m1, d_m2]

    def alloc(n):
        bufs.append(ctx.alloc(n))
        return bufs[-1]
    hL, hR = np.empty((H, W), np.uint8), np.empty((H, W), np.uint8)
    ctx.d2h(hL, dL)
    ctx.d2h(hR, dR)
    d_fl = ctx.to_device(np.ascontiguousarray(np.stack([hL] * 3, -1)))
    d_fr = ctx.to_device(np.ascontiguousarray(np.stack([hR] * 3, -1)))
    bufs += [d_fl, d_fr]
    d_rl, d_rr, d_gl, d_gr = alloc(W * H * 3), alloc(W * H * 3), alloc(W * H), alloc(W * H)
    d_dl, d_dr, d_f, d_n = (alloc(W * H * 2) for _ in range(4))
    left = r3d.reference_matcher(numDisparities=D, blockSize=5)
    right = r3d.createRightMatcher(left)
    wls = r3d.createDisparityWLSFilter(left)
    wls.setLambda(8000)
    wls.setSigmaColor(1.5)
    left._ctx = right._ctx = wls._ctx = ctx

    def frame():
        for src, dst, g in ((d_fl, d_rl, d_gl), (d_fr, d_rr, d_gr)):
            ctx.call("r3d_remap_u8_dev", vp(src), W, H, W * 3, 3, vp(d_m1), vp(d_m2), W, H, 0, vp(dst), vp(g))
        left.compute_device(d_gl, d_gr, W, H, W, d_dl)
        right.compute_device(d_gr, d_gl, W, H, W, d_dr)
        wls.filter_device(d_dl, d_dr, d_gl, 1, W, W, H, d_f)
        ctx.call("r3d_normalize_minmax_s16_dev", vp(d_f), W * H, 0.0, 255.0, vp(d_n))
    for _ in range(2):
        frame()
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        frame()
    ctx.record(e1)
    ctx.sync()
    ms = ctx.elapsed_ms(e0, e1) / reps
    for b in bufs:
        ctx.free(b)
    return {"metric": "depth2.py frame iterations/s @8MP d=128 (remap+grey x2, SGBM left+right, WLS filter, normalize)",
            "value": round(1e3 / ms, 2), "unit": "frames/s", "ms_per_frame": round(ms, 4), "frames": reps}


def bench_c5(r3d, ctx, rank, world, n_views=8, reps=3):
    """BASELINE config C5: an 8-view 8 MP batch, views dealt round-robin to the ranks (one view per GPU at N = 8; all eight on
    the one GPU at N = 1, so c5.batch_ms at N = 1 vs N = 8 is the strong-scaling figure north_star asks for).  Per rank and view:
    SGM -> reprojection -> voxel 0.01 -> normals, device-resident; then ONE all-gather-v of the clouds over RCCL (device
    tensors, no host staging), every owned view registered to view 0 with GICP, all-gather of the 4x4s, every view transformed
    into view 0's frame.  View v shows the same synthetic scene (own texture / noise seed) displaced by a known pose, so the
    registration has a known answer.  Reported: per-stage ms (max over ranks), the batch time, the fused-cloud checksum (equal on
    all ranks), the pose error, and the hand-off to the mesher's input on rank 0 (mesh_reconstruction.py:22-37 reads a legacy
    cloud: D2H + binary PLY as io_formats writes it), which is outside batch_ms."""
    import numpy as np
    import torch
    import torch.distributed as dist
    Dm = r3d.distributed
    use_dist = dist.is_available() and dist.is_initialized()
    views = Dm.shard_views(n_views, rank, world)
    Q = r3d.pipeline.scaled_Q(np.load(os.path.join(ROOT, "tests", "golden", "jetson_stereo_8MP_stereo.npz"))["Q"], W / 960.0, unit=1e-3)
    poses = {v: (np.eye(4) if v == 0 else r3d.synth.rigid((0.2 * v, 1.0, 0.1 * (v % 3)), 0.25 + 0.05 * v,
                                                            (0.002 + 0.0005 * v, -0.0015, 0.001 * (v % 4)))) for v in range(n_views)}
    m = r3d.StereoSGBM_create(numDisparities=D, mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY, **C2_KW)
    m._ctx = ctx
    stream = torch.cuda.Stream()
    cap = 1 << 20
    out = {}
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)          # library kernels, RCCL and torch ops share one stream: ordered without events
        try:
            imgs = {}
            for v in views:
                L, R, _ = r3d.synth.stereo_pair(W, H, D, seed=20241008 + v)
                imgs[v] = (torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
            d_disp = torch.empty(W * H, dtype=torch.int16, device="cuda")
            bufs = {v: torch.empty((2, cap, 3), dtype=torch.float64, device="cuda") for v in views}
            best = None
            for rep in range(reps + 1):             # rep 0 warms arenas, RCCL channels and the matcher workspace
                if use_dist:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                local = {}
                for v in views:
                    local[v] = r3d.pipeline.view_to_cloud_tensors(imgs[v][0].data_ptr(), imgs[v][1].data_ptr(), d_disp.data_ptr(), W, H, Q, m,
                                                                  bufs[v], voxel=0.01, max_nn=30, max_depth=3.0,
                                                                  pose=np.linalg.inv(poses[v]))
                e1.record()
                tm = {}
                fused, Ts = r3d.pipeline.multi_view_fuse_tensors(local, n_views, threshold=0.02, mode=r3d.cloud_ops.GICP,
                                                                  max_iteration=30, ctx=ctx, timings=tm)
                torch.cuda.synchronize()
                if use_dist:
                    dist.barrier()
                    torch.cuda.synchronize()
                wall = 1e3 * (time.perf_counter() - t0)
                tm["view_ms"] = e0.elapsed_time(e1)
                tm["batch_ms"] = wall
                if rep and (best is None or wall < best["batch_ms"]):
                    best = dict(tm)
            stages = torch.tensor([best[k] for k in ("view_ms", "exchange_ms", "register_ms", "fuse_ms", "batch_ms")], dtype=torch.float64, device="cuda")
            chk = torch.stack([fused[0].sum(0), fused[1].sum(0)]).reshape(-1)                    # 6 numbers
            if use_dist:
                dist.all_reduce(stages, op=dist.ReduceOp.MAX)
                allc = [torch.empty_like(chk) for _ in range(world)]
                dist.all_gather(allc, chk)
                same = all(torch.equal(allc[0], c) for c in allc)
            else:
                same = True
            terr = max(float(np.abs(Ts[v] - poses[v]).max()) for v in range(n_views))
            st = stages.cpu().numpy()
            out = {"workload": f"C5: {n_views} views of 3264x2448 D=128, {len(views)} per rank on {world} rank(s): SGM -> cloud (voxel 0.01, "
                               "normals k30) resident in HBM -> all-gather-v (RCCL, device tensors) -> GICP of every view to view 0 -> "
                               "all-gather of the 4x4s -> fused cloud on every rank",
                   "n_views": n_views, "views_per_rank": len(views), "view_ms": round(float(st[0]), 3), "exchange_ms": round(float(st[1]), 3),
                   "register_ms": round(float(st[2]), 3), "fuse_ms": round(float(st[3]), 3), "batch_ms": round(float(st[4]), 3),
                   "views_per_s": round(1e3 * n_views / float(st[4]), 2), "fused_points": int(fused.shape[1]),
                   "fused_checksum": [float(x) for x in chk.cpu().numpy()], "checksum_equal_on_all_ranks": bool(same),
                   "pose_error_max_abs": terr, "collective": ("rccl all_gather_into_tensor on device tensors" if use_dist and (world > 1 or os.environ.get("R3D_FORCE_DIST"))
                                  else "single rank: no collective issued"),
                   "stage_times": "max over ranks of the best of %d repetitions; batch_ms is wall time between barriers" % reps}
            if rank == 0:                            # hand-off to the mesher's input (outside batch_ms)
                t0 = time.perf_counter()
                arr = fused.cpu().numpy()
                pc = r3d.PointCloud(arr[0], normals=arr[1])
                t1 = time.perf_counter()
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"r3d_c5_fused_{os.getpid()}.ply")
                r3d.io_formats.write_ply(path, pc.points, normals=pc.normals)
                t2 = time.perf_counter()
                out["handoff"] = {"download_ms": round(1e3 * (t1 - t0), 2), "write_ply_ms": round(1e3 * (t2 - t1), 2),
                                  "ply_bytes": os.path.getsize(path), "consumer": "mesh_reconstruction.py:22-37 (Poisson, CPU; out of scope)"}
                os.remove(path)
        finally:
            torch.cuda.synchronize()
            ctx.set_stream(None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gicp", action="store_true", help="skip the secondary metric (GICP iterations/s at 1M points)")
    ap.add_argument("--extras", action="store_true",
                    help="add two informational legs after the timed region: the same maps with three in flight "
                         "(batch entry point) and one depth2.py frame iteration (remap, both matchers, WLS filter, "
                         "normalize).  Off by default so that a rocprofv3 --stats summary of the default command "
                         "averages every SGM kernel over un-overlapped C2 launches only, like the `roofline` object")
    ap.add_argument("--no-torch", action="store_true", help="keep torch out of the process (system HIP runtime; implies --no-c5)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 leg (8-view batch: view chain -> RCCL all-gather-v -> registration)")
    ap.add_argument("--repeats", type=int, default=5, help="extra repetitions of the K-step timed region for ms_per_step min / median")
    ap.add_argument("--lanes", type=int, default=1,
                    help="maps in flight per GPU. 1 (default): strictly one map after the other, so that the per-kernel HIP-event "
                         "durations behind `roofline` are uncontended and agree with rocprofv3 --stats; 3: the K maps go through "
                         "r3d_sgbm_compute_batch_dev and overlap on the library's internal lanes (+14 %% maps/s on C2)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON, rank 0): libraries that print banners to file descriptor 1 (RCCL announces its
    # version there when the group comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.no_torch:                              # torch-free process on the system HIP runtime (e.g. under rocprofv3)
        os.environ["R3D_NO_TORCH_PRELOAD"] = "1"
        args.no_c5 = True
        if world > 1:
            raise SystemExit("--no-torch is a single-process option")
    r3d = importlib.import_module("3d_reconstruction_project_amd")
    ctx = r3d.Context(local_rank)                  # imports torch first when it is installed: one HIP runtime per process (_lib.py)
    dist = None
    if world > 1 or os.environ.get("R3D_FORCE_DIST"):      # R3D_FORCE_DIST: rehearse the multi-rank code path with one rank
        import torch
        import torch.distributed as dist
        # binds this rank to GPU LOCAL_RANK before any other GPU call and opens the RCCL group; the SGM context keeps its own
        # stream (the C5 leg switches it to the stream it shares with torch / RCCL)
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank that dies inside a collective must not leave the others waiting for the default 10 minutes
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=240))
    L, R, _ = r3d.synth.stereo_pair(W, H, D, seed=20241008 + rank)
    dL, dR = ctx.to_device(L), ctx.to_device(R)
    lanes = max(1, min(args.lanes, 3))
    dDs = [ctx.alloc(W * H * 2) for _ in range(lanes)]          # one output per lane (maps in flight never share one)
    dD = dDs[0]
    m = r3d.StereoSGBM_create(numDisparities=D, mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY, **C2_KW)
    m._ctx = ctx

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if lanes == 1:
        for _ in range(args.warmup):
            m.compute_device(dL, dR, W, H, W, dD)
    else:                                        # warm every lane (workspace allocation happens at a lane's first use)
        nw = max(args.warmup, lanes)
        m.compute_batch_device([dL] * nw, [dR] * nw, W, H, W, [dDs[i % lanes] for i in range(nw)])
    ctx.set_profiling(True)
    ctx.sgbm_profile()                         # reset accumulators
    e0, e1 = ctx.event(), ctx.event()          # created before the clock starts
    barrier()
    t0 = time.perf_counter()
    ctx.record(e0)
    if lanes == 1:
        for _ in range(args.steps):
            m.compute_device(dL, dR, W, H, W, dD)
    else:
        # the K steps are K independent maps: the library pipelines them over its internal lanes (own stream and
        # workspace each), so that the cost / hscan / vscan kernels of consecutive maps overlap
        m.compute_batch_device([dL] * args.steps, [dR] * args.steps, W, H, W, [dDs[i % lanes] for i in range(args.steps)])
    ctx.record(e1)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ctx.elapsed_ms(e0, e1)
    prof = ctx.sgbm_profile()                  # average per-kernel launch duration over the timed region (HIP events)
    # robustness: the same K-step region repeated (one map in flight, profiling events off); `value` stays the first region's
    rep_ms = []
    ctx.set_profiling(False)
    for _ in range(max(0, args.repeats)):
        barrier()
        tr = time.perf_counter()
        if lanes == 1:
            for _ in range(args.steps):
                m.compute_device(dL, dR, W, H, W, dD)
        else:
            m.compute_batch_device([dL] * args.steps, [dR] * args.steps, W, H, W, [dDs[i % lanes] for i in range(args.steps)])
        barrier()
        rep_ms.append(1e3 * (time.perf_counter() - tr) / args.steps)
    if dist is not None and rep_ms:
        t = torch.tensor(rep_ms, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_ms = [float(x) for x in t.cpu().numpy()]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary metric on every rank when N > 1 (weak scaling: one 1M-point cloud pair per GPU, no communication inside the
    # registration loop); rank 0 reports the sum of the per-rank rates
    gicp_multi = None
    if dist is not None and (world > 1 or os.environ.get("R3D_FORCE_DIST") == "gicp") and not args.no_gicp:
        ctx.set_profiling(False)
        gm = bench_gicp(r3d, ctx, cpu=False)
        t = torch.tensor([gm["value"], gm["ms_per_iteration"]], dtype=torch.float64, device="cuda")
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gicp_multi = dict(gm, value=round(float(tsum[0].item()), 2), ms_per_iteration=round(float(tmax[1].item()), 4),
                          per_gpu=round(float(tsum[0].item()) / world, 2), n_gpus=world,
                          note="sum over ranks of the per-rank rate; ms_per_iteration is the slowest rank's")
    c5 = None
    if not args.no_c5:
        ctx.set_profiling(False)
        try:
            c5 = bench_c5(r3d, ctx, rank, world)
        except Exception as e:  # noqa: BLE001  the headline line must still be printed; the failure is reported in its place
            import traceback
            traceback.print_exc()
            c5 = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        value = world * args.steps / elapsed
        # default build: the cost kernel runs slab by slab on a second stream underneath the forward phase of the horizontal scan
        # (bracket "cost+hscan_fwd"), the backward phase is its own launch ("hscan_bwd").  The horizontal scan as a whole
        # (both brackets) is what the roofline is quoted on: it still owns the one compulsory volume write.
        overlapped = "hscan_bwd" in prof
        if overlapped:
            prof = dict(prof)
            prof["hscan"] = prof["cost+hscan_fwd"] + prof["hscan_bwd"]
        single = {k: v for k, v in prof.items() if k not in ("cost+hscan_fwd", "hscan_bwd")} if overlapped else prof
        dom = max(single, key=single.get) if single else None
        roofline = None
        if dom:
            ach = ALG_BYTES.get(dom, 0) / (prof[dom] * 1e-3) / 1e9
            traffic = None
            traffic_src = None
            tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tp):
                # counters cannot be read from inside the run they describe (rocprofv3 --pmc serialises dispatches): the figure is
                # the last committed PMC pass, tagged with its file date so a reader can tell whether it belongs to this build
                traffic_src = "profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE pass, file dated %s)" % time.strftime(
                    "%Y-%m-%d", time.gmtime(os.path.getmtime(tp)))
                # HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE doubled: gfx950 tallies 128-B requests at
                # 64 B, MI355X_MICROARCH.md "HBM"; WRITE_SIZE as is; both in KiB), written by tools/pmc_summary.py
                with open(tp) as f:
                    t = json.load(f).get(KERNEL_OF.get(dom, dom))
                if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
                    traffic = round((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / 1e9, 3)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(ach * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_unit": "GB/launch", "traffic_source": traffic_src,
                        "kernel_ms": {k: round(v, 4) for k, v in prof.items()},
                        "note": ("kernel durations are HIP-event brackets on each lane's stream; with %d maps in flight they "
                                 "include the time a kernel shares the chip with other maps' kernels" % lanes) if lanes > 1 else
                                ("hscan = forward phase (4 column-slab launches of k_hscan2<PHASE 1>, each waiting for the cost kernel of its "
                                 "slab on a second stream, bracket cost+hscan_fwd) + backward phase (one k_hscan2<PHASE 2> launch, bracket "
                                 "hscan_bwd); the volume is still written once and read three times per map (C twice by hscan, C + sum by "
                                 "vscan): the 40 %% pipeline target stays NOT MET until it is touched twice instead of six times"
                                 if overlapped else
                                 "one map in flight, kernels back to back; by the counters the volume is written twice and read four times per "
                                 "map (cost writes C; hscan reads C twice and writes the sum; vscan reads C and the sum: 12.9 GB against "
                                 "B_sgm = 4.06 GB): the 40 % pipeline target is NOT MET, DESIGN.md section 7 has the traces that rule out "
                                 "overlapping these chain kernels inside one map"),
                        "pipeline_frac": round(ALG_BYTES["map"] * (args.steps / (dev_ms * 1e-3)) / HBM_PEAK, 4)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import sgbm_oracle as so            # checker timed as the CPU baseline ("port")
            p = so.make_params(numDisparities=D, **C2_KW)
            threads = min(4, os.cpu_count() or 1)             # OpenCV runs its 4 fixed stripes under parallel_for_
            n = 0
            tc = time.perf_counter()
            while n < 3 or (time.perf_counter() - tc < 8.0 and n < 40):
                so.compute(L, R, p, nthreads=threads)
                n += 1
            dt = time.perf_counter() - tc
            model = "unknown"
            try:
                with open("/proc/cpuinfo") as f:
                    model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
            except OSError:
                pass
            cpu = {"value": round(n / dt, 3), "unit": "disparity-maps/s", "cores": threads, "kind": "port",
                   "sample": f"{n} full 3264x2448 D=128 maps, C restatement of OpenCV StereoSGBM 3WAY "
                             f"(oracle/sgbm3way.c, -O3 AVX2), host has {os.cpu_count()} cpus ({model})"}
            try:                                             # the real reference path, if this box ever has it
                import cv2
            except ImportError:
                cv2 = None
            if cv2 is not None:
                cv2.setNumThreads(threads)
                ref = cv2.StereoSGBM_create(numDisparities=D, mode=cv2.STEREO_SGBM_MODE_SGBM_3WAY, **C2_KW)
                n = 0
                tc = time.perf_counter()
                while n < 3 or (time.perf_counter() - tc < 8.0 and n < 40):
                    ref.compute(L, R)
                    n += 1
                dt = time.perf_counter() - tc
                cpu = {"value": round(n / dt, 3), "unit": "disparity-maps/s", "cores": threads, "kind": "reference",
                       "sample": f"{n} full 3264x2448 D=128 maps, cv2 {cv2.__version__} StereoSGBM MODE_SGBM_3WAY, "
                                 f"host has {os.cpu_count()} cpus ({model})", "port": cpu}
        piped = None
        if world == 1 and lanes == 1 and args.extras:
            # informational second leg (not `value`): the same maps through the batch entry point, three in flight on
            # the library's lanes, so the cost / hscan / vscan kernels of consecutive maps overlap
            ctx.set_profiling(False)
            outs = [dD] + [ctx.alloc(W * H * 2) for _ in range(2)]
            nb = max(6, 3 * ((args.steps + 2) // 3))
            m.compute_batch_device([dL] * 3, [dR] * 3, W, H, W, outs)            # lanes 1-2 allocate at first use
            ctx.sync()
            tp0 = time.perf_counter()
            m.compute_batch_device([dL] * nb, [dR] * nb, W, H, W, [outs[i % 3] for i in range(nb)])
            ctx.sync()
            tp1 = time.perf_counter()
            piped = {"maps_in_flight": 3, "maps": nb, "value": round(nb / (tp1 - tp0), 2), "unit": "disparity-maps/s",
                     "ms_per_map": round(1e3 * (tp1 - tp0) / nb, 4),
                     "pipeline_frac": round(ALG_BYTES["map"] * nb / (tp1 - tp0) / HBM_PEAK, 4),
                     "entry_point": "r3d_sgbm_compute_batch_dev"}
        frame_loop = None
        if world == 1 and args.extras:
            ctx.set_profiling(False)
            frame_loop = bench_frame_loop(r3d, ctx, dL, dR)
        gicp = gicp_multi
        if world == 1 and gicp_multi is None and not args.no_gicp:
            gicp = bench_gicp(r3d, ctx, cpu=not args.no_cpu_baseline)
        out = {"metric": "disparity-maps/s @8MP d=128", "value": round(value, 2), "unit": "disparity-maps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "int16", "data": "synthetic",
               "config": {"workload": "C2: 3264x2448 rectified pair, numDisparities=128, blockSize=5, "
                                      "MODE_SGBM_3WAY (depth2.py params), one pair per GPU resident in HBM",
                          "parallelism": f"dp{world} (one view per GPU, no collective in the SGM step)",
                          "maps_in_flight_per_gpu": lanes},
               "ms_per_step_repeats": ({"n": len(rep_ms), "min": round(min(rep_ms), 4), "median": round(sorted(rep_ms)[len(rep_ms) // 2], 4),
                                        "max": round(max(rep_ms), 4)} if rep_ms else None),
               "roofline": roofline, "cpu_baseline": cpu, "pipelined": piped, "frame_loop": frame_loop, "secondary": gicp, "c5": c5}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
