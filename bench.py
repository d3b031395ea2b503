#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: disparity-maps/s at 8 MP (3264x2448), 128 disparities.

One "step" = one full StereoSGBM(3-way).compute over one synthetic rectified pair that is already resident in
HBM (one pair per rank; weak scaling: every rank owns one view, no data-path collective inside the SGM step).
Prints ONE JSON line on rank 0.  N > 1, either form:
    python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself as fresh child processes: launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment)
Rank -> GPU -> stream -> RCCL group binding lives in ONE place: 3d_reconstruction_project_amd/distributed.init().
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Eight hardware queues instead of HIP's four: the pipelined legs keep three SGM lanes, the context stream and the cloud contexts'
# streams busy at once (INTEGRATION.md "Recommended environment"; +8 % on `pipelined`).  The runtime reads it at the first GPU call
# of the process, so the application sets it -- here, before anything touches the GPU; the package itself never edits the environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

W, H, D = 3264, 2448, 128                     # BASELINE config C2
C2_KW = dict(minDisparity=0, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, uniquenessRatio=15,
             speckleWindowSize=0, speckleRange=2, preFilterCap=63)      # Calib_depth/depth2.py:139-158
HBM_PEAK = 8.0e12                             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MIXED_STREAM_TBPS = 5.0                       # measured plateau of a 1:1 read + write stream on this chip (tools/gpu_streambench_waves.py)
C5_WATCHDOG_S = 200                           # rank 0 prints the headline and the process exits if the C5 leg has not returned by then
W1 = W - D
# algorithmic bytes: whole map = 4*W*H + 4*W*H*D with int16 costs; per kernel of the current split:
ALG_BYTES = {
    "map": 4 * W * H + 4 * W1 * H * D,
    "hscan": 2 * W1 * H * D,                  # the one compulsory volume WRITE (L_left + L_right)
    "hscan_bwd": 2 * W1 * H * D,              # same write, issued by the backward-phase launch when the phases are separate
    "vscan_wta": 2 * W1 * H * D + 4 * W * H,  # the one compulsory volume READ + disparity/cost out
    "cost": 2 * W * H,                        # reads both images; the cost volume itself is not algorithmic
    "cost_fwd": 2 * W * H,                    # v5: cost along rows + forward chain in one kernel (images in; C out is not algorithmic)
}
KERNEL_OF = {"cost": "k_cost2", "cost_fwd": "k_cost_fwd", "hscan": "k_hscan2", "hscan_bwd": "k_hscan2", "vscan_wta": "k_vscan2",
             "prefilter": "k_prefilter", "lrcheck": "k_lrcheck", "median3": "k_median3"}


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
                            source_normals=sn, target_normals=tn, ctx=ctx) for _ in range(5)]
    by_loop = sorted(runs, key=lambda r: r["loop_ms"])
    res = by_loop[len(by_loop) // 2]                         # median of 5 repetitions; the spread is reported (loop_ms_repeats)
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
           # a queue stall on the box (10-60 ms once in a few dozen loops, profiles/r03_stall_device_clock.log) must show in the
           # record instead of vanishing in a median
           "loop_ms_repeats": {"n": len(by_loop), "min": round(by_loop[0]["loop_ms"], 3), "median": round(res["loop_ms"], 3),
                               "max": round(by_loop[-1]["loop_ms"], 3)},
           "stall_suspect": bool(by_loop[-1]["loop_ms"] > 3.0 * res["loop_ms"]),
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


def cpu_baseline_leg(L, R):
    """The checker timed as the CPU baseline ("port"): oracle/sgbm3way.c on 4 threads (OpenCV runs its 4 fixed stripes under
    parallel_for_), a bounded sample of full C2 maps; cv2's StereoSGBM ("reference") as well if this box ever has it."""
    from oracle import sgbm_oracle as so
    p = so.make_params(numDisparities=D, **C2_KW)
    threads = min(4, os.cpu_count() or 1)
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
    return cpu


def bench_c5(r3d, ctx, rank, world, n_views=8, reps=3):
    """BASELINE config C5: an 8-view 8 MP batch, views dealt round-robin to the ranks (one view per GPU at N = 8; all eight on
    the one GPU at N = 1, so c5.batch_ms at N = 1 vs N = 8 is the strong-scaling figure north_star asks for).  Per rank and view:
    SGM -> reprojection -> voxel 0.01 -> normals, device-resident; then ONE all-gather-v of the clouds over RCCL (device
    tensors, no host staging), every owned view registered to view 0 with GICP, all-gather of the 4x4s, every view transformed
    into view 0's frame.  A rank that owns several views runs them through pipeline.views_to_cloud_tensors: three SGM maps in
    flight on the library's lanes, the cloud stages of view i underneath the SGM kernels of the later views (second Context),
    i.e. the N = 1 figure is the best one GPU can do, not eight views strictly one after the other.
    View v shows the same synthetic scene (own texture / noise seed) displaced by a known pose, so the registration has a known
    answer.  Reported: per-stage ms (max over ranks), the batch time, the fused-cloud checksum (equal on all ranks), the pose
    error, and the hand-off to the mesher's input on rank 0 (mesh_reconstruction.py:22-37 reads a legacy cloud: D2H + binary PLY
    as io_formats writes it), which is outside batch_ms.
    Failure handling: every stage that can fail on ONE rank (the view chain, a registration) is followed by an agreement
    (distributed.agree / the flag that travels with the transforms), so all ranks leave the leg together with the same error
    instead of the healthy ones waiting inside a collective."""
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
    cap = 1 << 20
    out = {}
    cloud_ctx = None
    with Dm.shared_stream(ctx):                      # library kernels, RCCL and torch ops share one stream: ordered without events
        try:
            err = True
            try:
                imgs = {}
                for v in views:
                    L, R, _ = r3d.synth.stereo_pair(W, H, D, seed=20241008 + v)
                    imgs[v] = (torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
                d_disps = [torch.empty(W * H, dtype=torch.int16, device="cuda") for _ in views]
                bufs = {v: torch.empty((2, cap, 3), dtype=torch.float64, device="cuda") for v in views}
                if len(views) > 1:
                    ncc = max(1, min(int(os.environ.get("R3D_C5_CLOUD_CTX", "3")), 4))
                    cloud_ctx = [r3d.Context(ctx.device) for _ in range(ncc)]
            except Exception as e:  # noqa: BLE001
                err = e
            Dm.agree(err, "C5 input set-up")
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
                err = True
                try:
                    if len(views) > 1 and os.environ.get("R3D_C5_SEQ") != "1":
                        got = r3d.pipeline.views_to_cloud_tensors([(imgs[v][0].data_ptr(), imgs[v][1].data_ptr()) for v in views],
                                                                  [d.data_ptr() for d in d_disps], W, H, Q, m, [bufs[v] for v in views],
                                                                  cloud_ctx, voxel=0.01, max_nn=30, max_depth=3.0,
                                                                  poses=[np.linalg.inv(poses[v]) for v in views])
                        local = dict(zip(views, got))
                    else:
                        for v, d_disp in zip(views, d_disps):
                            local[v] = r3d.pipeline.view_to_cloud_tensors(imgs[v][0].data_ptr(), imgs[v][1].data_ptr(), d_disp.data_ptr(), W, H, Q,
                                                                          m, bufs[v], voxel=0.01, max_nn=30, max_depth=3.0,
                                                                          pose=np.linalg.inv(poses[v]))
                except Exception as e:  # noqa: BLE001
                    err = e
                e1.record()
                Dm.agree(err, "C5 view chain")
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
                   "n_views": n_views, "views_per_rank": len(views),
                   "view_chain": ("3 SGM maps in flight (r3d_sgbm_compute_batch_events_dev), cloud stages of view i on %d further context(s) "
                                  "(own stream, arena and host thread each) underneath the SGM kernels of views i+1.." % len(cloud_ctx)
                                  if len(views) > 1 else "one view per rank: SGM then cloud stages, one stream"),
                   "view_ms": round(float(st[0]), 3), "exchange_ms": round(float(st[1]), 3),
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
            for c in (cloud_ctx or []):
                c.close()
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start N ranks as FRESH child processes
    -- torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 -- BEFORE this process has made any GPU call (it
    never makes one: a process that has initialised the GPU must not be replaced or forked on this pool).  Rank 0's single JSON
    line is relayed on stdout, everything else goes to stderr, and a non-zero exit of any rank becomes ours."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    # The children inherit this process's environment unchanged (plus OMP_NUM_THREADS; GPU_MAX_HW_QUEUES was set at import above).
    # HSA_ENABLE_IPC_MODE_LEGACY is NOT defaulted here any more: the pool that runs this bench exports it (=0: its host driver
    # supports dmabuf IPC only, which RCCL's intra-node transport needs), other hosts may need the opposite; say so once if absent.
    env = dict(os.environ)
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in env:
        print("bench.py: HSA_ENABLE_IPC_MODE_LEGACY is not set; if RCCL fails with 'hipIpcGetMemHandle: invalid argument' export "
              "HSA_ENABLE_IPC_MODE_LEGACY=0 (hosts whose driver offers dmabuf IPC only)", file=sys.stderr, flush=True)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                                  # children keep stdout for the JSON line only; relay the last one
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def dry_control(args, json_fd):
    """--dry-control: the control flow of a multi-rank run without touching a GPU or the HIP library (CPU test of the launcher):
    rank binding from the environment, process group on the chosen backend, barrier, max-over-ranks reduction, one JSON line from
    rank 0.  --dry-fail-rank R makes rank R exit non-zero after the group is up, to prove that the failure propagates."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(args.backend or "gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    dist.barrier()
    if args.dry_fail_rank == rank:
        os._exit(7)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    who = [None] * world
    dist.all_gather_object(who, {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "ppid": os.getppid()})
    if rank == 0:
        out = {"metric": "disparity-maps/s @8MP d=128", "value": None, "unit": "disparity-maps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "dry_control": True, "backend": dist.get_backend(), "max_over_ranks_s": float(t.item()),
               "ranks": who}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gicp", action="store_true", help="skip the secondary metric (GICP iterations/s at 1M points)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two informational legs that follow the timed region: the same maps with three in flight "
                         "(batch entry point, `pipelined`) and one depth2.py frame iteration (`frame_loop`: remap, both matchers, "
                         "WLS filter, normalize).  Use it for a rocprofv3 --stats pass whose per-kernel averages must cover "
                         "un-overlapped C2 launches only, like the `roofline` object")
    ap.add_argument("--extras", action="store_true", help="accepted for compatibility: the extras are on by default")
    ap.add_argument("--no-torch", action="store_true", help="keep torch out of the process (system HIP runtime; implies --no-c5)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 leg (8-view batch: view chain -> RCCL all-gather-v -> registration)")
    ap.add_argument("--repeats", type=int, default=5, help="extra repetitions of the K-step timed region for ms_per_step min / median")
    ap.add_argument("--lanes", type=int, default=1,
                    help="maps in flight per GPU. 1 (default): strictly one map after the other, so that the per-kernel HIP-event "
                         "durations behind `roofline` are uncontended and agree with rocprofv3 --stats; 3: the K maps go through "
                         "r3d_sgbm_compute_batch_dev and overlap on the library's internal lanes (+14 %% maps/s on C2)")
    ap.add_argument("--backend", default=None, help="process-group backend (default: nccl = RCCL; gloo only with --dry-control)")
    ap.add_argument("--dry-control", action="store_true", help="launcher / rendezvous / reduction control flow only: no GPU, no HIP library")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="with --dry-control: this rank exits non-zero (failure propagation test)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing GPU-related has been imported or called in this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line (the JSON, rank 0): libraries that print banners to file descriptor 1 (RCCL announces its
    # version there when the group comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.dry_control:
        return dry_control(args, json_fd)
    if args.backend not in (None, "nccl"):
        raise SystemExit("the measured path runs on RCCL (backend nccl); gloo is for --dry-control only")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.no_torch:                              # torch-free process on the system HIP runtime (e.g. under rocprofv3)
        os.environ["R3D_NO_TORCH_PRELOAD"] = "1"
        args.no_c5 = True
        if world > 1:
            raise SystemExit("--no-torch is a single-process option")
    r3d = importlib.import_module("3d_reconstruction_project_amd")
    dist = None
    if args.no_torch:
        ctx = r3d.Context(local_rank)
    else:
        # ONE place binds rank -> GPU -> stream -> process group (distributed.init): torch.cuda.set_device(LOCAL_RANK) before any
        # other GPU call, a library context on that device sharing one stream with torch, RCCL group when world > 1 (or
        # R3D_FORCE_DIST, the one-rank rehearsal).  240 s collective timeout: a rank that dies must not hold the others for 10 min
        import torch
        rank, world, ctx = r3d.distributed.init("nccl", ctx=r3d.Context(local_rank), timeout_s=240)
        import torch.distributed as tdist
        dist = tdist if tdist.is_initialized() else None
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

    # informational second leg (not `value`), on every rank: the same maps through the batch entry point, three in flight on the
    # library's lanes, so the cost / hscan / vscan kernels of consecutive maps overlap.  Whole-job figure = sum over ranks.
    piped = None
    if lanes == 1 and not args.no_extras:
        nl = max(1, min(int(os.environ.get("R3D_SGM_LANES", "3")), 6))           # the library reads the same variable
        outs = [dD] + [ctx.alloc(W * H * 2) for _ in range(nl - 1)]
        nb = max(2 * nl, nl * ((args.steps + nl - 1) // nl))
        m.compute_batch_device([dL] * nl, [dR] * nl, W, H, W, outs)            # the other lanes allocate at first use
        barrier()
        tp0 = time.perf_counter()
        m.compute_batch_device([dL] * nb, [dR] * nb, W, H, W, [outs[i % nl] for i in range(nb)])
        barrier()
        pdt = time.perf_counter() - tp0
        if dist is not None:
            t = torch.tensor([pdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pdt = float(t.item())
        piped = {"maps_in_flight": nl, "maps": nb * world, "value": round(world * nb / pdt, 2), "unit": "disparity-maps/s",
                 "ms_per_map": round(1e3 * pdt / nb, 4),
                 "pipeline_frac": round(ALG_BYTES["map"] * nb / pdt / HBM_PEAK, 4),
                 "entry_point": "r3d_sgbm_compute_batch_dev", "n_gpus": world}

    # informational third leg (never `value`; SURVEY.md 8d asks for it beside the device-resident figure): the same maps through the
    # HOST-buffer entry point r3d_sgbm_compute with pageable numpy arrays: 2 x 8 MB up and 16 MB down per map included
    host_api = None
    if lanes == 1 and not args.no_extras and rank == 0:
        m.compute(L, R)                                         # staging buffers allocate at first use
        nh = max(4, min(args.steps, 10))
        th0 = time.perf_counter()
        for _ in range(nh):
            m.compute(L, R)
        hdt = time.perf_counter() - th0
        host_api = {"value": round(nh / hdt, 2), "unit": "disparity-maps/s", "ms_per_map": round(1e3 * hdt / nh, 4), "maps": nh,
                    "entry_point": "r3d_sgbm_compute (pageable numpy arrays in and out: H2D 2 x %.1f MB + D2H %.1f MB per map over PCIe, "
                                   "synchronous)" % (W * H / 1e6, 2 * W * H / 1e6), "n_gpus": 1}

    # secondary metric on every rank when N > 1 (weak scaling: one 1M-point cloud pair per GPU, no communication inside the
    # registration loop); rank 0 reports the sum of the per-rank rates
    gicp_multi = None
    if dist is not None and (world > 1 or os.environ.get("R3D_FORCE_DIST") == "gicp") and not args.no_gicp:
        err = True
        try:
            gm = bench_gicp(r3d, ctx, cpu=False)
        except (Exception, SystemExit) as e:  # noqa: BLE001
            err = e
        r3d.distributed.agree(err, "GICP leg")          # every rank raises together if one failed (no rank left in a collective)
        t = torch.tensor([gm["value"], gm["ms_per_iteration"]], dtype=torch.float64, device="cuda")
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gicp_multi = dict(gm, value=round(float(tsum[0].item()), 2), ms_per_iteration=round(float(tmax[1].item()), 4),
                          per_gpu=round(float(tsum[0].item()) / world, 2), n_gpus=world,
                          note="sum over ranks of the per-rank rate; ms_per_iteration is the slowest rank's")

    def headline(c5, gicp, cpu, frame_loop):
        value = world * args.steps / elapsed
        prof_ = dict(prof)
        # R3D_SGM_OVERLAP builds: the cost kernel runs slab by slab underneath the forward phase (bracket "cost+hscan_fwd"), the
        # backward phase is its own launch ("hscan_bwd"); the horizontal scan as a whole is what the roofline is quoted on
        overlapped = "hscan_bwd" in prof_ and "cost+hscan_fwd" in prof_
        if overlapped:
            prof_["hscan"] = prof_["cost+hscan_fwd"] + prof_["hscan_bwd"]
        single = {k: v for k, v in prof_.items() if k not in ("cost+hscan_fwd", "hscan_bwd")} if overlapped else prof_
        dom = max(single, key=single.get) if single else None
        roofline = None
        if dom:
            ach = ALG_BYTES.get(dom, 0) / (prof_[dom] * 1e-3) / 1e9
            traffic = traffic_src = total_traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tp):
                # counters cannot be read from inside the run they describe (rocprofv3 --pmc serialises dispatches): the figure is
                # the last committed PMC pass, tagged with its file date so a reader can tell whether it belongs to this build.
                # HBM bytes per launch (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM";
                # WRITE_SIZE as is; both in KiB), written by tools/pmc_summary.py
                traffic_src = "profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE pass, file dated %s)" % time.strftime(
                    "%Y-%m-%d", time.gmtime(os.path.getmtime(tp)))
                with open(tp) as f:
                    tj = json.load(f)
                t = tj.get(KERNEL_OF.get(dom, dom))
                if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
                    traffic = round((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / 1e9, 3)
                tot = [(2 * tj[k]["FETCH_SIZE"] + tj[k]["WRITE_SIZE"]) * 1024 / 1e9 for k in set(KERNEL_OF.values())
                       if k in tj and "FETCH_SIZE" in tj[k] and "WRITE_SIZE" in tj[k]]
                total_traffic = round(sum(tot), 3) if tot else None
            map_gb = ALG_BYTES["map"] / 1e9
            floor_gb = total_traffic or 12.9
            floor_ms = floor_gb / MIXED_STREAM_TBPS
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(ach * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_unit": "GB/launch", "traffic_source": traffic_src,
                        "kernel_ms": {k: round(v, 4) for k, v in prof_.items()},
                        "note": ("kernel durations are HIP-event brackets on each lane's stream; with %d maps in flight they "
                                 "include the time a kernel shares the chip with other maps' kernels" % lanes) if lanes > 1 else
                                ("one map in flight, kernels back to back; by the counters the volume is written twice and read four times per "
                                 "map (cost writes C; hscan reads C twice and writes the sum; vscan reads C and the sum: %.1f GB against "
                                 "B_sgm = %.2f GB): the 40 %% pipeline target is NOT MET" % (floor_gb, map_gb)),
                        "pipeline_frac": round(ALG_BYTES["map"] * (args.steps / (dev_ms * 1e-3)) / HBM_PEAK, 4),
                        # the ceiling of THIS decomposition, stated once: three separate passes over a materialised
                        # cost volume move floor_gb per map; a mixed read + write stream sustains ~5 TB/s on this chip
                        # (tools/gpu_streambench_waves.py), so the decomposition's own floor is floor_ms
                        "floor_note": {"decomposition": "cost -> hscan (fwd + bwd) -> vscan over a materialised int16 volume",
                                       "traffic_gb_per_map": round(floor_gb, 2), "mixed_stream_tb_s": MIXED_STREAM_TBPS,
                                       "floor_ms_per_map": round(floor_ms, 3), "floor_maps_per_s": round(1e3 / floor_ms, 1),
                                       "achieved_frac_of_floor": round(floor_ms / (dev_ms / args.steps), 4),
                                       "frac_of_8TBs_on_B_sgm": round(ALG_BYTES["map"] * (args.steps / (dev_ms * 1e-3)) / HBM_PEAK, 4)}}
        return {"metric": "disparity-maps/s @8MP d=128", "value": round(value, 2), "unit": "disparity-maps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "int16", "data": "synthetic",
                "config": {"workload": "C2: 3264x2448 rectified pair, numDisparities=128, blockSize=5, "
                                       "MODE_SGBM_3WAY (depth2.py params), one pair per GPU resident in HBM",
                           "parallelism": f"dp{world} (one view per GPU, no collective in the SGM step)",
                           "maps_in_flight_per_gpu": lanes},
                "ms_per_step_repeats": ({"n": len(rep_ms), "min": round(min(rep_ms), 4), "median": round(sorted(rep_ms)[len(rep_ms) // 2], 4),
                                         "max": round(max(rep_ms), 4)} if rep_ms else None),
                "roofline": roofline, "cpu_baseline": cpu, "pipelined": piped, "host_api": host_api, "frame_loop": frame_loop, "secondary": gicp, "c5": c5}

    # single-rank legs that need no collective (rank 0 only): CPU baseline, frame loop, GICP at N = 1
    cpu = frame_loop = None
    gicp = gicp_multi
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_leg(L, R)
        if world == 1 and not args.no_extras:
            frame_loop = bench_frame_loop(r3d, ctx, dL, dR)
        if world == 1 and gicp_multi is None and not args.no_gicp:
            gicp = bench_gicp(r3d, ctx, cpu=not args.no_cpu_baseline)

    # C5 is the one leg with collectives in its data path.  The headline is complete BEFORE it starts; if the leg does not come
    # back (a peer died inside a collective) rank 0's watchdog still prints the line, with the failure in `c5`, and exits non-zero
    c5 = None
    if not args.no_c5:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(C5_WATCHDOG_S):
                if rank == 0:
                    os.write(json_fd, (json.dumps(headline({"error": f"C5 leg did not return within {C5_WATCHDOG_S} s (a rank is stuck in or "
                                                                     "never reached a collective); headline measured before the leg"},
                                                           gicp, cpu, frame_loop)) + "\n").encode())
                os._exit(4)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            c5 = bench_c5(r3d, ctx, rank, world)
        except Exception as e:  # noqa: BLE001  every rank leaves bench_c5 together (agreement after each rank-local stage)
            import traceback
            traceback.print_exc()
            c5 = {"error": f"{type(e).__name__}: {e}"}
        done.set()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(headline(c5, gicp, cpu, frame_loop)) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if c5 is not None and "error" in c5:
        sys.exit(5)


if __name__ == "__main__":
    main()
