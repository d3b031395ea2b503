"""Per-frame comparison of the scanning loops (config C4, frames 8..15) between the product (resident and host-model paths)
and the oracle loop: iterations, fitness, |T - T_oracle|, and the warm time of the resident loop.  Run on the GPU box:
python tools/gpu_diag_fuse.py [icp] [gicp]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
from oracle import cloud_oracle as co
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
intr = co.read_intrinsics(os.path.join(G, "camera_intrinsic.json"))
raw = [co.backproject(co.read_png16(os.path.join(G, f"output84/depth_{i:05d}.png")), intr)[0] for i in range(8, 16)]
flavours = [a for a in sys.argv[1:] if a in ("icp", "gicp")] or ["icp", "gicp"]
for fl in flavours:
    if fl == "icp":
        frames = [co.voxel_down_sample(f, 0.01) for f in raw]
        feed = [r3d.PointCloud(f) for f in frames]
    else:
        frames = []
        for f in raw:
            p = co.voxel_down_sample_tensor(f, 0.01)
            frames.append((p, co.estimate_normals_hybrid(p, 0.05, 30)))
        feed = [r3d.PointCloud(p, normals=n) for p, n in frames]
    wl, gl, hl = [], [], []
    t0 = time.perf_counter(); want = co.fuse_loop(frames, fl, log=wl); t_or = time.perf_counter() - t0
    got = r3d.pipeline.fuse(feed, flavour=fl, log=gl)
    host = r3d.pipeline.fuse(feed, flavour=fl, resident=False, log=hl)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); r3d.pipeline.fuse(feed, flavour=fl); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); r3d.pipeline.fuse(feed, flavour=fl, resident=False); t_host = time.perf_counter() - t0
    rows = []
    for i, (w, g, h) in enumerate(zip(wl, gl, hl)):
        rows.append({"frame": 9 + i, "it": [w["iterations"], g["iterations"], h["iterations"]], "fit_oracle": round(w["fitness"], 6),
                     "dfit": g["fitness"] - w["fitness"], "drmse": g["inlier_rmse"] - w["inlier_rmse"],
                     "dT_resident": float(np.abs(g["T"] - w["T"]).max()), "dT_host": float(np.abs(h["T"] - w["T"]).max()),
                     "loop_ms": round(g["loop_ms"], 3), "setup_ms": round(g["setup_ms"], 3)})
    print(json.dumps({"flavour": fl, "points": len(got.points), "resident_equals_host": bool(np.array_equal(got.points, host.points)),
                      "max_dp_vs_oracle": float(np.abs(got.points - want[0]).max()), "fuse_ms_resident": round(1e3 * min(ts), 2),
                      "fuse_ms_host_model": round(1e3 * t_host, 2), "oracle_s": round(t_or, 1), "frames": rows}), flush=True)
