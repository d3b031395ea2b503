"""Are the sporadic 10-50 ms registration loops a property of a YOUNG process?  The 12-call sequence of tools/gpu_bench_gicp.py
(three modes x four repetitions, host arrays, 1 M points) repeated several times inside ONE process: stalls per pass.
R3D_ICP_DEBUG=1 prints whether the device or the completion was late.  Usage (GPU box): python tools/gpu_stall_young_process_probe.py [passes]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
co = r3d.cloud_ops
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t_start = time.perf_counter()
src, tgt, T_star = r3d.synth.cloud_pair(1_000_000)
src, tgt = src.astype(np.float64), tgt.astype(np.float64)
sn = co.estimate_normals(src, None, 20); tn = co.estimate_normals(tgt, None, 20)
out = []
for p in range(passes):
    rows = []
    for mode, name in ((co.GICP, "gicp"), (co.P2PLANE, "p2plane"), (co.P2P, "p2p")):
        for rep in range(4):
            res = co.registration(src, tgt, 0.02, mode=mode, max_iteration=20, relative_fitness=-1, relative_rmse=-1, source_normals=sn, target_normals=tn)
            rows.append((name, rep, round(res["loop_ms"], 2), round(res["setup_ms"], 2)))
    slow = [r for r in rows if r[2] > 8.0]
    out.append({"pass": p, "process_age_s_at_end": round(time.perf_counter() - t_start, 1), "slow_loops": slow, "slow_setups": [r for r in rows if r[3] > 8.0]})
    print(json.dumps(out[-1]), flush=True)
