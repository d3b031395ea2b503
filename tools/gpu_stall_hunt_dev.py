"""Like gpu_stall_hunt.py but on clouds that are resident in HBM (r3d_icp_dev: no uploads), 1 M points, many calls: the sporadic
~50 ms registration loop shows up about once in a few dozen calls.  Prints per-call loop_ms outliers; run it under
rocprofv3 --hip-trace --kernel-trace and feed the output directory to tools/stall_report.py."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
co = r3d.cloud_ops
ctx = r3d.default_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
s, t, _ = r3d.synth.cloud_pair(1_000_000)
s, t = s.astype(np.float64), t.astype(np.float64)
tn = co.estimate_normals(t, None, 20)
d = [ctx.to_device(a) for a in (s, t, tn)]
rows = []
for i in range(n):
    t0 = time.perf_counter()
    r = co.registration_device(d[0], len(s), d[1], len(t), 0.02, mode=co.P2PLANE, max_iteration=20, relative_fitness=-1, relative_rmse=-1,
                               d_target_normals=d[2], ctx=ctx)
    rows.append((1e3 * (time.perf_counter() - t0), r["setup_ms"], r["loop_ms"]))
a = np.array(rows)
med = np.median(a, 0)
print(json.dumps({"calls": n, "median_ms": {"call": round(float(med[0]), 3), "setup": round(float(med[1]), 3), "loop": round(float(med[2]), 3)},
                  "outliers": [{"call": int(i), "ms": round(float(a[i, 0]), 2), "setup_ms": round(float(a[i, 1]), 2), "loop_ms": round(float(a[i, 2]), 2)}
                               for i in np.nonzero(a[:, 0] > 3 * med[0])[0]]}))
