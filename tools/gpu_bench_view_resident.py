"""Times the device-resident C5 view chain (r3d_sgbm_compute_dev -> r3d_disparity_to_cloud_resident) alone, 10 views back to back,
inputs and outputs in HBM.  Under rocprofv3 --kernel-trace --stats it gives the per-kernel split of the non-SGM part.  GPU box."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
W, H, D = 3264, 2448, 128
ctx = r3d.default_context(0)
L, R, _ = r3d.synth.stereo_pair(W, H, D, seed=3)
Q = r3d.pipeline.scaled_Q(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "jetson_stereo_8MP_stereo.npz"))["Q"], W / 960.0, unit=1e-3)
m = r3d.reference_matcher(numDisparities=D, blockSize=5)
m._ctx = ctx
dL, dR, dD = ctx.to_device(L), ctx.to_device(R), ctx.alloc(W * H * 2)
cap = 1 << 20
dP, dN = ctx.alloc(cap * 24), ctx.alloc(cap * 24)
def view():
    m.compute_device(dL, dR, W, H, W, dD)
    return r3d.cloud_ops.disparity_to_cloud_resident(dD, W, H, Q, dP, dN, cap, 0, 3.0, None, 0.01, 0.02, 30, ctx=ctx)
n = view(); ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    n = view()
ctx.sync()
print(json.dumps({"ms_per_view": round(1e2 * (time.perf_counter() - t0), 3), "points": n}))
