"""Does the wave count per CU decide the duration of the SGM chain kernels?  k_hscan2 runs one wave per 4 rows, k_vscan2 one wave
per 16 columns and stripe: at C2 that is 612 and 784 single-wave workgroups on 256 CUs, i.e. 100 CUs carry three hscan waves (the
others two) and 16 CUs carry four vscan waves (the others three).  If a CU's memory pipeline is the limit, the kernel lasts as
long as its most loaded CU.  Probe: image sizes whose wave counts are exact multiples of 256 against the C2 size.
Prints per-kernel milliseconds (HIP events, average of 10 maps).  GPU box."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
ctx = r3d.default_context(0)
D = 128
kw = dict(minDisparity=0, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=15, speckleWindowSize=0, speckleRange=2, preFilterCap=63)
out = []
for W, H in [(3264, 2448), (3264, 2048), (3264, 3072), (3200, 2448), (3200, 2048), (3264, 2448)]:
    rng = np.random.default_rng(1)
    L = rng.integers(0, 255, (H, W)).astype(np.uint8); R = np.roll(L, -40, 1)
    dL, dR, dD = ctx.to_device(L), ctx.to_device(R), ctx.alloc(W * H * 2)
    m = r3d.StereoSGBM_create(numDisparities=D, mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY, **kw); m._ctx = ctx
    for _ in range(3):
        m.compute_device(dL, dR, W, H, W, dD)
    ctx.sync(); ctx.set_profiling(True); ctx.sgbm_profile()
    for _ in range(10):
        m.compute_device(dL, dR, W, H, W, dD)
    ctx.sync(); prof = ctx.sgbm_profile(); ctx.set_profiling(False)
    for p in (dL, dR, dD):
        ctx.free(p)
    hw, vw = (H + 3) // 4, ((W - D + 15) // 16) * 4
    row = {"W": W, "H": H, "hscan_waves": hw, "hscan_waves_per_cu": round(hw / 256, 2), "vscan_waves": vw, "vscan_waves_per_cu": round(vw / 256, 2),
           "ms": {k: round(v, 4) for k, v in prof.items()},
           "hscan_us_per_Mcell": round(1e3 * prof["hscan"] / ((W - D) * H * 1e-6), 3), "vscan_us_per_Mcell": round(1e3 * prof["vscan_wta"] / ((W - D) * H * 1e-6), 3),
           "cost_us_per_Mcell": round(1e3 * prof["cost"] / ((W - D) * H * 1e-6), 3)}
    out.append(row)
    print(json.dumps(row), flush=True)
