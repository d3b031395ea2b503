"""GPU-box diagnostic: stage-by-stage comparison of the HIP SGBM against the oracle, plus 8 MP timing.
Writes gpurun_out/diag_sgm.txt.  Usage: python tools/gpu_diag_sgm.py [--big]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_project_amd")
from oracle import sgbm_oracle as so  # noqa: E402
from tests import sgbm_numpy_ref as ref  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "diag_sgm.txt"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


KW = dict(minDisparity=0, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=15, speckleWindowSize=0,
          speckleRange=2, preFilterCap=63)


def mism(name, a, b):
    a = np.asarray(a).astype(np.int64)
    b = np.asarray(b).astype(np.int64)
    bad = a != b
    log(f"  {name}: mismatches {int(bad.sum())} / {bad.size}")
    if bad.any():
        idx = np.argwhere(bad)
        log("    first:", idx[:6].tolist(), "got", a[bad][:6].tolist(), "want", b[bad][:6].tolist())
        for ax in range(bad.ndim):
            other = tuple(i for i in range(bad.ndim) if i != ax)
            prof = bad.sum(axis=other)
            nz = np.nonzero(prof)[0]
            log(f"    axis{ax}: bad idx range [{nz.min()}, {nz.max()}], count {len(nz)} of {bad.shape[ax]}")
    return int(bad.sum())


def stage_case(W, H, D, seed):
    log(f"case {W}x{H} D={D}")
    L, R, _ = r3d.synth.stereo_pair(W, H, D, seed=seed)
    m = r3d.StereoSGBM_create(numDisparities=D, mode=2, **KW)
    got = m.compute(L, R)
    want_h = os.environ.get("R3D_SGM_IMPL", "v2") != "v3"
    st = m.debug_fetch(want_cost=True, want_hsum=want_h, want_raw=True)
    p = so.make_params(numDisparities=D, **KW)
    want, want_raw = so.compute(L, R, p, nthreads=8, return_raw=True)
    C = so.cost_rows(L, R, p, 0, 0, H)
    n = mism("cost", st["cost"], C)
    if n == 0 and want_h and W * H * D <= 96 * 64 * 32:
        hs = np.zeros_like(C, dtype=np.int64)
        W1 = C.shape[1]
        for y in range(H):
            prev, pm = np.zeros(D, np.int64), 0
            Ll = np.zeros((W1, D), np.int64)
            for x in range(W1):
                prev, pm = ref.step(C[y, x].astype(np.int64), prev, pm, 600, 2400)
                Ll[x] = prev
            prev, pm = np.zeros(D, np.int64), 0
            for x in range(W1 - 1, -1, -1):
                prev, pm = ref.step(C[y, x].astype(np.int64), prev, pm, 600, 2400)
                hs[y, x] = Ll[x] + prev
        mism("hsum", st["hsum"], hs)
    mism("raw", st["raw"], want_raw)
    mism("final", got, want)


ctx = r3d.default_context(0)
try:
    ctx.selftest()
    log("selftest OK")
except Exception as e:  # noqa: BLE001
    log("selftest FAILED:", e)

stage_case(96, 64, 32, 0)
stage_case(200, 90, 16, 1)
stage_case(333, 121, 64, 2)
stage_case(500, 203, 128, 5)
stage_case(700, 150, 256, 6)

if "--big" in sys.argv:
    W, H, D = 3264, 2448, 128
    t0 = time.time()
    L, R, _ = r3d.synth.stereo_pair(W, H, D)
    log(f"generated 8MP pair in {time.time() - t0:.1f}s")
    m = r3d.StereoSGBM_create(numDisparities=D, mode=2, **KW)
    dL, dR = ctx.to_device(L), ctx.to_device(R)
    dD = ctx.alloc(W * H * 2)
    ctx.set_profiling(True)
    for it in range(3):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        m.compute_device(dL, dR, W, H, W, dD)
        ctx.record(e1)
        log(f"8MP iter {it}: total {ctx.elapsed_ms(e0, e1):.3f} ms", ctx.sgbm_profile())
    t0 = time.time()
    p = so.make_params(numDisparities=D, **KW)
    want = so.compute(L, R, p, nthreads=4)
    log(f"oracle 8MP 4 threads: {time.time() - t0:.2f}s")
    got = np.empty((H, W), np.int16)
    ctx.d2h(got, dD)
    mism("8MP final vs oracle", got, want)
