import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
r3d = importlib.import_module("3d_reconstruction_project_amd")
W, H, D = 3264, 2448, 128
L, R, _ = r3d.synth.stereo_pair(W, H, D)
m = r3d.reference_matcher(numDisparities=D, blockSize=5)
for _ in range(3): m.compute(L, R)
t0 = time.perf_counter(); N = 20
for _ in range(N): m.compute(L, R)
dt = (time.perf_counter() - t0) / N
print(f"host-buffer API (pageable numpy in/out, 2x8 MB H2D + 16 MB D2H per map): {1e3*dt:.3f} ms/map = {1/dt:.1f} maps/s")
