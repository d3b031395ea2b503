"""k_normals at 1 M points (config C3 setup: kNN 20 PCA normals): wall time of estimate_normals from host arrays."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
r3d = importlib.import_module("3d_reconstruction_project_amd")
co = r3d.cloud_ops
src, tgt, _ = r3d.synth.cloud_pair(1_000_000)
src = src.astype(np.float64)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
radius = float(sys.argv[2]) if len(sys.argv) > 2 else None
co.estimate_normals(src, radius, k)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); n = co.estimate_normals(src, radius, k); ts.append(time.perf_counter() - t0)
print("k", k, "radius", radius, "occ", os.environ.get("R3D_KNN_OCC"), "ms", [round(1e3 * t, 2) for t in ts], "checksum", float(np.abs(n).sum()))
