"""Per-frame timing of the scanning loop (pipeline.fuse) on the recorded fixture frames: which part of align_point_clouds
takes the time as the model grows."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
from oracle import cloud_oracle as co
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
intr = co.read_intrinsics(os.path.join(G, "camera_intrinsic.json"))
frames = [co.backproject(co.read_png16(os.path.join(G, f"output84/depth_{i:05d}.png")), intr)[0] for i in range(8, 16)]
pa = r3d.PointCloudAlignment(verbose=False)
for rep in range(2):
    model = r3d.PointCloud(frames[0].copy())
    for i in range(1, 8):
        t0 = time.perf_counter()
        a = pa.align_point_clouds(r3d.PointCloud(frames[i]), model)
        dt = time.perf_counter() - t0
        lr = pa.last_result
        t1 = time.perf_counter(); model += a; dt2 = time.perf_counter() - t1
        print(rep, i, "model", len(model.points), "call_ms", round(1e3 * dt, 2), "setup", round(lr["setup_ms"], 2), "loop", round(lr["loop_ms"], 2),
              "it", lr["iterations"], "per_it_us", round(1e3 * lr["loop_ms"] / (lr["iterations"] + 1), 1), "concat_ms", round(1e3 * dt2, 2), flush=True)
