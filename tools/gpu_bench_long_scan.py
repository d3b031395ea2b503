"""The reference's scanning loop (main.py:34-54: every frame aligned to the growing model, model += aligned) over the WHOLE
recorded scan test/output84 (76 depth frames; the reference's own artefact of such a run is 76 / 87 frames long), through
pipeline.fuse_depth_frames: per-frame set-up and loop milliseconds, iterations and model size, to see how the per-frame cost
grows with the model (the target cloud is re-down-sampled and re-indexed every frame: pointcloud_alignment.py:22-23).
Usage (GPU box): python tools/gpu_bench_long_scan.py [n_frames] > gpurun_out/rNN_long_scan.json"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 76
with open(os.path.join(G, "camera_intrinsic.json")) as f:
    intr = json.load(f)
depths = [r3d.io_formats.read_depth_png(os.path.join(G, f"output84/depth_{i:05d}.png")) for i in range(8, 8 + n_frames)]
cam = r3d.cloud_ops.depth_camera(intr)
out = {"frames": len(depths), "impl": os.environ.get("R3D_MODEL_IMPL", "default")}
runs = []
for rep in range(3):          # the first pass grows the device arena to the size of the final model
    log = []
    t0 = time.perf_counter()
    model = r3d.pipeline.fuse_depth_frames(depths, cam, log=log)
    runs.append((1e3 * (time.perf_counter() - t0), log))
out["total_ms_first_pass"] = round(runs[0][0], 1)
dt, log = min(runs[1:], key=lambda r: r[0])
out["total_ms"] = round(dt, 1)
out["model_points"] = int(len(model.points))
size = out["model_points"] - sum(r["appended"] for r in log)      # points of the frame that initialised the model
for r in log:                                                    # model size each frame was aligned TO
    r["model_points"] = size
    size += r["appended"]
pick = [i for i in (0, 11, 31, len(log) - 1) if i < len(log)]     # log[i] = scan frame i + 9 (frame 8 initialises the model)
out["sum_setup_ms"] = round(sum(r["setup_ms"] for r in log), 2)
out["sum_loop_ms"] = round(sum(r["loop_ms"] for r in log), 2)
out["iterations_total"] = int(sum(r["iterations"] for r in log))
out["at_frame"] = {str(i + 9): {k: (round(log[i][k], 3) if isinstance(log[i][k], float) else log[i][k])
                                for k in ("setup_ms", "loop_ms", "iterations", "model_points", "frame_points", "appended")} for i in pick}
out["setup_ms_per_frame"] = [round(r["setup_ms"], 3) for r in log]
out["loop_ms_per_frame"] = [round(r["loop_ms"], 3) for r in log]
out["iterations_per_frame"] = [int(r["iterations"]) for r in log]
out["model_points_per_frame"] = [int(r["model_points"]) for r in log]
out["voxel_table"] = log[-1].get("voxel_table")
out["setup_growth_frame9_to_last"] = round(log[-1]["setup_ms"] / log[0]["setup_ms"], 2)
out["checksum"] = [float(x) for x in np.asarray(model.points).sum(0)]
print(json.dumps(out))
