"""Copies the DATA files (no code) that pin the cloud oracle out of the reference's recorded runs.
Run once in the build container (needs /root/reference); the copies are committed because /root/reference does not
exist on the GPU box.  Sources:
  test/output84/  written by test/check84.py:139-191   (voxel 0.02, normals Hybrid r=0.04 max_nn=20); all 76 depth images, clouds 8..11
  test/output/    written by test/check_lama1.py:132-186 (same + SOR(20, 2.0), normals max_nn=30)
  test/dataset/realsense/camera_intrinsic.json  (test/generate_intrinsics.py:28-41)
  Calib_depth/jetson_stereo_8MP_stereo.npz      (Q matrix for disparity -> cloud)
"""
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

jobs = []
for i in range(8, 84):      # the whole recorded scan (76 frames, 6.6 MB): tools/gpu_bench_long_scan.py runs main.py's loop over all of it
    jobs.append((f"test/output84/depth_{i:05d}.png", f"output84/depth_{i:05d}.png"))
for i in range(8, 12):
    jobs.append((f"test/output84/pcd_{i:05d}.ply", f"output84/pcd_{i:05d}.ply"))
    jobs.append((f"test/output/depth_{i:05d}.png", f"output/depth_{i:05d}.png"))
    jobs.append((f"test/output/pcd_{i:05d}.ply", f"output/pcd_{i:05d}.ply"))
jobs.append(("test/output84/color_00008.png", "output84/color_00008.png"))
jobs.append(("test/dataset/realsense/camera_intrinsic.json", "camera_intrinsic.json"))
jobs.append(("Calib_depth/jetson_stereo_8MP_stereo.npz", "jetson_stereo_8MP_stereo.npz"))

for src, dst in jobs:
    d = os.path.join(HERE, dst)
    os.makedirs(os.path.dirname(d), exist_ok=True)
    shutil.copyfile(os.path.join(REF, src), d)
    os.chmod(d, 0o644)
print(f"copied {len(jobs)} files")
