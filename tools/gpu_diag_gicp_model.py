"""Isolates the first frame at which the GICP-flavour scanning loop leaves the oracle: model normals after frame 9 (re-estimated,
orientation kept) and the registration of frame 10 against the ORACLE's model with the oracle's normals.  GPU box."""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r3d = importlib.import_module("3d_reconstruction_project_amd")
from oracle import cloud_oracle as co
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
intr = co.read_intrinsics(os.path.join(G, "camera_intrinsic.json"))
fr = []
for i in (8, 9, 10):
    p = co.voxel_down_sample_tensor(co.backproject(co.read_png16(os.path.join(G, f"output84/depth_{i:05d}.png")), intr)[0], 0.01)
    fr.append((p, co.estimate_normals_hybrid(p, 0.05, 30)))
mp, mn = co.fuse_loop(fr[:2], "gicp")
# (1) normals re-estimated on the merged model, orientation kept
log = []
mp_raw, mn_raw = fr[0][0], fr[0][1]
res = co.registration(fr[1][0], mp_raw, 0.02, mode="gicp", max_iteration=30, target_normals=mn_raw,
                      target_cov=co.covariances_from_normals(mn_raw), source_cov=co.covariances_from_normals(fr[1][1]))
T = res["T"]
merged = np.concatenate([mp_raw, co.transform_points(T, fr[1][0])])
prev = np.concatenate([mn_raw, fr[1][1] @ T[:3, :3].T])
g_n = r3d.cloud_ops.estimate_normals(merged, 0.05, 30, prev_normals=prev)
o_n = co.estimate_normals_hybrid(merged, 0.05, 30, prev_normals=prev)
err = np.abs(g_n - o_n).max(1)
serr = np.minimum(err, np.abs(g_n + o_n).max(1))
out = {"merged_points": len(merged), "normals_max_err": float(err.max()), "normals_sign_agnostic_max_err": float(serr.max()),
       "n_sign_flips": int((err > 1.0).sum()), "n_err_gt_1e-6": int((serr > 1e-6).sum()),
       "oracle_model_equals": bool(np.abs(merged - mp).max() < 1e-12 and np.abs(o_n - mn).max() < 1e-12)}
# (2) registration of frame 10 against the oracle's model / normals, iteration by iteration
rows = []
for it in (0, 1, 2, 3, 5, 10, 20, 30):
    w = co.registration(fr[2][0], mp, 0.02, mode="gicp", max_iteration=it, target_normals=mn,
                        target_cov=co.covariances_from_normals(mn), source_cov=co.covariances_from_normals(fr[2][1]))
    g = r3d.cloud_ops.registration(fr[2][0], mp, 0.02, mode=2, max_iteration=it, source_normals=fr[2][1], target_normals=mn)
    rows.append({"max_it": it, "it": [w["iterations"], g["iterations"]], "corr": [w["correspondences"], g["correspondences"]],
                 "drmse": g["inlier_rmse"] - w["inlier_rmse"], "dT": float(np.abs(g["T"] - w["T"]).max())})
out["registration"] = rows
out["target_nx_lt_m099"] = int((mn[:, 0] < -0.99).sum())
out["source_nx_lt_m099"] = int((fr[2][1][:, 0] < -0.99).sum())
out["target_unit_norm_err"] = float(np.abs(np.linalg.norm(mn, axis=1) - 1).max())
# (3) the product's own model after frames 8, 9 and the registration of frame 10 against IT
feed = [r3d.PointCloud(p, normals=n) for p, n in fr[:2]]
pm = r3d.pipeline.fuse(feed, flavour="gicp")
out["product_model_dp"] = float(np.abs(pm.points - mp).max())
out["product_model_dn"] = float(np.abs(pm.normals - mn).max())
rows = []
for it in (18, 19, 20, 21, 22, 25, 30):
    w = co.registration(fr[2][0], mp, 0.02, mode="gicp", max_iteration=it, relative_fitness=-1, relative_rmse=-1, target_normals=mn,
                        target_cov=co.covariances_from_normals(mn), source_cov=co.covariances_from_normals(fr[2][1]))
    g = r3d.cloud_ops.registration(fr[2][0], pm.points, 0.02, mode=2, max_iteration=it, relative_fitness=-1, relative_rmse=-1,
                                   source_normals=fr[2][1], target_normals=pm.normals)
    rows.append({"it": it, "corr": [w["correspondences"], g["correspondences"]], "rmse_oracle": w["inlier_rmse"],
                 "drmse": g["inlier_rmse"] - w["inlier_rmse"], "dT": float(np.abs(g["T"] - w["T"]).max())})
out["frame10_vs_product_model"] = rows
g = r3d.cloud_ops.registration(fr[2][0], pm.points, 0.02, mode=2, max_iteration=30, source_normals=fr[2][1], target_normals=pm.normals)
out["frame10_product_model_iterations"] = g["iterations"]
print(json.dumps(out))
