#!/usr/bin/env python3
"""How sensitive is the GICP flavour of the scanning loop (test/GICP1.py:134-155) to rounding-level differences?  CPU only.

The oracle loop over the recorded frames 8..15 (config C4) is run as is and with the FIRST registration's 4x4 (frame 9) perturbed:
(a) ONE entry moved by one unit in the last place; (b) every entry of its 3x4 part moved by +-5e-16 (three seeds) -- the size of
the difference between the HIP loop and the oracle at that frame (4.7e-16 .. 8e-16: another summation order in the 6x6
reduction).  Recorded per run: per frame the iteration counts, fitness and |T - T_baseline|; how many re-estimated model normals
moved by more than 1e-6 / 1e-3 after frame 9 and how degenerate those neighbourhoods are; the deviation of the final fused cloud.
Writes profiles/r03_gicp_sensitivity.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cloud_oracle as co  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
intr = co.read_intrinsics(os.path.join(G, "camera_intrinsic.json"))
frames = []
for i in range(8, 16):
    p = co.voxel_down_sample_tensor(co.backproject(co.read_png16(os.path.join(G, f"output84/depth_{i:05d}.png")), intr)[0], 0.01)
    frames.append((p, co.estimate_normals_hybrid(p, 0.05, 30)))


def run(hook=None):
    log, snaps = [], []

    def h(k, T):
        T2 = hook(k, T) if hook is not None else None
        snaps.append(np.array(T if T2 is None else T2))
        return T2
    model, normals = co.fuse_loop(frames, "gicp", log=log, hook=h)
    return model, normals, log


def bump_one(k, T):
    if k != 0:
        return None
    T = T.copy()
    T[0, 3] = np.nextafter(T[0, 3], np.inf)            # ONE unit in the last place of one translation component of frame 9
    return T


def bump_all(seed):
    def f(k, T):
        if k != 0:
            return None
        T = T.copy()
        T[:3] += 5e-16 * np.random.default_rng(seed).choice([-1.0, 1.0], (3, 4))
        return T
    return f


ma, na, la = run()
p0 = np.concatenate([frames[0][0], co.transform_points(la[0]["T"], frames[1][0])])
prev0 = np.concatenate([frames[0][1], co.transform_points(la[0]["T"], frames[1][1], rotate_only=True)])
nbr0 = co.hybrid_neighbors(p0, 0.05, 30)
n0, cov0 = co._pca_normals(p0, nbr0, prev0)
w = np.linalg.eigvalsh(cov0)
gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
runs = []
for name, hook in [("one entry of T(frame 9) + 1 ulp", bump_one)] + [(f"3x4 part of T(frame 9) +- 5e-16, seed {sd}", bump_all(sd)) for sd in (0, 1, 2)]:
    mb, nb, lb = run(hook)
    Tb = hook(0, la[0]["T"])
    pb = np.concatenate([frames[0][0], co.transform_points(Tb, frames[1][0])])
    nbr_b = co.hybrid_neighbors(pb, 0.05, 30)
    nb9, _ = co._pca_normals(pb, nbr_b, prev0)
    dn = np.abs(n0 - nb9).max(1)
    moved = np.argsort(-dn)[:8]
    i31, d31 = co._nearest_total_order(p0, p0[moved], 31)           # the 30th and 31st candidates of the most-moved queries
    runs.append({"perturbation": name, "max_point_shift": float(np.abs(p0 - pb).max()),
                 "model_normals_after_frame_9": {"points": int(len(p0)), "moved_by_more_than_1e-6": int((dn > 1e-6).sum()),
                                                 "moved_by_more_than_1e-3": int((dn > 1e-3).sum()), "largest_move": float(dn.max()),
                                                 "most_moved": [{"index": int(i), "dn": float(dn[i]), "neighbours": int(len(nbr0[i])),
                                                                 "relative_gap_of_the_two_smallest_eigenvalues": float(gap[i]),
                                                                 "neighbour_set_changed": bool(set(nbr0[i].tolist()) != set(nbr_b[i].tolist())),
                                                                 "d2_of_30th_and_31st_candidate": [float(d31[j, 29]), float(d31[j, 30])]}
                                                                for j, i in enumerate(moved) if dn[i] > 1e-9]},
                 "per_frame": [{"frame": 9 + i, "iterations": [a["iterations"], b["iterations"]], "fitness": [a["fitness"], b["fitness"]],
                                "correspondences": [int(a["correspondences"]), int(b["correspondences"])],
                                "dT": float(np.abs(a["T"] - b["T"]).max())} for i, (a, b) in enumerate(zip(la, lb))],
                 "final_cloud_max_deviation_m": float(np.abs(ma - mb).max())})
    print(name, [r["iterations"] for r in runs[-1]["per_frame"]], runs[-1]["final_cloud_max_deviation_m"], flush=True)
rec = {"what": "oracle GICP scanning loop (frames 8..15 of test/output84, test/GICP1.py:134-155) against itself under rounding-level perturbations "
               "of the first registration's transform",
       "stop_rule": "RegistrationICP stops when |d fitness| < 1e-6 and |d rmse| < 1e-6; fitness = inliers / N with N ~ 38 k, so ANY change of the inlier "
                    "COUNT (1/N = 2.6e-5) keeps the loop running: the stop iteration is decided by single correspondences at the 0.02 m threshold",
       "baseline_iterations": [r["iterations"] for r in la], "runs": runs,
       "bar": "north_star: 1e-3 on coordinates / normals",
       "mechanism": "the normals that move by 0.04 .. 0.08 are NOT ill-conditioned (eigenvalue gaps 0.65 .. 0.76, 30 neighbours): their 30th and 31st "
                    "nearest candidates are equidistant to the last bit (back-projected pixels of equal depth form exact lattices), so WHICH of them "
                    "enters the Hybrid(0.05, 30) neighbourhood is decided by rounding; the changed normal moves the GICP optimum by ~1e-9 .. 1e-5, "
                    "which flips single correspondences at the 0.02 m threshold, which moves the stop iteration, which moves T by 1e-4 .. 1e-2",
       "conclusion": "a free-running comparison of this loop is not a parity statement beyond the first frames: the reference's own loop, restated, "
                     "does not reproduce itself under a perturbation of one unit in the last place"}
out = os.path.join(ROOT, "profiles", "r03_gicp_sensitivity.json")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print("wrote", out)
