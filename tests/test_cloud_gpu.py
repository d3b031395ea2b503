"""GPU parity tests of the point-cloud kernels (through the C ABI) against the oracle and the recorded fixtures.
Tolerances: voxel means / SOR membership bit-exact; normals 1e-6 sign-agnostic (fixture PLYs) ; registration
transforms 1e-8 against the float64 oracle (north_star bar: 1e-3 on coordinates / normals)."""
import os

import numpy as np
import pytest

from oracle import cloud_oracle as co
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
INTR = co.read_intrinsics(os.path.join(GOLDEN, "camera_intrinsic.json"))


def _sorted(p, *rest):
    o = np.lexsort(p.T[::-1])
    return (p[o],) + tuple(r[o] for r in rest)


def _frame(sub, i):
    d = co.read_png16(os.path.join(GOLDEN, f"{sub}/depth_{i:05d}.png"))
    return co.backproject(d, INTR)[0]


def _sign_agnostic_err(a, b):
    return np.minimum(np.abs(a - b).max(1), np.abs(a + b).max(1))


@pytest.mark.parametrize("frame", [8, 11])
def test_voxel_downsample_reproduces_recorded_ply(r3d, frame):
    pts = _frame("output84", frame)
    ply = co.read_ply(os.path.join(GOLDEN, f"output84/pcd_{frame:05d}.ply"))
    got, _, _ = r3d.cloud_ops.voxel_down_sample(pts, 0.02)
    a, = _sorted(got)
    b, = _sorted(ply["points"])
    assert a.shape == b.shape and np.abs(a - b).max() == 0.0
    # and it already comes out in lexicographic voxel-key order == the oracle's order
    np.testing.assert_array_equal(got, co.voxel_down_sample(pts, 0.02))


def test_voxel_downsample_colors_and_small_voxels(r3d):
    from PIL import Image
    d = co.read_png16(os.path.join(GOLDEN, "output84/depth_00008.png"))
    col = np.asarray(Image.open(os.path.join(GOLDEN, "output84/color_00008.png")))
    pts, (v, u) = co.backproject(d, INTR)
    c = col[v, u] / 255.0
    for voxel in (0.02, 0.01, 0.0025):
        gp, gc, _ = r3d.cloud_ops.voxel_down_sample(pts, voxel, colors=c)
        wp, wc = co.voxel_down_sample(pts, voxel, colors=c)
        np.testing.assert_array_equal(gp, wp)
        np.testing.assert_array_equal(gc, wc)


@pytest.mark.parametrize("sub,k", [("output84", 20), ("output", 30)])
def test_hybrid_normals_reproduce_recorded_ply(r3d, sub, k):
    ply = co.read_ply(os.path.join(GOLDEN, f"{sub}/pcd_00008.ply"))
    """PINNED: cumulant covariance + FastEigen3x3 with the recorded build's fused multiply-adds (cloud.hip fast_eigen3x3,
    oracle/normals.c) against the normals the reference itself wrote -- SIGNED (the closed form's sign is the recorded one),
    within 5e-12 (bar 1e-3), most of them bit for bit; the device acos / cos may differ from the host's in the last place."""
    n = r3d.cloud_ops.estimate_normals(ply["points"], 0.04, k)
    err = np.abs(n - ply["normals"]).max(1)
    assert err.max() < 5e-12 and (err == 0).mean() > 0.5
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-12
    # and against the oracle on the same points
    assert np.abs(n - co.estimate_normals_hybrid(ply["points"], 0.04, k)).max() < 5e-12
    # previous normals fix the sign (legacy EstimateNormals keeps the old orientation)
    n2 = r3d.cloud_ops.estimate_normals(ply["points"], 0.04, k, prev_normals=-ply["normals"])
    assert np.abs(n2 + ply["normals"]).max() < 5e-12


def test_knn_normals_and_sparse_points(r3d):
    pts = co.voxel_down_sample(_frame("output84", 9), 0.01)
    got = r3d.cloud_ops.estimate_normals(pts, None, 20)
    want = co.estimate_normals_knn(pts, 20)
    assert _sign_agnostic_err(got, want).max() < 1e-6
    far = np.array([[0, 0, 0], [5, 5, 5.0], [5.001, 5, 5], [9, 9, 9.0], [0.001, 0, 0], [0, 0.001, 0], [0.0005, 0.0005, 0.001]])
    n = r3d.cloud_ops.estimate_normals(far, 0.01, 30)
    assert (n[[1, 2, 3]] == [0, 0, 1]).all()                 # fewer than 3 neighbours in range


def test_statistical_and_radius_outlier_masks(r3d):
    pts = co.voxel_down_sample(_frame("output", 8), 0.02)
    got = r3d.cloud_ops.statistical_outlier_mask(pts, 20, 2.0)
    np.testing.assert_array_equal(got, co.statistical_outlier_mask(pts, 20, 2.0))
    ply = co.read_ply(os.path.join(GOLDEN, "output/pcd_00008.ply"))
    a, = _sorted(pts[got])
    b, = _sorted(ply["points"])
    assert a.shape == b.shape and np.abs(a - b).max() == 0.0     # == the reference's recorded result
    np.testing.assert_array_equal(r3d.cloud_ops.radius_outlier_mask(pts, 4, 0.03), co.radius_outlier_mask(pts, 4, 0.03))


def test_knn_graph_matches_kdtree(r3d):
    from scipy.spatial import cKDTree
    pts = co.voxel_down_sample(_frame("output84", 10), 0.02)
    nbr, d2 = r3d.cloud_ops.knn_graph(pts, 16)
    d, _ = cKDTree(pts).query(pts, k=16)
    assert np.abs(np.sqrt(d2) - d).max() < 1e-12
    idx, wd2 = co._nearest_total_order(pts, pts, 16)              # ties broken by (distance, index) on both sides
    np.testing.assert_array_equal(nbr, idx)
    np.testing.assert_array_equal(d2, wd2)
    assert (nbr[:, 0] == np.arange(len(pts))).all()
    pts32 = pts.astype(np.float32).astype(np.float64)            # fp32-rounded coordinates: many exact ties
    nbr, _ = r3d.cloud_ops.knn_graph(pts32, 24)
    np.testing.assert_array_equal(nbr, co._nearest_total_order(pts32, pts32, 24)[0])


def test_transform_points(r3d, synth):
    rng = np.random.default_rng(0)
    p = rng.normal(size=(1000, 3))
    T = synth.rigid((1, 2, 3), 33.0, (0.1, -0.2, 0.3))
    assert np.abs(r3d.cloud_ops.transform_points(p, T) - co.transform_points(T, p)).max() < 1e-14
    assert np.abs(r3d.cloud_ops.transform_points(p, T, rotate_only=True) - p @ T[:3, :3].T).max() < 1e-14


def test_transform_blocks_is_the_per_block_transform_in_one_launch(r3d, synth):
    """r3d_transform_blocks_dev (the fuse step of the multi-view exchange): 19 ragged blocks incl. an empty one, points and
    rotate-only normals, in place and out of place -- bit for bit r3d_transform_points_dev block by block (two launches: 16 + 3)."""
    ctx = r3d.default_context(0)
    rng = np.random.default_rng(5)
    sizes = [0 if b == 7 else int(rng.integers(1, 3000)) for b in range(19)]
    host = [rng.normal(size=(n, 3)) for n in sizes]
    Ts = [synth.rigid(tuple(rng.normal(size=3)), float(rng.uniform(0, 90)), tuple(rng.normal(size=3))) for _ in sizes]
    ro = [b % 3 == 1 for b in range(19)]
    d_in = [ctx.to_device(h) if len(h) else 0 for h in host]
    d_out = [ctx.alloc(h.nbytes) if len(h) and b % 2 else d_in[b] for b, h in enumerate(host)]      # even blocks in place
    want = [r3d.cloud_ops.transform_points(h, T, rotate_only=r) if len(h) else h for h, T, r in zip(host, Ts, ro)]
    r3d.cloud_ops.transform_blocks_device([(d_in[b], sizes[b], Ts[b], d_out[b], ro[b]) for b in range(19)], ctx=ctx)
    for b in range(19):
        if sizes[b] == 0:
            continue
        got = np.empty_like(host[b])
        ctx.d2h(got, d_out[b])
        assert np.array_equal(got, want[b]), b
    for b in range(19):
        if sizes[b]:
            ctx.free(d_in[b])
            if d_out[b] != d_in[b]:
                ctx.free(d_out[b])


def _sphere(n, seed, r=1.0):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 3))
    return r * v / np.linalg.norm(v, axis=1, keepdims=True)


def test_p2p_exact_recovery(r3d, synth):
    src = _sphere(4000, 0, 0.5) * np.array([1.0, 0.8, 0.6])
    T = synth.rigid((0.3, -0.5, 0.8), 0.2, (0.001, -0.0015, 0.0008))
    res = r3d.cloud_ops.registration(src, co.transform_points(T, src), 0.02, mode=0, max_iteration=30)
    assert np.abs(res["T"] - T).max() < 1e-10 and res["fitness"] == 1.0 and res["inlier_rmse"] < 1e-10


@pytest.mark.parametrize("mode,name", [(0, "p2p"), (1, "p2plane"), (2, "gicp")])
def test_registration_matches_oracle_iteration_for_iteration(r3d, synth, mode, name):
    rng = np.random.default_rng(3)
    sc = np.array([1.0, 0.7, 0.5])
    tgt = _sphere(20000, 1, 0.3) * sc + rng.normal(0, 2e-4, (20000, 3))
    T = synth.rigid((0.3, -0.5, 0.8), 1.0, (0.004, -0.002, 0.003))
    src = co.transform_points(np.linalg.inv(T), _sphere(15000, 2, 0.3) * sc + rng.normal(0, 2e-4, (15000, 3)))
    tn = co.estimate_normals_knn(tgt, 20)
    sn = co.estimate_normals_knn(src, 20)
    kw = {}
    if name != "p2p":
        kw["target_normals"] = tn
    if name == "gicp":
        kw["target_cov"] = co.covariances_from_normals(tn)
        kw["source_cov"] = co.covariances_from_normals(sn)
    # 0: one evaluation, no update; 7 / 8 / 9: either side of the batch of eight evaluations the host enqueues at a time
    for max_it in (0, 1, 3, 7, 8, 9, 40):
        want = co.registration(src, tgt, 0.02, mode=name, max_iteration=max_it, **kw)
        got = r3d.cloud_ops.registration(src, tgt, 0.02, mode=mode, max_iteration=max_it, source_normals=sn, target_normals=tn)
        assert got["iterations"] == want["iterations"]
        assert got["correspondences"] == want["correspondences"]
        assert abs(got["fitness"] - want["fitness"]) < 1e-12 and abs(got["inlier_rmse"] - want["inlier_rmse"]) < 1e-9
        assert np.abs(got["T"] - want["T"]).max() < 1e-8
    # and the converged answer is the right one
    R_err = got["T"][:3, :3] @ T[:3, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(R_err) - 1) / 2, -1, 1)))
    assert ang < (0.7 if name == "p2p" else 0.05)


def test_registration_with_initial_guess_and_no_overlap(r3d, synth):
    src = _sphere(3000, 5, 0.2)
    T = synth.rigid((0, 0, 1), 5.0, (0.05, 0.0, 0.0))
    tgt = co.transform_points(T, src)
    init = synth.rigid((0, 0, 1), 4.8, (0.049, 0.0005, 0.0))
    got = r3d.cloud_ops.registration(src, tgt, 0.01, init=init, mode=0, max_iteration=30)
    want = co.registration(src, tgt, 0.01, init=init, mode="p2p", max_iteration=30)
    assert np.abs(got["T"] - want["T"]).max() < 1e-8 and got["iterations"] == want["iterations"]
    far = r3d.cloud_ops.registration(src, tgt + 10.0, 0.01, mode=0, max_iteration=5)
    assert far["fitness"] == 0.0 and far["correspondences"] == 0 and np.abs(far["T"] - np.eye(4)).max() == 0


def test_align_point_clouds_drop_in(r3d, capsys):
    """main.py:48 calls align_point_clouds(frame, combined) positionally; result = down-sampled transformed source."""
    src, tgt = _frame("output84", 9), _frame("output84", 8)
    pa = r3d.PointCloudAlignment()
    out = pa.align_point_clouds(r3d.PointCloud(src), r3d.PointCloud(tgt))
    printed = capsys.readouterr().out
    assert "Downsampling point clouds using voxel size: 0.01" in printed and "Performing ICP alignment" in printed
    s = co.voxel_down_sample(src, 0.01)
    t = co.voxel_down_sample(tgt, 0.01)
    want = co.registration(s, t, 0.02, mode="p2p", max_iteration=100)
    assert pa.last_result["iterations"] == want["iterations"]
    assert np.abs(pa.last_result["T"] - want["T"]).max() < 1e-8
    assert np.abs(np.asarray(out.points) - co.transform_points(want["T"], s)).max() < 1e-8   # bar: 1e-3
    assert out.has_normals() and len(out.points) == len(s)
    # GICP flavour (test/GICP1.py:81-104)
    g = r3d.GeneralizedICPAlignment()
    sp = r3d.PointCloud(s)
    tp = r3d.PointCloud(t)
    out2 = g.align_point_clouds(sp, tp)
    sn, tn = co.estimate_normals_hybrid(s, 0.05, 30), co.estimate_normals_hybrid(t, 0.05, 30)
    want2 = co.registration(s, t, 0.02, mode="gicp", max_iteration=30, target_normals=tn,
                            target_cov=co.covariances_from_normals(tn), source_cov=co.covariances_from_normals(sn))
    assert g.last_result["iterations"] == want2["iterations"]
    assert np.abs(np.asarray(out2.points) - co.transform_points(want2["T"], s)).max() < 1e-6


def test_normal_estimation_drop_in(r3d):
    """normal_estimation.py:12-22 on a fixture cloud: fp32-rounded coordinates, k<=50 within 0.05, then orientation."""
    pts = co.voxel_down_sample(_frame("output84", 8), 0.01)
    out = r3d.NormalEstimation("CUDA:0").estimate_normals(r3d.PointCloud(pts))
    p32 = pts.astype(np.float32).astype(np.float64)
    want = co.estimate_normals_hybrid(p32, 0.05, 50)
    n = np.asarray(out.normals)
    assert _sign_agnostic_err(n, want).max() < 1e-5
    # orientation: neighbouring normals agree after propagation (a depth-camera surface is one open sheet)
    from scipy.spatial import cKDTree
    _, idx = cKDTree(p32).query(p32, k=2)
    assert ((n * n[idx[:, 1]]).sum(1) > 0).mean() > 0.98


def test_multi_scale_point_to_plane(r3d, synth):
    """test/check2.py:143-156: three scales [15, 5, 1.5] * voxel with [30, 20, 10] iterations, each from the previous."""
    rng = np.random.default_rng(4)
    sc = np.array([1.0, 0.7, 0.5])
    tgt = _sphere(15000, 1, 0.3) * sc + rng.normal(0, 2e-4, (15000, 3))
    T = synth.rigid((0.3, -0.5, 0.8), 6.0, (0.03, -0.02, 0.025))
    src = co.transform_points(np.linalg.inv(T), _sphere(12000, 2, 0.3) * sc + rng.normal(0, 2e-4, (12000, 3)))
    tn = co.estimate_normals_knn(tgt, 20)
    voxel = 0.004
    Tg, results = r3d.multi_scale_icp(r3d.PointCloud(src), r3d.PointCloud(tgt, normals=tn), voxel)
    Tw = np.eye(4)
    for scale, n_it, got in zip((15.0, 5.0, 1.5), (30, 20, 10), results):
        want = co.registration(src, tgt, voxel * scale, init=Tw, mode="p2plane", max_iteration=n_it, target_normals=tn)
        Tw = want["T"]
        assert got["iterations"] == want["iterations"] and np.abs(got["T"] - want["T"]).max() < 1e-8
    R_err = Tg[:3, :3] @ T[:3, :3].T
    assert np.degrees(np.arccos(np.clip((np.trace(R_err) - 1) / 2, -1, 1))) < 0.05


def test_cloud_edge_cases(r3d):
    co_g = r3d.cloud_ops
    one = np.array([[0.1, 0.2, 0.3]])
    p, _, _ = co_g.voxel_down_sample(one, 0.01)
    assert np.array_equal(p, one)
    assert (co_g.estimate_normals(one, 0.05, 30) == [[0, 0, 1]]).all()
    few = np.array([[0, 0, 0], [0.01, 0, 0], [0, 0.01, 0], [0.01, 0.01, 0.0]])
    n = co_g.estimate_normals(few, 0.05, 30)
    assert np.abs(np.abs(n[:, 2]) - 1).max() < 1e-12                      # coplanar points: normal = +-z
    dup = np.repeat(few, 5, axis=0)                                       # exact duplicates: distance ties everywhere
    np.testing.assert_array_equal(co_g.voxel_down_sample(dup, 0.5)[0], co.voxel_down_sample(dup, 0.5))
    nbr, d2 = co_g.knn_graph(dup, 7)
    np.testing.assert_array_equal(nbr, co._nearest_total_order(dup, dup, 7)[0])
    empty = co_g.voxel_down_sample(np.zeros((0, 3)), 0.01)[0]
    assert empty.shape == (0, 3)
    with pytest.raises(r3d.R3DError):
        co_g.registration(np.zeros((0, 3)), few, 0.02)
    with pytest.raises(r3d.R3DError):
        co_g.registration(few, few, 0.02, mode=co_g.P2PLANE)             # point-to-plane without target normals
    with pytest.raises(r3d.R3DError):
        co_g.registration(few, few, -1.0)
    res = co_g.registration(few, few, 0.02, max_iteration=0)
    assert res["iterations"] == 0 and res["fitness"] == 1.0 and np.array_equal(res["T"], np.eye(4))


def test_orient_normals_makes_a_closed_surface_consistent(r3d):
    p = _sphere(8000, 7, 0.5)
    rng = np.random.default_rng(1)
    n = p / np.linalg.norm(p, axis=1, keepdims=True) * np.where(rng.random(len(p)) < 0.5, -1.0, 1.0)[:, None]
    out = r3d.cloud_ops.orient_normals(p, n, 12)
    s = np.sign((out * p).sum(1))
    assert abs(s.mean()) == 1.0                         # all outward or all inward
    assert s[np.argmax(p[:, 2])] == 1.0                 # seeded at the highest point, turned towards +z
    np.testing.assert_array_equal(np.abs(out), np.abs(n))
    np.testing.assert_array_equal(out, r3d.cloud_ops.orient_normals(p, n, 12))     # deterministic


def test_lds_tiled_search_kernel_matches_default():
    """R3D_ICP_IMPL=exact (all-float64 search) and =tiled (LDS-tiled all-float64 search) are kept for A/B beside the default
    two-stage search.  tiled: same candidates in the same order,
    so the correspondence counts are identical; the sums are reduced in a different order (per wave instead of per
    256-thread block), so transforms agree to rounding (1e-12), in every mode."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = (
        "import importlib, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "r3d = importlib.import_module('3d_reconstruction_project_amd')\n"
        "src, tgt, _ = r3d.synth.cloud_pair(60000, scale=0.3)\n"
        "src, tgt = src.astype(np.float64), tgt.astype(np.float64)\n"
        "sn, tn = r3d.cloud_ops.estimate_normals(src, None, 20), r3d.cloud_ops.estimate_normals(tgt, None, 20)\n"
        "for mode in (0, 1, 2):\n"
        "    r = r3d.cloud_ops.registration(src, tgt, 0.02, mode=mode, max_iteration=6, source_normals=sn, target_normals=tn)\n"
        "    print('RES', mode, r['T'].tobytes().hex(), r['correspondences'], repr(r['inlier_rmse']))\n"
        # a thin slab whose source overhangs the target on every side: queries in border cells and outside the grid
        "rng = np.random.default_rng(5)\n"
        "tgt = np.c_[rng.random((40000, 2)), 0.002 * rng.standard_normal(40000)]\n"
        "src = np.c_[rng.random((30000, 2)) * 1.1 - 0.05, 0.002 * rng.standard_normal(30000)] + np.array([0.004, -0.003, 0.002])\n"
        "sn, tn = r3d.cloud_ops.estimate_normals(src, None, 20), r3d.cloud_ops.estimate_normals(tgt, None, 20)\n"
        "for mode in (0, 1, 2):\n"
        "    r = r3d.cloud_ops.registration(src, tgt, 0.02, mode=mode, max_iteration=6, source_normals=sn, target_normals=tn)\n"
        "    print('RES', 3 + mode, r['T'].tobytes().hex(), r['correspondences'], repr(r['inlier_rmse']))\n")
    outs = []
    for impl in ("exact", "tiled", "default", "f32"):
        env = dict(os.environ, R3D_ICP_IMPL=impl)
        o = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        lines = [ln.split() for ln in o.stdout.splitlines() if ln.startswith("RES")]
        assert len(lines) == 6, o.stdout + o.stderr
        outs.append(lines)
    for a, b in zip(outs[0], outs[1]):
        assert a[3] == b[3] and int(a[3]) > 20000                                     # correspondences
        Ta, Tb = (np.frombuffer(bytes.fromhex(x[2])).reshape(4, 4) for x in (a, b))
        assert np.abs(Ta - Tb).max() < 1e-12 and abs(float(a[4]) - float(b[4])) < 1e-12
    # the two-stage searches (default: packed 10-bit cell-relative candidates with row pruning; f32: float32 copies of all
    # nine rows; both followed by the exact evaluation of the best three) find exactly the correspondences of the all-float64
    # search and sum them in the same order: bit-identical transforms
    assert outs[0] == outs[2]
    assert outs[0] == outs[3]


def test_pooled_walk_of_large_clouds_matches_the_all_float64_search():
    """Clouds of more than 131 072 source points take the packed search whose lanes pool their row lists per wave
    (nn_block_q10<1>, R3D_ICP_POOL).  Against the all-float64 search (R3D_ICP_IMPL=exact) and against the per-lane walk
    (R3D_ICP_POOL=0): the same correspondences summed in the same order, so bit-identical transforms -- on a curved surface and
    on a patch so dense (a dozen points per cell at the finest grid) that the waves' queues overflow and fall back to the
    per-lane walk, or exceed the 1 024-candidate ordinal range and take the float64 block search."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = (
        "import importlib, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "r3d = importlib.import_module('3d_reconstruction_project_amd')\n"
        "src, tgt, _ = r3d.synth.cloud_pair(200000, scale=0.5)\n"
        "src, tgt = src.astype(np.float64), tgt.astype(np.float64)\n"
        "sn, tn = r3d.cloud_ops.estimate_normals(src, None, 20), r3d.cloud_ops.estimate_normals(tgt, None, 20)\n"
        "for mode in (0, 2):\n"
        "    r = r3d.cloud_ops.registration(src, tgt, 0.02, mode=mode, max_iteration=4, source_normals=sn, target_normals=tn)\n"
        "    print('RES', mode, r['T'].tobytes().hex(), r['correspondences'], repr(r['inlier_rmse']))\n"
        "rng = np.random.default_rng(11)\n"
        "for k, (n, side) in enumerate(((300000, 0.19), (300000, 0.11))):\n"
        "    tgt = np.c_[rng.random((n, 2)) * side, 0.0004 * rng.standard_normal(n)]\n"
        "    src = np.c_[rng.random((n // 2 + 70000, 2)) * side, 0.0004 * rng.standard_normal(n // 2 + 70000)] + np.array([0.002, -0.001, 0.001])\n"
        "    r = r3d.cloud_ops.registration(src, tgt, 0.02, mode=0, max_iteration=3)\n"
        "    print('RES', 10 + k, r['T'].tobytes().hex(), r['correspondences'], repr(r['inlier_rmse']))\n")
    outs = []
    for env_add in ({"R3D_ICP_IMPL": "exact"}, {}, {"R3D_ICP_POOL": "0"}):
        env = dict(os.environ, **env_add)
        o = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        lines = [ln.split() for ln in o.stdout.splitlines() if ln.startswith("RES")]
        assert len(lines) == 4, o.stdout + o.stderr
        outs.append(lines)
    assert all(int(x[3]) > 100000 for x in outs[0])
    assert outs[0] == outs[1]
    assert outs[0] == outs[2]


def test_non_finite_coordinates_are_refused(r3d):
    """A zero disparity reprojects to infinity; a grid around such a point cannot be built, so the cloud entry points raise
    instead of looping or indexing with garbage."""
    pts = np.random.default_rng(0).random((1000, 3))
    pts[17, 1] = np.inf
    with pytest.raises(r3d.R3DError, match="non-finite"):
        r3d.cloud_ops.voxel_down_sample(pts, 0.1)
    with pytest.raises(r3d.R3DError, match="non-finite"):
        r3d.cloud_ops.estimate_normals(pts, 0.2, 10)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fused_align_call_equals_chained_entry_points(r3d, mode):
    """r3d_align_point_clouds (one device-resident call) == voxel_down_sample x2 -> estimate_normals x2 -> registration ->
    transform through the separate host-buffer entry points: same kernels, same correspondences, same iteration counts.  Since
    round 4 the fused call lays the registration's search grid out in the bounding box of the target BEFORE down-sampling (it has
    that box from the voxel grid and saves a bounding-box round trip; the chained call only ever sees the down-sampled target), so
    the sources are visited in a different order and the 29 sums differ in their last bits: transforms equal to 1e-12, not bit for
    bit; everything that does not pass through those sums (voxel means, colours) stays exact."""
    ops = r3d.cloud_ops
    src, tgt = _frame("output84", 10), _frame("output84", 8)
    rng = np.random.default_rng(0)
    col = rng.random(src.shape)
    got = ops.align_point_clouds(src, tgt, 0.02, 0.01, 40, mode, 0.02, 30, source_colors=col)
    sp, sc, _ = ops.voxel_down_sample(src, 0.01, col)
    tp, _, _ = ops.voxel_down_sample(tgt, 0.01)
    sn, tn = ops.estimate_normals(sp, 0.02, 30), ops.estimate_normals(tp, 0.02, 30)
    res = ops.registration(sp, tp, 0.02, np.eye(4), mode, 40, 1e-6, 1e-6, sn, tn)
    assert np.abs(got["T"] - res["T"]).max() < 1e-12
    assert got["iterations"] == res["iterations"] and got["correspondences"] == res["correspondences"]
    np.testing.assert_array_equal(got["points"], ops.transform_points(sp, got["T"]))
    np.testing.assert_array_equal(got["normals"], ops.transform_points(sn, got["T"], rotate_only=True))
    np.testing.assert_array_equal(got["colors"], sc)
    # no down-sampling, no normals: plain point-to-point on the raw clouds
    if mode == 0:
        raw = ops.align_point_clouds(sp, tp, 0.02, None, 10, 0, None, 0)
        ref = ops.registration(sp, tp, 0.02, np.eye(4), 0, 10)
        np.testing.assert_array_equal(raw["T"], ref["T"])
        assert raw["normals"] is None and raw["colors"] is None and len(raw["points"]) == len(sp)
    else:
        with pytest.raises(r3d.R3DError):
            ops.align_point_clouds(sp, tp, 0.02, None, 10, mode, None, 0)             # this mode needs normals


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17])
def test_tiny_clouds_stay_finite(r3d, n):
    """Clouds of one to a few points: every cloud entry point returns finite results.  One correspondence pair (or coincident
    pairs) gives a zero cross-covariance, whose SVD is U = V = I: point-to-point reduces to the translation between the means;
    collinear pairs (rank 1) take the smallest rotation; rank-deficient 6x6 systems return the identity update like
    SolveJacobianSystemAndObtainExtrinsicMatrix."""
    ops = r3d.cloud_ops
    p = np.random.default_rng(n).random((n, 3))
    v = ops.voxel_down_sample(p, 0.5)[0]
    assert 1 <= len(v) <= n
    nr = ops.estimate_normals(p, 0.5, 30)
    assert nr.shape == (n, 3) and np.isfinite(nr).all()
    shift = np.array([0.001, -0.002, 0.0015])
    res = ops.registration(p, p + shift, 0.05, mode=0, max_iteration=5)
    assert np.isfinite(res["T"]).all() and res["correspondences"] == n
    assert np.abs(res["T"][:3, 3] + res["T"][:3, :3] @ p.mean(0) - (p + shift).mean(0)).max() < 1e-12   # means are mapped onto each other
    if n == 1:
        np.testing.assert_allclose(res["T"], np.block([[np.eye(3), shift[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]), atol=1e-15)
    for mode in (1, 2):
        res = ops.registration(p, p + shift, 0.05, mode=mode, max_iteration=5, source_normals=nr, target_normals=nr)
        assert np.isfinite(res["T"]).all()
    a = ops.align_point_clouds(p, p + shift, 0.05, 0.1, 5, 0, 0.2, 10)
    assert np.isfinite(a["T"]).all()


def test_degenerate_geometry_stays_finite(r3d):
    ops = r3d.cloud_ops
    p = np.zeros((100, 3)); p[:, 0] = np.linspace(0, 1, 100)                      # collinear
    nr = ops.estimate_normals(p, 0.2, 20)
    assert np.isfinite(nr).all()
    for mode in (0, 1, 2):
        res = ops.registration(p, p + np.array([0.0, 0.001, 0.0]), 0.05, mode=mode, max_iteration=3, source_normals=nr, target_normals=nr)
        assert np.isfinite(res["T"]).all(), mode
    res = ops.registration(p, p + np.array([0.0, 0.001, 0.0]), 0.05, mode=0, max_iteration=1)
    np.testing.assert_allclose(res["T"][:3, :3], np.eye(3), atol=1e-12)          # smallest rotation for rank-1 pairs: none
    d = np.ones((50, 3))                                                          # all points coincident
    assert np.isfinite(ops.estimate_normals(d, 0.2, 20)).all()
    assert np.isfinite(ops.registration(d, d, 0.05, mode=0, max_iteration=3)["T"]).all()


# ------------------------------------------------------------------------------- normal orientation (rows c2 / f-3)
def _oriented_case(r3d, pts, k):
    """Product (Qhull edges on the host, k-NN graph on the device, trees + propagation in the library) vs the oracle."""
    n0 = co.estimate_normals_hybrid(pts, 0.05, 30)
    edges = r3d.orientation.delaunay_edges(pts)
    np.testing.assert_array_equal(edges, co.delaunay_edges(pts))            # same Qhull call on both sides
    got = r3d.orientation.orient_normals_consistent_tangent_plane(pts, n0, k)
    want = co.orient_normals(pts, n0, k)
    return n0, got, want


def test_orient_normals_matches_oracle_sign_for_sign(r3d):
    """normal_estimation.py:21 orient_normals_consistent_tangent_plane(100) on a recorded frame (fp32-rounded, voxel 0.01)."""
    pts = co.voxel_down_sample(_frame("output84", 8), 0.01).astype(np.float32).astype(np.float64)
    n0, got, want = _oriented_case(r3d, pts, 100)
    np.testing.assert_array_equal(got, want)                                # every sign equal, magnitudes untouched
    np.testing.assert_array_equal(np.abs(got), np.abs(n0))
    assert (got != n0).any(axis=1).sum() > 100                              # the propagation did flip a share of them
    assert got[np.argmax(pts[:, 2]), 2] >= 0


@pytest.mark.parametrize("k", [2, 12, 40])
def test_orient_normals_oracle_parity_synthetic(r3d, k):
    rng = np.random.default_rng(k)
    p = _sphere(6000, 7, 0.5) + rng.normal(0, 2e-3, (6000, 3))
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    n = n * np.where(rng.random(len(p)) < 0.5, -1.0, 1.0)[:, None]
    got = r3d.orientation.orient_normals_consistent_tangent_plane(p, n, k)
    np.testing.assert_array_equal(got, co.orient_normals(p, n, k))
    s = np.sign((got * p).sum(1))
    assert abs(s.mean()) == 1.0 and s[np.argmax(p[:, 2])] == 1.0
    # two far-apart sheets: the Delaunay graph keeps them connected (the k-NN-only variant roots each component itself)
    q = np.concatenate([p, p * np.array([1.0, 1.0, 0.2]) + np.array([5.0, 0.0, -3.0])])
    nq = np.concatenate([n, n])
    np.testing.assert_array_equal(r3d.orientation.orient_normals_consistent_tangent_plane(q, nq, k), co.orient_normals(q, nq, k))
    with pytest.raises(ValueError):
        r3d.orientation.orient_normals_consistent_tangent_plane(p[:3], n[:3], k)
    with pytest.raises(r3d.R3DError):
        r3d.cloud_ops.orient_normals(p, n, k, delaunay_edges=np.array([[0, len(p)]]))


def test_normal_estimation_drop_in_orientation_equals_oracle(r3d):
    pts = co.voxel_down_sample(_frame("output84", 10), 0.01)
    out = r3d.NormalEstimation("CUDA:0").estimate_normals(r3d.PointCloud(pts))
    p32 = pts.astype(np.float32).astype(np.float64)
    n_gpu_unoriented = r3d.cloud_ops.estimate_normals(p32, 0.05, 50)
    want = co.orient_normals(p32, n_gpu_unoriented, 100)
    np.testing.assert_array_equal(np.asarray(out.normals), want)
    assert _sign_agnostic_err(n_gpu_unoriented, co.estimate_normals_hybrid(p32, 0.05, 50)).max() < 1e-5


# ------------------------------------------------------------------------- tensor voxel grid / outlier stages (f-1)
def test_tensor_voxel_downsample_matches_oracle(r3d):
    """o3d.t voxel_down_sample on a Float32 cloud (pointcloud_processing.py:27): float32 keys from origin 0, float32 means."""
    from PIL import Image
    d = co.read_png16(os.path.join(GOLDEN, "output84/depth_00008.png"))
    col = np.asarray(Image.open(os.path.join(GOLDEN, "output84/color_00008.png")))
    pts, (v, u) = co.backproject(d, INTR)
    c = col[v, u] / 255.0
    for voxel in (0.02, 0.0025):
        gp, gc, _ = r3d.cloud_ops.voxel_down_sample(pts, voxel, colors=c, tensor=True)
        wp, wc = co.voxel_down_sample_tensor(pts, voxel, colors=c)
        np.testing.assert_array_equal(gp, wp)
        np.testing.assert_array_equal(gc, wc)
        assert np.array_equal(gp, gp.astype(np.float32))                  # float32 values
    neg = np.array([[-0.015, 0.0, 0.0], [-0.005, 0.0, 0.0], [0.005, 0.0, 0.0], [0.0049, 0.009, 0.0]])
    np.testing.assert_array_equal(r3d.cloud_ops.voxel_down_sample(neg, 0.01, tensor=True)[0], co.voxel_down_sample_tensor(neg, 0.01))


def test_process_point_cloud_chain_matches_oracle(r3d):
    """pointcloud_processing.py:23-44: tensor voxel 0.0025 -> remove_statistical_outlier(30, 1.2) -> remove_radius_outlier(16, 0.01)."""
    pts = _frame("output84", 9)
    out = r3d.PointCloudProcessingWithCUDA("CUDA:0").process_point_cloud(r3d.PointCloud(pts))
    w = co.voxel_down_sample_tensor(pts, 0.0025)
    w = w[co.statistical_outlier_mask(w, 30, 1.2)]
    w = w[co.radius_outlier_mask(w, 16, 0.01)]
    np.testing.assert_array_equal(np.asarray(out.points), w)
    assert 1000 < len(w) < len(pts)


def test_statistical_outlier_with_coincident_points(r3d):
    """>= k coincident points score 0: the original leaves them out of the mean / deviation sums and rejects them."""
    rng = np.random.default_rng(5)
    base = rng.random((400, 3))
    pts = np.concatenate([base, np.repeat(base[:7], 6, axis=0)])            # 7 sites with 7 coincident points each
    got = r3d.cloud_ops.statistical_outlier_mask(pts, 5, 1.0)
    np.testing.assert_array_equal(got, co.statistical_outlier_mask(pts, 5, 1.0))
    assert not got[400:].any() and not got[:7].any() and got.sum() > 100
    assert r3d.cloud_ops.statistical_outlier_mask(np.zeros((0, 3)), 5, 1.0).shape == (0,)
    assert r3d.cloud_ops.radius_outlier_mask(np.zeros((0, 3)), 5, 1.0).shape == (0,)
    assert r3d.cloud_ops.estimate_normals(np.zeros((0, 3)), 0.05, 30).shape == (0, 3)


# ------------------------------------------------------------------------------------- BASELINE config C3 at full size
@pytest.fixture(scope="module")
def c3(r3d):
    src, tgt, T_star = r3d.synth.cloud_pair(1_000_000)
    src, tgt = src.astype(np.float64), tgt.astype(np.float64)
    sn = r3d.cloud_ops.estimate_normals(src, None, 20)
    tn = r3d.cloud_ops.estimate_normals(tgt, None, 20)
    return src, tgt, sn, tn, T_star


def test_c3_knn20_normals_at_one_million_points(r3d, c3):
    """kNN-20 PCA normals of the 1 M-point clouds (the 'no normals' branch of GICP) against the oracle on 50 000 random queries."""
    src, tgt, sn, tn, _ = c3
    for pts, nrm, seed in ((src, sn, 0), (tgt, tn, 1)):
        q = np.random.default_rng(seed).choice(len(pts), 50_000, replace=False)
        want = co.estimate_normals_knn(pts, 20, queries=q)
        err = _sign_agnostic_err(nrm[q], want)
        assert err.max() < 1e-6 and np.median(err) < 1e-12
        assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-12


def test_c3_gicp_at_one_million_points_matches_oracle(r3d, c3):
    """SURVEY 8d pass rule for C3: ||T - T*||_F <= 1e-3 after exactly 20 iterations, and the evaluation after 0, 1 and 2
    updates (correspondence count, fitness, inlier RMSE, transform) equal to the CPU restatement.  Exercises the paths only a
    cloud of this size reaches: the 144 MB dense cell table, the one-residency XCD-share grid, the outer-shell walk of the
    unaligned first evaluation."""
    src, tgt, sn, tn, T_star = c3
    kw = dict(relative_fitness=-1, relative_rmse=-1, source_normals=sn, target_normals=tn)
    res = r3d.cloud_ops.registration(src, tgt, 0.02, mode=r3d.cloud_ops.GICP, max_iteration=20, **kw)
    assert res["iterations"] == 20 and np.linalg.norm(res["T"] - T_star) <= 1e-3
    hist = []
    want = co.registration(src, tgt, 0.02, mode="gicp", max_iteration=2, relative_fitness=-1, relative_rmse=-1, target_normals=tn,
                           target_cov=co.covariances_from_normals(tn), source_cov=co.covariances_from_normals(sn), history=hist)
    w0 = co.registration(src, tgt, 0.02, mode="gicp", max_iteration=0, target_normals=tn,
                         target_cov=co.covariances_from_normals(tn), source_cov=co.covariances_from_normals(sn))
    stages = [(w0["fitness"], w0["inlier_rmse"])] + hist
    for max_it in (0, 1, 2):
        got = r3d.cloud_ops.registration(src, tgt, 0.02, mode=r3d.cloud_ops.GICP, max_iteration=max_it, **kw)
        fit, rmse = stages[max_it]
        assert got["iterations"] == max_it
        assert got["correspondences"] == round(fit * len(src))
        assert abs(got["fitness"] - fit) < 1e-12 and abs(got["inlier_rmse"] - rmse) < 1e-9       # bar: 1e-6
    assert got["correspondences"] == want["correspondences"]
    assert np.abs(got["T"] - want["T"]).max() < 1e-9
    # point-to-point and point-to-plane at full size: one update each against the oracle
    for mode, name in ((r3d.cloud_ops.P2P, "p2p"), (r3d.cloud_ops.P2PLANE, "p2plane")):
        g = r3d.cloud_ops.registration(src, tgt, 0.02, mode=mode, max_iteration=1, **kw)
        w = co.registration(src, tgt, 0.02, mode=name, max_iteration=1, relative_fitness=-1, relative_rmse=-1, target_normals=tn)
        assert g["correspondences"] == w["correspondences"] and abs(g["inlier_rmse"] - w["inlier_rmse"]) < 1e-9
        assert np.abs(g["T"] - w["T"]).max() < 1e-9


# ------------------------------------------------------------------ depth image -> cloud (PINNED by the recorded PLY files)
@pytest.mark.parametrize("sub,frame", [("output84", 8), ("output84", 11), ("output", 8), ("output", 10)])
def test_backproject_depth_reproduces_recorded_ply(r3d, sub, frame):
    """create_from_rgbd_image + flip (test/check84.py:155-159,172-178) on the device: equal to the oracle point for point, and
    after voxel_down_sample(0.02) (+ the outlier filter of that capture) equal to the PLY the reference itself recorded."""
    from PIL import Image
    d = co.read_png16(os.path.join(GOLDEN, f"{sub}/depth_{frame:05d}.png"))
    cam = r3d.cloud_ops.depth_camera(INTR)
    pts, col, pix = r3d.cloud_ops.backproject_depth(d, cam, want_pixels=True)
    want, (v, u) = co.backproject(d, INTR)
    np.testing.assert_array_equal(pts, want)
    np.testing.assert_array_equal(pix, v * d.shape[1] + u)
    assert col is None
    ply = co.read_ply(os.path.join(GOLDEN, f"{sub}/pcd_{frame:05d}.ply"))
    vox, _, _ = r3d.cloud_ops.voxel_down_sample(pts, 0.02)
    if sub == "output":                                        # that capture also ran remove_statistical_outlier(20, 2.0)
        vox = vox[r3d.cloud_ops.statistical_outlier_mask(vox, 20, 2.0)]
    a, = _sorted(vox)
    b, = _sorted(ply["points"])
    assert a.shape == b.shape and np.abs(a - b).max() == 0.0
    if sub == "output84" and frame == 8:                       # colours: channel / 255, voxel means as recorded
        img = np.asarray(Image.open(os.path.join(GOLDEN, "output84/color_00008.png")))
        p2, c2 = r3d.cloud_ops.backproject_depth(d, cam, color=img)
        np.testing.assert_array_equal(p2, want)
        np.testing.assert_array_equal(c2, img[v, u] / 255.0)
    # options: no flip, other truncation; an image without a valid pixel gives an empty cloud
    nf, _ = r3d.cloud_ops.backproject_depth(d, r3d.cloud_ops.depth_camera(INTR, depth_trunc=1.5, flip=False))
    wf, _ = co.backproject(d, INTR, depth_trunc=1.5, flip=False)
    np.testing.assert_array_equal(nf, wf)
    assert r3d.cloud_ops.backproject_depth(np.zeros((4, 5), np.uint16), cam)[0].shape == (0, 3)


def test_backproject_depth_exactly_at_the_truncation(r3d):
    """a depth exactly at depth_trunc is dropped (`>=` in double, as the original)."""
    d = np.array([[2999, 3000, 3001, 0]], np.uint16)
    for trunc in (3.0, 3.0005):
        cam = r3d.cloud_ops.depth_camera(INTR, depth_scale=1000.0, depth_trunc=trunc)
        got, _ = r3d.cloud_ops.backproject_depth(d, cam)
        want, _ = co.backproject(d, INTR, depth_scale=1000.0, depth_trunc=trunc)
        np.testing.assert_array_equal(got, want)
        assert len(got) == (1 if trunc == 3.0 else 2)


def test_scanning_loop_from_depth_images_equals_loop_over_clouds(r3d):
    """main.py:34-54 fed with the recorded depth PNGs (back-projection on the device, model in HBM) == the same loop over the
    oracle's back-projected clouds, colours included; failed captures (None, all-zero, everything beyond depth_trunc) are skipped."""
    from PIL import Image
    depths = [co.read_png16(os.path.join(GOLDEN, f"output84/depth_{i:05d}.png")) for i in (8, 9, 10)]
    img = np.asarray(Image.open(os.path.join(GOLDEN, "output84/color_00008.png")))
    cam = r3d.cloud_ops.depth_camera(INTR)
    far = np.full_like(depths[0], 60000)                       # 60 m: beyond depth_trunc
    feed = [None, depths[0], np.zeros_like(depths[0]), depths[1], far, depths[2]]
    log = []
    got = r3d.pipeline.fuse_depth_frames(feed, cam, colors=[img] * len(feed), log=log)
    clouds = []
    for d in depths:
        p, (v, u) = co.backproject(d, INTR)
        clouds.append(r3d.PointCloud(p, colors=img[v, u] / 255.0))
    want = r3d.pipeline.fuse(clouds, flavour="icp")
    assert len(log) == 2 and log[0]["frame_points"] == len(clouds[1])
    np.testing.assert_array_equal(got.points, want.points)
    np.testing.assert_array_equal(got.colors, want.colors)
    assert r3d.pipeline.fuse_depth_frames([None, far], cam).points.shape == (0, 3)


# ------------------------------------------------------------------ the hand-written stable counting sort behind every grid build
def _host_keys(p, org, cell, dims, order):
    nx, ny, nz = dims
    if order == 1:
        c = np.floor((p - org) / cell).astype(np.int64)
    else:
        c = np.floor((p - org) * (1.0 / cell)).astype(np.int64)
    c = np.clip(c, 0, np.array(dims) - 1)
    if order == 0:
        return (c[:, 2] * ny + c[:, 1]) * nx + c[:, 0]
    if order == 1:
        return (c[:, 0] * ny + c[:, 1]) * nz + c[:, 2]

    def spread(v):
        out = np.zeros_like(v)
        for b in range(21):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return spread(c[:, 0]) | spread(c[:, 1]) << 1 | spread(c[:, 2]) << 2


@pytest.mark.parametrize("case", ["random_x_fastest", "voxel_raster_runs", "morton", "tiny", "one_bucket_many_runs", "wide_keys"])
def test_counting_sort_is_the_stable_key_index_order(r3d, case):
    """the run-based counting sort (k_cs_runs / place / rank / emit) must produce THE stable order by
    (cell key, original index) -- what a host lexsort gives and what the library radix sort (kept as the fallback) gives."""
    rng = np.random.default_rng(11)
    org = np.array([-1.0, -2.0, 0.5])
    if case == "random_x_fastest":
        p, cell, dims, order = org + rng.random((200_003, 3)) * [3.0, 2.0, 1.0], 0.01, (301, 201, 101), 0
    elif case == "voxel_raster_runs":      # pixels of an image in raster order: long runs of equal voxel keys
        u, v = np.meshgrid(np.arange(1500), np.arange(700))
        z = 1.0 + 0.2 * np.sin(u / 90.0) * np.cos(v / 70.0)
        p = org + np.stack([u.ravel() * 0.0008 * z.ravel(), v.ravel() * 0.0008 * z.ravel(), z.ravel() - 0.7], 1)
        cell, dims, order = 0.01, (160, 80, 80), 1
    elif case == "morton":
        p, cell, dims, order = org + rng.random((70_001, 3)) * [1.0, 1.0, 1.0], 0.004, (251, 251, 251), 2
    elif case == "tiny":
        p, cell, dims, order = org + rng.random((1, 3)), 0.5, (3, 3, 3), 0
    elif case == "one_bucket_many_runs":    # > 4096 single-point runs in ONE voxel column: the counting sort declines, radix takes over
        p = np.tile(org + [0.005, 0.005, 0.0], (12_000, 1))
        p[:, 2] += 0.01 * (np.arange(12_000) % 7) + 0.005
        cell, dims, order = 0.01, (4, 4, 16), 1
    else:                                   # key space beyond 32 bits: 64-bit keys
        p, cell, dims, order = org + rng.random((50_000, 3)) * [2.0, 2.0, 2.0], 0.001, (2001, 2001, 2001), 1
    want_keys = _host_keys(p, org, cell, dims, order)
    want = np.lexsort((np.arange(len(p)), want_keys))
    for impl in (0, 1):
        idx, keys = r3d.cloud_ops.debug_sort_by_cell(p, org, cell, dims, order, impl)
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(keys.astype(np.int64), want_keys[want])


def test_voxel_grid_survives_the_deferred_sort_test_failing(r3d):
    """Round 4: the voxel grid defers its sort's fallback test into the read-back of the segment count (one host round trip less).
    More than 4 096 single-point runs in ONE bucket make that test fail AFTER the counting sort's kernels were enqueued (they must
    have guarded themselves: `k_cs_rank` / `k_cs_emit` read the test on the device); the grid then sorts again with the library
    radix sort and must still be the oracle's, bit for bit -- and a well-behaved cloud right afterwards as well (same arena)."""
    pts = np.tile([-1.0 + 0.005, -2.0 + 0.005, 0.5], (42_000, 1))
    pts[:, 2] += 0.01 * (np.arange(42_000) % 7) + 0.005            # seven voxels of one column, visited in turn: 6 000 runs per voxel
    col = np.random.default_rng(1).random(pts.shape)
    got = r3d.cloud_ops.voxel_down_sample(pts, 0.01, col)
    want = co.voxel_down_sample(pts, 0.01, colors=col)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    assert len(got[0]) == 7
    ok = np.random.default_rng(2).random((50_000, 3))
    np.testing.assert_array_equal(r3d.cloud_ops.voxel_down_sample(ok, 0.02)[0], co.voxel_down_sample(ok, 0.02))


def test_normals_on_degenerate_neighbourhoods_follow_the_closed_form(r3d):
    """The branches of FastEigen3x3 that scanned surfaces rarely take, kernel against oracle (same formulas, same fused
    multiply-adds): exactly planar lattices (a zero eigenvalue, two equal ones), axis-aligned point sets whose covariance is
    exactly diagonal (the `norm == 0` branch picks an axis), coincident points (zero covariance -> (0,0,1)), fewer than three
    neighbours, and collinear points (the normal is any unit vector orthogonal to the line: only that is asserted)."""
    g = np.arange(12) * 0.01
    plane = np.stack(np.meshgrid(g, g, [0.25]), -1).reshape(-1, 3)                           # lattice in z = 0.25
    n = r3d.cloud_ops.estimate_normals(plane, 0.025, 30)
    w = co.estimate_normals_hybrid(plane, 0.025, 30)
    assert np.abs(np.abs(n[:, 2]) - 1).max() < 1e-9 and np.abs(n - w).max() < 1e-9
    tilted = plane @ np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]]).T               # the same lattice on a tilted plane
    n, w = r3d.cloud_ops.estimate_normals(tilted, 0.025, 30), co.estimate_normals_hybrid(tilted, 0.025, 30)
    assert np.abs(n - w).max() < 1e-7 and np.abs(np.abs(n @ np.array([0.6, 0, 0.8])) - 1).max() < 1e-7
    cross = np.array([[0, 0, 0], [0.01, 0, 0], [-0.01, 0, 0], [0, 0.02, 0], [0, -0.02, 0], [0, 0, 0.005], [0, 0, -0.005]])
    n, w = r3d.cloud_ops.estimate_normals(cross, 1.0, 30), co.estimate_normals_hybrid(cross, 1.0, 30)   # exactly diagonal covariance
    np.testing.assert_array_equal(n, w)
    np.testing.assert_array_equal(n[0], [0, 0, 1])                                           # smallest spread along z
    same = np.concatenate([np.tile([[0.3, 0.2, 0.1]], (6, 1)), [[5.0, 5, 5], [5.001, 5, 5]]])
    n, w = r3d.cloud_ops.estimate_normals(same, 0.05, 30), co.estimate_normals_hybrid(same, 0.05, 30)
    # fewer than 3 neighbours: (0,0,1) on both sides.  Six COINCIDENT points: with fused multiply-adds E[xx] - E[x]^2 is a
    # rounding residue (1e-18), not zero, so the closed form returns some unit vector there (as the recording build does)
    np.testing.assert_array_equal(n[6:], w[6:])
    assert (n[6:] == [0, 0, 1]).all() and np.abs(np.linalg.norm(n[:6], axis=1) - 1).max() < 1e-9 and np.isfinite(w).all()
    prev = np.tile([[0.0, 1.0, 0.0]], (len(same), 1))
    kept = r3d.cloud_ops.estimate_normals(same, 0.05, 30, prev_normals=prev)
    np.testing.assert_array_equal(kept[6:], prev[6:])                                        # a cloud with normals keeps them where none can be computed
    assert ((kept[:6] * prev[:6]).sum(1) >= 0).all()                                         # and every other one is turned towards the old one
    line = np.stack([np.linspace(0, 0.05, 9), np.linspace(0, 0.1, 9) * 0.5, np.zeros(9)], 1)
    n = r3d.cloud_ops.estimate_normals(line, 1.0, 30)
    d = (line[-1] - line[0]) / np.linalg.norm(line[-1] - line[0])
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-9 and np.abs(n @ d).max() < 1e-6


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, 65536, 1_000_003])
def test_hand_written_scan_matches_cumsum(r3d, n):
    """k_scan_sums / k_scan_apply around their tile (4096) and vector (16) boundaries, sum and running maximum."""
    rng = np.random.default_rng(n)
    v = rng.integers(0, 50, n).astype(np.int32)
    got = r3d.cloud_ops.debug_exclusive_scan(v)
    want = np.concatenate([[0], np.cumsum(v[:-1], dtype=np.int64)]).astype(np.int32)
    np.testing.assert_array_equal(got, want)
    w = (rng.integers(0, 1000, n) * (rng.random(n) < 0.1)).astype(np.int32)            # mostly zeros, like the per-line "last slot" table
    gotm = r3d.cloud_ops.debug_exclusive_scan(w, maximum=True)
    wantm = np.concatenate([[0], np.maximum.accumulate(w[:-1])]).astype(np.int32) if n > 1 else np.zeros(1, np.int32)
    np.testing.assert_array_equal(gotm, wantm)
