"""GPU tests of the orchestration layer: disparity->cloud, the frame-fusion loop (BASELINE config C4 on fixture
frames) and the single-rank path of the multi-view fusion."""
import os

import numpy as np
import pytest

from oracle import cloud_oracle as co
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
INTR = co.read_intrinsics(os.path.join(GOLDEN, "camera_intrinsic.json"))


def test_reproject_disparity_matches_formula(r3d):
    Q = np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))["Q"]
    rng = np.random.default_rng(0)
    disp = rng.integers(-16, 128 * 16, (120, 200)).astype(np.int16)
    disp[:, :16] = -16
    pts, pix = r3d.cloud_ops.reproject_disparity(disp, Q, 0, want_pixels=True)
    v, u = np.nonzero(disp >= 0)
    np.testing.assert_array_equal(pix, v * 200 + u)                                  # row-major order, valid only
    d = disp[v, u] / 16.0
    h = np.stack([u, v, d, np.ones_like(d)], 0).astype(np.float64)
    X = (Q[:, 0:1] * h[0] + Q[:, 1:2] * h[1]) + Q[:, 2:3] * h[2] + Q[:, 3:4] * h[3]
    with np.errstate(divide="ignore", invalid="ignore"):          # disparity 0 reprojects to infinity (W = 0), as in cv2
        want = (X[:3] / X[3]).T
    finite = np.isfinite(want).all(1)
    assert np.abs(pts[finite] - want[finite]).max() <= 1e-9 * np.abs(want[finite]).max()
    empty = r3d.cloud_ops.reproject_disparity(np.full((10, 20), -16, np.int16), Q, 0)
    assert empty.shape == (0, 3)


def _c4_frames():
    return [co.backproject(co.read_png16(os.path.join(GOLDEN, f"output84/depth_{i:05d}.png")), INTR)[0] for i in range(8, 16)]


def test_fuse_loop_icp_flavour_all_eight_c4_frames(r3d):
    """BASELINE config C4, main.py:34-54 flavour: recorded frames 8..15 (voxel 0.01), first frame initialises the model, every
    later frame is aligned to the growing model with align_point_clouds(threshold=0.02, voxel_size=0.01, max_iter=100)
    (pointcloud_alignment.py:6-43) and its down-sampled transformed copy is appended; failed captures are skipped."""
    frames = [co.voxel_down_sample(f, 0.01) for f in _c4_frames()]
    feed = [r3d.PointCloud(f) for f in frames]
    feed[3:3] = [None, r3d.PointCloud()]                                # main.py:39,53-54: skipped
    got = r3d.pipeline.fuse(feed, flavour="icp")
    log = []
    want, _ = co.fuse_loop(frames, "icp", threshold=0.02, voxel_size=0.01, max_iter=100, log=log)
    assert got.points.shape == want.shape and len(log) == 7
    assert np.abs(got.points - want).max() < 1e-6                      # bar (north_star): 1e-3 on coordinates


def test_fuse_loop_gicp_flavour_all_eight_c4_frames(r3d):
    """BASELINE config C4, test/GICP1.py:134-155 flavour: frames are tensor-voxel-down-sampled (:71-72) and carry
    Hybrid(0.05, 30) normals (:77); every later frame is registered to the WHOLE model with registration_generalized_icp
    (:99-102), appended with its normals, and the model's normals are re-estimated (:148) keeping their orientation.
    The PRODUCT runs free (its own growing model, never reset); every one of its seven steps is checked against the oracle
    applied to the same input, i.e. to the product's model state downloaded before the step: iteration count, fitness, RMSE,
    transform, appended points, re-estimated normals.
    Against the oracle's OWN free-running loop (co.fuse_loop) the product is then compared up to the first frame where the two
    trajectories part: frame 9 must agree to 1e-8, and they must part no earlier than frame 10.  Beyond that point a comparison
    is not a parity statement: profiles/r03_gicp_sensitivity.json (tools/cpu_gicp_sensitivity.py) shows the oracle loop against
    ITSELF under a one-ulp perturbation of frame 9's transform ending 45 mm apart (bar 1e-3) -- Hybrid(0.05, 30) neighbourhoods
    whose 30th and 31st candidates are equidistant to the last bit change membership, the changed normals move the optimum by
    1e-9 .. 1e-5, single correspondences at the 0.02 m threshold flip, and the stop rule (|d fitness| < 1e-6, i.e. an unchanged
    inlier COUNT) moves the stop iteration."""
    frames = []
    for f in _c4_frames():
        p = co.voxel_down_sample_tensor(f, 0.01)
        frames.append((p, co.estimate_normals_hybrid(p, 0.05, 30)))
    model = r3d.cloud_ops.ResidentModel()
    model.append(frames[0][0], None, frames[0][1])
    for p, n in frames[1:]:
        mp, _, mn = model.download()
        want = co.registration(p, mp, 0.02, mode="gicp", max_iteration=30, target_normals=mn,
                               target_cov=co.covariances_from_normals(mn), source_cov=co.covariances_from_normals(n))
        got = model.register_append(p, 0.02, r3d.cloud_ops.GICP, 30, source_normals=n)
        assert got["iterations"] == want["iterations"] and got["correspondences"] == want["correspondences"]
        assert abs(got["fitness"] - want["fitness"]) < 1e-12 and abs(got["inlier_rmse"] - want["inlier_rmse"]) < 1e-9
        assert np.abs(got["T"] - want["T"]).max() < 1e-8                                  # bar: 1e-3
        model.estimate_normals(0.05, 30)
        mp2, _, mn2 = model.download()
        assert len(mp2) == len(mp) + len(p)
        np.testing.assert_array_equal(mp2[:len(mp)], mp)
        assert np.abs(mp2[len(mp):] - co.transform_points(want["T"], p)).max() < 1e-8     # bar: 1e-3
        prev = np.concatenate([mn, co.transform_points(got["T"], n, rotate_only=True)])
        want_n, covs = co._pca_normals(mp2, co.hybrid_neighbors(mp2, 0.05, 30), prev)
        err = np.abs(mn2 - want_n).max(1)
        # a normal is defined up to the separation of the two smallest eigenvalues of its covariance: where they (nearly)
        # coincide (collinear neighbours at the rim of a depth image) its direction is noise on BOTH sides
        w = np.linalg.eigvalsh(covs)
        gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
        loose = err > 1e-6
        assert loose.mean() < 1e-3 and (gap[loose] < 1e-6).all() and err[gap > 1e-6].max() < 1e-6
    final = model.download()
    model.close()
    # the loop function drives exactly these calls
    glog = []
    got = r3d.pipeline.fuse([r3d.PointCloud(p, normals=n) for p, n in frames], flavour="gicp", log=glog)
    np.testing.assert_array_equal(got.points, final[0])
    np.testing.assert_array_equal(got.normals, final[2])
    # free-running oracle loop: equal up to the first divergent frame, which must not be the first one
    wlog = []
    co.fuse_loop(frames, "gicp", log=wlog)
    assert len(glog) == len(wlog) == 7
    together = 0
    for g, w in zip(glog, wlog):
        if g["iterations"] != w["iterations"] or np.abs(g["T"] - w["T"]).max() > 1e-6:
            break
        together += 1
    assert together >= 1                                                           # frame 9 (and whatever follows it in step)
    assert np.abs(glog[0]["T"] - wlog[0]["T"]).max() < 1e-8 and glog[0]["correspondences"] == wlog[0]["correspondences"]
    n0, n1 = len(frames[0][0]), len(frames[1][0])
    assert np.abs(got.points[n0:n0 + n1] - co.transform_points(wlog[0]["T"], frames[1][0])).max() < 1e-8


def test_incremental_model_voxel_table_equals_full_revoxelisation(r3d):
    """SURVEY section 5 "growing target cloud": the resident model keeps its legacy voxel grid as a table that each aligned frame
    is merged into (full rebuild only when the model's minimum corner moves).  Over 24 frames of the recorded scan the loop must
    give, bit for bit, what the reference-shaped loop gives that re-voxelises every model point for every frame
    (pointcloud_alignment.py:22-23) -- and the incremental path must actually have run."""
    cam = r3d.cloud_ops.depth_camera(INTR)
    depths = [co.read_png16(os.path.join(GOLDEN, f"output84/depth_{i:05d}.png")) for i in range(8, 32)]
    log = []
    a = r3d.pipeline.fuse_depth_frames(depths, cam, log=log)
    vt = log[-1]["voxel_table"]
    assert vt["updates"] >= 5 and vt["rebuilds"] >= 1 and vt["updates"] + vt["rebuilds"] == len(log), vt
    clouds = [r3d.PointCloud(co.backproject(d, INTR)[0]) for d in depths]
    b = r3d.pipeline.fuse(clouds, flavour="icp", resident=False)              # host model: voxel_down_sample of everything, every frame
    np.testing.assert_array_equal(a.points, b.points)
    # the table against the oracle's voxel grid of the final model minus its last frame (what the last alignment was run against)
    n_last = log[-1]["appended"]
    want = co.voxel_down_sample(np.asarray(a.points)[:-n_last], 0.01)
    assert vt["voxels"] == len(want)


def test_resident_model_loop_equals_host_model_loop(r3d):
    """The HBM-resident model (r3d_model_*) must give exactly what the reference-shaped loop over host clouds gives (which
    re-uploads and re-voxelises the whole model per frame), for both flavours, colours included."""
    from PIL import Image
    frames = []
    for i in (8, 9, 10, 11):
        d = co.read_png16(os.path.join(GOLDEN, f"output84/depth_{i:05d}.png"))
        pts, (v, u) = co.backproject(d, INTR)
        col = np.asarray(Image.open(os.path.join(GOLDEN, "output84/color_00008.png")))[v, u] / 255.0
        p, c = co.voxel_down_sample(pts, 0.02, colors=col)
        frames.append((p, c))
    feed = [r3d.PointCloud(p, colors=c) for p, c in frames]
    a = r3d.pipeline.fuse(feed, flavour="icp", voxel_size=0.02, threshold=0.04)
    b = r3d.pipeline.fuse(feed, flavour="icp", voxel_size=0.02, threshold=0.04, resident=False)
    np.testing.assert_array_equal(a.points, b.points)
    np.testing.assert_array_equal(a.colors, b.colors)
    assert a.has_colors() and not a.has_normals() and len(a) > len(frames[0][0])
    feed = [r3d.PointCloud(p, colors=c, normals=co.estimate_normals_hybrid(p, 0.05, 30)) for p, c in frames]
    a = r3d.pipeline.fuse(feed, flavour="gicp")
    b = r3d.pipeline.fuse(feed, flavour="gicp", resident=False)
    np.testing.assert_array_equal(a.points, b.points)
    np.testing.assert_array_equal(a.normals, b.normals)
    np.testing.assert_array_equal(a.colors, b.colors)
    # container semantics of the resident model
    m = r3d.cloud_ops.ResidentModel()
    assert m.size() == (0, False, False)
    m.append(frames[0][0], frames[0][1])
    m.append(frames[1][0])                                   # no colours on the appended cloud: legacy += drops them
    assert m.size() == (len(frames[0][0]) + len(frames[1][0]), False, False)
    with pytest.raises(r3d.R3DError):
        r3d.cloud_ops.ResidentModel().align_append(frames[0][0])          # empty model
    p, c, n = m.download()
    np.testing.assert_array_equal(p, np.concatenate([frames[0][0], frames[1][0]]))
    assert c is None and n is None
    m.close()


def test_view_to_cloud_and_single_rank_multi_view(r3d, synth):
    W, H, D = 640, 480, 64
    Q = r3d.pipeline.scaled_Q(np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))["Q"], W / 960.0, unit=1e-3)
    m = r3d.reference_matcher(numDisparities=D, blockSize=5)
    clouds = {}
    poses = {0: np.eye(4), 1: synth.rigid((0, 1, 0), 0.4, (0.003, -0.002, 0.001))}
    for v in (0, 1):
        L, R, _ = synth.stereo_pair(W, H, D, seed=100 + v)
        clouds[v] = r3d.pipeline.view_to_cloud(L, R, Q, m, voxel=0.004, pose=np.linalg.inv(poses[v]))
        assert clouds[v].has_normals() and len(clouds[v]) > 2000
    fused, Ts = r3d.pipeline.multi_view_fuse(clouds, 2, threshold=0.02)
    assert len(fused) == len(clouds[0]) + len(clouds[1])
    R_err = Ts[1][:3, :3] @ poses[1][:3, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(R_err) - 1) / 2, -1, 1)))
    assert ang < 0.2 and np.abs(Ts[1][:3, 3] - poses[1][:3, 3]).max() < 2e-3


def test_device_resident_view_chain_equals_host_chain_and_oracle(r3d, synth):
    """r3d_disparity_to_cloud_dev (SGM map stays in HBM -> reprojection -> depth filter -> pose -> voxel grid -> normals) must
    give exactly what chaining the host-buffer entry points gives, and that equals the oracle chain."""
    from oracle import sgbm_oracle as so
    W, H, D = 512, 300, 64
    Q = r3d.pipeline.scaled_Q(np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))["Q"], W / 960.0, unit=1e-3)
    m = r3d.reference_matcher(numDisparities=D, blockSize=5)
    L, R, _ = synth.stereo_pair(W, H, D, seed=77)
    pose = synth.rigid((0.2, 1, 0.1), 3.0, (0.01, -0.02, 0.005))
    kw = dict(voxel=0.004, pose=pose, max_depth=0.6, max_nn=20)
    a = r3d.pipeline.view_to_cloud(L, R, Q, m, device_resident=True, **kw)
    b = r3d.pipeline.view_to_cloud(L, R, Q, m, device_resident=False, **kw)
    assert len(a) > 1000
    np.testing.assert_array_equal(a.points, b.points)
    np.testing.assert_array_equal(a.normals, b.normals)
    # oracle chain on the oracle's disparity map
    kwm = dict(minDisparity=0, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=15, speckleWindowSize=0,
               speckleRange=2, preFilterCap=63)
    disp = so.compute(L, R, so.make_params(numDisparities=D, **kwm), nthreads=4)
    ys, xs = np.nonzero(disp >= 0)
    v = np.stack([xs, ys, disp[ys, xs] / 16.0, np.ones(len(xs))], 0).astype(np.float64)
    X = Q @ v
    pts = (X[:3] / X[3]).T
    pts = pts[np.abs(pts[:, 2]) <= 0.6]
    pts = co.transform_points(pose, pts)
    pts = co.voxel_down_sample(pts, 0.004)
    order = lambda p: p[np.lexsort(p.T[::-1])]                                           # noqa: E731
    np.testing.assert_allclose(order(a.points), order(pts), atol=1e-9)
    # options off: raw reprojection only
    ctx = m.context
    d_l, d_r, d_d = ctx.to_device(L), ctx.to_device(R), ctx.alloc(W * H * 2)
    m.compute_device(d_l, d_r, W, H, W, d_d)
    raw, nrm = r3d.cloud_ops.disparity_to_cloud_device(d_d, W, H, Q, 0, None, None, 0, None, 0, ctx=ctx)
    assert nrm is None and len(raw) == int((disp >= 0).sum())
    np.testing.assert_array_equal(raw, r3d.cloud_ops.reproject_disparity(disp, Q, 0))
    with pytest.raises(r3d.R3DError):
        r3d.cloud_ops.disparity_to_cloud_device(d_d, W, H, Q, 0, None, None, 0, None, 0, capacity=10, ctx=ctx)
    for p in (d_l, d_r, d_d):
        ctx.free(p)


# ---- BASELINE config C5 at its real size: the device-tensor chain and the exchange path against the oracle -------------------

C5_W, C5_H, C5_D = 3264, 2448, 128
C5_KW = dict(minDisparity=0, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=15, speckleWindowSize=0,
             speckleRange=2, preFilterCap=63)                                  # Calib_depth/depth2.py:139-158


def _c5_pose(synth, v):
    return np.eye(4) if v == 0 else synth.rigid((0.2 * v, 1.0, 0.1 * (v % 3)), 0.25 + 0.05 * v, (0.002 + 0.0005 * v, -0.0015, 0.001 * (v % 4)))


def _c5_oracle_view(L, R, Q, pose_inv):
    """oracle chain of one view: StereoSGBM 3-way -> reprojectImageTo3D -> |z| <= 3 -> pose -> legacy voxel grid 0.01 ->
    Hybrid(0.02, 30) normals (what pipeline.view_to_cloud_tensors does on the device)."""
    from oracle import sgbm_oracle as so
    disp = so.compute(L, R, so.make_params(numDisparities=C5_D, **C5_KW), nthreads=4)
    ys, xs = np.nonzero(disp >= 0)
    hv = np.stack([xs, ys, disp[ys, xs] / 16.0, np.ones(len(xs))], 0).astype(np.float64)
    X = (Q[:, 0:1] * hv[0] + Q[:, 1:2] * hv[1]) + Q[:, 2:3] * hv[2] + Q[:, 3:4] * hv[3]
    with np.errstate(divide="ignore", invalid="ignore"):
        pts = (X[:3] / X[3]).T
    pts = pts[np.isfinite(pts).all(1)]
    pts = pts[np.abs(pts[:, 2]) <= 3.0]
    pts = co.transform_points(pose_inv, pts)
    pts = co.voxel_down_sample(pts, 0.01)
    return disp, pts, co.estimate_normals_hybrid(pts, 0.02, 30)


def _lex(p):
    return np.lexsort(p.T[::-1])


@pytest.fixture(scope="module")
def c5_views(r3d, synth):
    """Two full 8 MP views of the C5 batch through the device-tensor chain (pipeline.view_to_cloud_tensors), on torch's DEFAULT
    stream (so the null-stream bridging of distributed.shared_stream is what orders the library against torch), plus the
    same two views through the pipelined several-views-per-GPU form (pipeline.views_to_cloud_tensors)."""
    import torch
    Q = r3d.pipeline.scaled_Q(np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))["Q"], C5_W / 960.0, unit=1e-3)
    m = r3d.StereoSGBM_create(numDisparities=C5_D, mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY, **C5_KW)
    imgs, seq, cap = {}, {}, 1 << 20
    d_disp = torch.empty(C5_W * C5_H, dtype=torch.int16, device="cuda")
    for v in (0, 1):
        L, R, _ = synth.stereo_pair(C5_W, C5_H, C5_D, seed=20241008 + v)
        imgs[v] = (L, R, torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
        buf = torch.empty((2, cap, 3), dtype=torch.float64, device="cuda")
        seq[v] = r3d.pipeline.view_to_cloud_tensors(imgs[v][2].data_ptr(), imgs[v][3].data_ptr(), d_disp.data_ptr(), C5_W, C5_H, Q, m, buf,
                                                    voxel=0.01, max_nn=30, max_depth=3.0, pose=np.linalg.inv(_c5_pose(synth, v)))
    ctx2 = [r3d.Context(m.context.device), r3d.Context(m.context.device)]     # two cloud contexts: one host thread each
    disps = [torch.empty(C5_W * C5_H, dtype=torch.int16, device="cuda") for _ in (0, 1)]
    bufs = [torch.empty((2, cap, 3), dtype=torch.float64, device="cuda") for _ in (0, 1)]
    piped = r3d.pipeline.views_to_cloud_tensors([(imgs[v][2].data_ptr(), imgs[v][3].data_ptr()) for v in (0, 1)], [d.data_ptr() for d in disps],
                                                C5_W, C5_H, Q, m, bufs, ctx2, voxel=0.01, max_nn=30, max_depth=3.0,
                                                poses=[np.linalg.inv(_c5_pose(synth, v)) for v in (0, 1)])
    torch.cuda.synchronize()
    for c in ctx2:
        c.close()
    return dict(Q=Q, imgs=imgs, seq=seq, piped=piped)


def test_c5_view_chain_at_8mp_equals_the_oracle_chain(r3d, synth, c5_views):
    """r3d_sgbm_compute_dev -> r3d_disparity_to_cloud_resident on torch-owned device memory at the C5 size,
    compared with the oracle chain on the same pair: same voxels (coordinates <= 1e-9, bar 1e-3), normals <= 1e-6 sign-agnostic
    (bar 1e-3).  The pipelined several-views form must give bit-identical tensors."""
    import torch
    for v in (0, 1):
        a, b = c5_views["seq"][v], c5_views["piped"][v]
        assert a.shape == b.shape and torch.equal(a, b)
    L, R = c5_views["imgs"][1][:2]
    _, want_p, want_n = _c5_oracle_view(L, R, c5_views["Q"], np.linalg.inv(_c5_pose(synth, 1)))
    got = c5_views["seq"][1].cpu().numpy()
    assert got.shape[1] == len(want_p) and len(want_p) > 50_000
    ia, ib = _lex(got[0]), _lex(want_p)
    assert np.abs(got[0][ia] - want_p[ib]).max() <= 1e-9
    gn, wn = got[1][ia], want_n[ib]
    err = np.minimum(np.abs(gn - wn).max(1), np.abs(gn + wn).max(1))
    # a PCA normal is defined up to the separation of the two smallest eigenvalues (see the GICP-flavour loop test above)
    assert (err > 1e-6).mean() < 1e-3 and np.median(err) < 1e-9


def test_c5_exchange_and_registration_under_rccl_equal_the_oracle(r3d, synth, c5_views, monkeypatch):
    """VERDICT r2 item 1c + ADVICE r2 (stream ordering): two 8 MP views through pipeline.multi_view_fuse_tensors with the RCCL
    collectives really issued (one-rank group, R3D_FORCE_DIST=1), torch on its default stream.  T of view 1 must equal the
    oracle's registration_generalized_icp on the same clouds (<= 1e-8, bar 1e-3), and the FUSED VALUES must equal T applied to
    the input clouds on the host (a fused cloud read before the asynchronous transform kernels finished would differ)."""
    import torch
    import torch.distributed as dist
    monkeypatch.setenv("R3D_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    import socket
    with socket.socket() as sk:                                     # a free port (a fixed one may be taken on a shared host)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert torch.cuda.current_stream().cuda_stream == 0            # the default stream: the case round 2 got wrong
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        local = {v: c5_views["seq"][v] for v in (0, 1)}
        tm = {}
        fused, Ts = r3d.pipeline.multi_view_fuse_tensors(local, 2, threshold=0.02, mode=r3d.cloud_ops.GICP, max_iteration=30, timings=tm)
        got = fused.cpu().numpy()                                   # default stream again: must be ordered after the fuse kernels
    finally:
        dist.destroy_process_group()
    src, tgt = (c5_views["seq"][v].cpu().numpy() for v in (1, 0))
    want = co.registration(src[0], tgt[0], 0.02, mode="gicp", max_iteration=30, target_normals=tgt[1],
                           target_cov=co.covariances_from_normals(tgt[1]), source_cov=co.covariances_from_normals(src[1]))
    assert np.array_equal(Ts[0], np.eye(4)) and np.abs(Ts[1] - want["T"]).max() < 1e-8
    assert np.abs(Ts[1] - _c5_pose(synth, 1)).max() < 1e-3          # and it is the right answer
    n0 = tgt.shape[1]
    np.testing.assert_array_equal(got[:, :n0], tgt)                 # view 0: identity
    assert np.abs(got[0, n0:] - co.transform_points(Ts[1], src[0])).max() < 1e-12
    assert np.abs(got[1, n0:] - co.transform_points(Ts[1], src[1], rotate_only=True)).max() < 1e-12
    assert set(tm) >= {"exchange_ms", "register_ms", "fuse_ms"}


def test_multi_view_fuse_host_front_end_values_on_default_stream(r3d, synth, c5_views):
    """pipeline.multi_view_fuse (default context, torch's default stream) -- the fused VALUES, not just the counts."""
    clouds = {}
    for v in (0, 1):
        a = c5_views["seq"][v].cpu().numpy()
        clouds[v] = r3d.PointCloud(a[0][:60000].copy(), normals=a[1][:60000].copy())
    fused, Ts = r3d.pipeline.multi_view_fuse(clouds, 2, threshold=0.02)
    n0 = len(clouds[0])
    np.testing.assert_array_equal(fused.points[:n0], clouds[0].points)
    assert np.abs(fused.points[n0:] - co.transform_points(Ts[1], clouds[1].points)).max() < 1e-12
    assert np.abs(fused.normals[n0:] - co.transform_points(Ts[1], clouds[1].normals, rotate_only=True)).max() < 1e-12
