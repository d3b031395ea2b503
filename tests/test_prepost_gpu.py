"""GPU parity tests of the stages either side of the matcher (csrc/prepost.hip through the C ABI) against
oracle/prepost_oracle.py: integer / fixed-point stages bit-exact, the float32 WLS filter within the stated tolerance."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _po():
    from oracle import prepost_oracle as po
    return po


def _calib():
    return np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))


def _color_pair(synth, W, H, D, seed):
    """BGR frames whose grey image is a textured stereo pair: three differently scaled copies + per-channel noise."""
    L, R, _ = synth.stereo_pair(W, H, D, seed=seed)
    rng = np.random.default_rng(seed)

    def col(g):
        f = np.stack([g * 0.8, g * 1.0, g * 0.9], -1) + rng.integers(0, 12, g.shape + (3,))
        return np.clip(f, 0, 255).astype(np.uint8)
    return col(L.astype(np.float64)), col(R.astype(np.float64))


@pytest.mark.parametrize("which", ["1", "2"])
def test_rectify_maps_bit_exact_calibration_file(r3d, which):
    c = _calib()
    size = (960, 540)                                         # the resolution the file was calibrated at
    a1, a2 = r3d.initUndistortRectifyMap(c["mtx" + which], c["dist" + which], c["R" + which], c["P" + which], size, r3d.CV_16SC2)
    b1, b2 = _po().init_undistort_rectify_map(c["mtx" + which], c["dist" + which], c["R" + which], c["P" + which], size)
    np.testing.assert_array_equal(a1, b1)
    np.testing.assert_array_equal(a2, b2)


def test_rectify_maps_rational_and_prism_terms_no_R(r3d):
    K = np.array([[800.0, 0, 320.5], [0, 790.0, 239.25], [0, 0, 1]])
    dist = np.array([0.11, -0.21, 1e-3, -2e-3, 0.05, 0.01, -0.02, 0.003, 1e-3, -1e-3, 2e-3, 5e-4])
    P = np.array([[760.0, 0, 330.0], [0, 760.0, 245.0], [0, 0, 1]])
    a1, a2 = r3d.initUndistortRectifyMap(K, dist, None, P, (641, 479))
    b1, b2 = _po().init_undistort_rectify_map(K, dist, None, P, (641, 479))
    np.testing.assert_array_equal(a1, b1)
    np.testing.assert_array_equal(a2, b2)
    with pytest.raises(r3d.R3DError):
        r3d.initUndistortRectifyMap(K, np.r_[dist, 0.01, 0.0], None, P, (64, 48))       # tilted sensor: refused, not ignored


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_remap_bit_exact_random_maps_with_border(r3d, cn):
    rng = np.random.default_rng(10 + cn)
    sh, sw, dh, dw = 97, 131, 80, 150
    src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
    m1 = np.stack([rng.integers(-4, sw + 4, (dh, dw)), rng.integers(-4, sh + 4, (dh, dw))], -1).astype(np.int16)
    m2 = rng.integers(0, 1024, (dh, dw)).astype(np.uint16)
    m2[::7, ::5] = 0                                                   # the saturated (32767,0,0,1) table entry
    np.testing.assert_array_equal(r3d.remap(src, m1, m2, r3d.INTER_LINEAR), _po().remap_fixed(src, m1, m2))
    np.testing.assert_array_equal(r3d.remap(src, m1, m2, borderValue=77), _po().remap_fixed(src, m1, m2, border_value=77))


def test_rectify_and_gray_fused_equals_two_steps(r3d, synth):
    c = _calib()
    size = (960, 540)
    m1, m2 = r3d.initUndistortRectifyMap(c["mtx1"], c["dist1"], c["R1"], c["P1"], size)
    frame, _ = _color_pair(synth, 960, 540, 64, 5)
    rect, gray = r3d.remap(frame, m1, m2, with_gray=True)
    po = _po()
    want = po.remap_fixed(frame, m1, m2)
    np.testing.assert_array_equal(rect, want)
    np.testing.assert_array_equal(gray, po.bgr2gray(want))
    np.testing.assert_array_equal(r3d.cvtColor(rect, r3d.COLOR_BGR2GRAY), gray)
    assert (rect.reshape(-1, 3).any(1)).mean() > 0.5                   # the calibration really maps into the frame


def test_bgr2gray_all_levels_and_strided_input(r3d):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (64, 300, 3), dtype=np.uint8)
    np.testing.assert_array_equal(r3d.cvtColor(img), _po().bgr2gray(img))
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], np.uint8)
    assert r3d.cvtColor(px).tolist() == [[29, 150, 76, 255]]
    bgra = rng.integers(0, 256, (10, 33, 4), dtype=np.uint8)
    np.testing.assert_array_equal(r3d.cvtColor(bgra), _po().bgr2gray(bgra))
    with pytest.raises(ValueError):
        r3d.cvtColor(img, code=4)


def test_normalize_minmax_bit_exact(r3d):
    rng = np.random.default_rng(4)
    a = rng.integers(-16, 2048, (211, 307)).astype(np.int16)
    np.testing.assert_array_equal(r3d.normalize(a, None, 0, 255, r3d.NORM_MINMAX), _po().normalize_minmax(a))
    np.testing.assert_array_equal(r3d.normalize(a, None, 10, 200), _po().normalize_minmax(a, 10, 200))
    c = np.full((5, 7), -3, np.int16)
    assert not r3d.normalize(c).any()


def _disparities(r3d, L, R, D, bs=5):
    left = r3d.reference_matcher(numDisparities=D, blockSize=bs)
    right = r3d.createRightMatcher(left)
    wls = r3d.createDisparityWLSFilter(left)
    assert left.getUniquenessRatio() == 0 and left.getDisp12MaxDiff() == 1000000 and left.getSpeckleWindowSize() == 0
    return left.compute(L, R), right.compute(R, L), wls


@pytest.mark.parametrize("W,H,D,guide_cn", [(333, 121, 32, 1), (640, 200, 64, 1), (257, 190, 48, 3), (95, 33, 16, 1), (160, 64, 32, 1)])
def test_wls_filter_vs_oracle(r3d, synth, W, H, D, guide_cn):
    """float32 pipeline.  Sequential solver (the oracle's operation order): confidence and output identical.  Default
    block-partitioned solver (same systems, different rounding): filtered disparity equal up to one LSB (1/16 px) on
    at most 0.1 % of the pixels."""
    L, R, _ = synth.stereo_pair(W, H, D, seed=W)
    dl, dr, wls = _disparities(r3d, L, R, D)
    wls.setLambda(8000)
    wls.setSigmaColor(1.5)
    guide = L if guide_cn == 1 else np.stack([L, np.roll(L, 1, 1), L[::-1]], -1).copy()
    want, wconf = _po().wls_filter(dl, guide, dr, 0, D, 5, lam=8000, sigma_color=1.5, return_confidence=True)
    got = wls.filter(dl, guide, None, dr)
    np.testing.assert_array_equal(wls.getConfidenceMap(), wconf)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert (got[:, :D] == -16).all() and got.dtype == np.int16 and wls.getROI() == (D, 0, W - D, H)
    wls.solver = r3d.stereo_prepost.SOLVER_SEQUENTIAL
    np.testing.assert_array_equal(wls.filter(dl, guide, None, dr), want)


def test_wls_filter_documented_bound_at_8mp(r3d, synth):
    """The contract of R3D_WLS_SOLVER_PARTITIONED written in include/r3d.h (depth2.py:164-166,255 at the C2 frame size): the
    confidence map is bit-exact; the filtered int16 map differs from the sequential float32 operation order (the oracle's, and
    R3D_WLS_SOLVER_SEQUENTIAL's) by at most 1 LSB (1/16 px) on fewer than 0.1 % of the pixels, never outside the ROI."""
    W, H, D = 3264, 2448, 128
    L, R, _ = synth.stereo_pair(W, H, D, seed=5)
    dl, dr, wls = _disparities(r3d, L, R, D)
    wls.setLambda(8000)
    wls.setSigmaColor(1.5)
    want, wconf = _po().wls_filter(dl, L, dr, 0, D, 5, lam=8000, sigma_color=1.5, return_confidence=True)
    got = wls.filter(dl, L, None, dr)
    np.testing.assert_array_equal(wls.getConfidenceMap(), wconf)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert (diff[:, :D] == 0).all()
    wls.solver = r3d.stereo_prepost.SOLVER_SEQUENTIAL
    np.testing.assert_array_equal(wls.filter(dl, L, None, dr), want)


def test_wls_negative_min_disparity_roi_and_accessors(r3d, synth):
    W, H, D, minD = 300, 90, 32, -8
    L, R, _ = synth.stereo_pair(W, H, D + minD, seed=9)
    left = r3d.StereoSGBM_create(minDisparity=minD, numDisparities=D, blockSize=7, P1=8 * 3 * 49, P2=32 * 3 * 49, preFilterCap=63,
                                 mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY)
    right = r3d.createRightMatcher(left)
    wls = r3d.createDisparityWLSFilter(left)
    assert wls.getDepthDiscontinuityRadius() == 4 and wls.getLRCthresh() == 24 and wls.getLambda() == 8000.0
    wls.setSigmaColor(2.5)
    wls.setLambda(500)
    dl, dr = left.compute(L, R), right.compute(R, L)
    got = wls.filter(dl, L, None, dr)
    want = _po().wls_filter(dl, L, dr, minD, D, 7, lam=500, sigma_color=2.5)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    fill = 16 * (minD - 1)
    assert (got[:, :D + minD] == fill).all() and (got[:, W + minD:] == fill).all()
    with pytest.raises(ValueError):
        wls.filter(dl, L, None, None)


def test_depth2_frame_loop_end_to_end(r3d, synth):
    """One iteration of Calib_depth/depth2.py:243-257 on synthetic colour frames: remap -> cvtColor -> left and right
    matcher -> WLS filter -> normalize, each stage fed with the previous stage's GPU output, against the same chain of
    oracles."""
    from oracle import sgbm_oracle as so
    po = _po()
    c = _calib()
    W, H, D = 960, 540, 64
    fl, fr = _color_pair(synth, W, H, D, 17)
    maps = [r3d.initUndistortRectifyMap(c["mtx" + k], c["dist" + k], c["R" + k], c["P" + k], (W, H), r3d.CV_16SC2) for k in "12"]
    rl = r3d.remap(fl, *maps[0], r3d.INTER_LINEAR)
    rr = r3d.remap(fr, *maps[1], r3d.INTER_LINEAR)
    gl, gr = r3d.cvtColor(rl, r3d.COLOR_BGR2GRAY), r3d.cvtColor(rr, r3d.COLOR_BGR2GRAY)
    left = r3d.reference_matcher(numDisparities=D, blockSize=5)
    right = r3d.createRightMatcher(left)
    wls = r3d.createDisparityWLSFilter(matcher_left=left)
    wls.setLambda(8000)
    wls.setSigmaColor(1.5)
    dl = left.compute(gl, gr).astype(np.int16)
    dr = right.compute(gr, gl).astype(np.int16)
    filt = wls.filter(dl, gl, None, dr)
    vis = np.uint8(r3d.normalize(filt, None, 0, 255, r3d.NORM_MINMAX))

    ogl = po.bgr2gray(po.remap_fixed(fl, *maps[0]))
    ogr = po.bgr2gray(po.remap_fixed(fr, *maps[1]))
    np.testing.assert_array_equal(gl, ogl)
    np.testing.assert_array_equal(gr, ogr)
    kw = dict(minDisparity=0, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1000000, uniquenessRatio=0, speckleWindowSize=0,
              speckleRange=2, preFilterCap=63)
    odl = so.compute(ogl, ogr, so.make_params(numDisparities=D, **kw), nthreads=8)
    kwr = dict(kw, minDisparity=-D + 1)
    odr = so.compute(ogr, ogl, so.make_params(numDisparities=D, **kwr), nthreads=8)
    np.testing.assert_array_equal(dl, odl)
    np.testing.assert_array_equal(dr, odr)
    ofilt = po.wls_filter(odl, ogl, odr, 0, D, 5, lam=8000, sigma_color=1.5)
    diff = np.abs(filt.astype(int) - ofilt.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    np.testing.assert_array_equal(vis, np.uint8(po.normalize_minmax(filt)))


def test_config_c1_depth1_chain_640x480(r3d, synth):
    """BASELINE config C1: one 640x480 pair through the depth1.py chain with the matcher that script builds
    (initialize_stereo_matcher_sgbm defaults, depth1.py:185-222: numDisparities 16, blockSize 5, P1 = 8*3*25, P2 = 32*3*25,
    3-way mode, createRightMatcher, WLS lambda 8000 / sigma 1.5) and the rectification of jetson_stereo_8MP_stereo.npz:
    initUndistortRectifyMap -> remap -> cvtColor -> left / right matcher -> WLS -> the int16 map depth1.py:331-337 displays."""
    from oracle import sgbm_oracle as so
    po = _po()
    c = _calib()
    W, H, D, bs = 640, 480, 16, 5
    fl, fr = _color_pair(synth, W, H, D, 41)
    s = W / 960.0                                             # the file was calibrated at 960x540: scale the pinhole parameters
    def scaled(K):
        K = np.array(K, dtype=np.float64)
        K[:2] *= s
        return K
    maps = [r3d.initUndistortRectifyMap(scaled(c["mtx" + k]), c["dist" + k], c["R" + k], scaled(c["P" + k]), (W, H), r3d.CV_16SC2) for k in "12"]
    for k, (m1, m2) in zip("12", maps):
        w1, w2 = po.init_undistort_rectify_map(scaled(c["mtx" + k]), c["dist" + k], c["R" + k], scaled(c["P" + k]), (W, H))
        np.testing.assert_array_equal(m1, w1)
        np.testing.assert_array_equal(m2, w2)
    gl = r3d.cvtColor(r3d.remap(fl, *maps[0], r3d.INTER_LINEAR), r3d.COLOR_BGR2GRAY)
    gr = r3d.cvtColor(r3d.remap(fr, *maps[1], r3d.INTER_LINEAR), r3d.COLOR_BGR2GRAY)
    ogl, ogr = po.bgr2gray(po.remap_fixed(fl, *maps[0])), po.bgr2gray(po.remap_fixed(fr, *maps[1]))
    np.testing.assert_array_equal(gl, ogl)
    np.testing.assert_array_equal(gr, ogr)
    kw = dict(minDisparity=0, blockSize=bs, P1=8 * 3 * bs ** 2, P2=32 * 3 * bs ** 2, disp12MaxDiff=1, uniquenessRatio=15,
              speckleWindowSize=0, speckleRange=2, preFilterCap=63)
    left = r3d.StereoSGBM_create(numDisparities=D, mode=r3d.STEREO_SGBM_MODE_SGBM_3WAY, **kw)
    plain = left.compute(gl, gr)                              # the matcher as depth1.py:202-214 creates it, before the WLS filter re-configures it
    np.testing.assert_array_equal(plain, so.compute(ogl, ogr, so.make_params(numDisparities=D, **kw), nthreads=4))
    right = r3d.createRightMatcher(left)
    wls = r3d.createDisparityWLSFilter(matcher_left=left)
    wls.setLambda(8000)
    wls.setSigmaColor(1.5)
    dl, dr = left.compute(gl, gr), right.compute(gr, gl)
    kwl = dict(kw, disp12MaxDiff=1000000, uniquenessRatio=0)
    odl = so.compute(ogl, ogr, so.make_params(numDisparities=D, **kwl), nthreads=4)
    odr = so.compute(ogr, ogl, so.make_params(numDisparities=D, **dict(kwl, minDisparity=-D + 1)), nthreads=4)
    np.testing.assert_array_equal(dl, odl)
    np.testing.assert_array_equal(dr, odr)
    filt = wls.filter(dl, gl, None, dr)
    diff = np.abs(filt.astype(int) - po.wls_filter(odl, ogl, odr, 0, D, bs, lam=8000, sigma_color=1.5).astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    wls.solver = r3d.stereo_prepost.SOLVER_SEQUENTIAL
    np.testing.assert_array_equal(wls.filter(dl, gl, None, dr), po.wls_filter(odl, ogl, odr, 0, D, bs, lam=8000, sigma_color=1.5))
