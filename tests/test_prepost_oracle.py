"""CPU tests of oracle/prepost_oracle.py (rectification maps, remap, BGR2GRAY, WLS filter, normalize): known answers
and algebraic properties.  OpenCV is absent, so these pin the restatement to hand-derived values only (parity unpinned)."""
import os

import numpy as np
import pytest

from oracle import prepost_oracle as po
from tests.conftest import GOLDEN


def _calib():
    return np.load(os.path.join(GOLDEN, "jetson_stereo_8MP_stereo.npz"))


def test_rectify_map_identity_and_quarter_pixel_shift():
    K = np.array([[500.0, 0, 31.0], [0, 400.0, 17.0], [0, 0, 1]])
    m1, m2 = po.init_undistort_rectify_map(K, None, None, K, (64, 40))
    xs, ys = np.meshgrid(np.arange(64), np.arange(40))
    np.testing.assert_array_equal(m1[..., 0], xs)
    np.testing.assert_array_equal(m1[..., 1], ys)
    assert not m2.any()
    P = K.copy()
    P[0, 2] += 0.25                                  # destination principal point moved right: source x = x - 0.25
    m1, m2 = po.init_undistort_rectify_map(K, None, None, P, (64, 40))
    np.testing.assert_array_equal(m1[..., 0], xs - 1)
    np.testing.assert_array_equal(m2, np.full((40, 64), 24, np.uint16))      # fraction 24/32 in x, 0 in y


def test_rectify_map_calibration_file_matches_direct_evaluation():
    """The running-sum walk must agree with evaluating the same model from closed-form grid coordinates (independent
    code, float64): equal up to one 1/32-pixel step where rounding sits on the fence."""
    c = _calib()
    size = (960, 540)
    m1, m2 = po.init_undistort_rectify_map(c["mtx1"], c["dist1"], c["R1"], c["P1"], size)
    iR = np.linalg.inv(c["P1"][:, :3] @ c["R1"])
    xs, ys = np.meshgrid(np.arange(size[0], dtype=np.float64), np.arange(size[1], dtype=np.float64))
    v = iR @ np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    x, y = v[0] / v[2], v[1] / v[2]
    k1, k2, p1, p2, k3 = c["dist1"].ravel()[:5]
    r2 = x * x + y * y
    kr = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    u = (c["mtx1"][0, 0] * xd + c["mtx1"][0, 2]).reshape(xs.shape)
    vv = (c["mtx1"][1, 1] * yd + c["mtx1"][1, 2]).reshape(xs.shape)
    iu = m1[..., 0].astype(np.int64) * 32 + (m2 & 31)
    iv = m1[..., 1].astype(np.int64) * 32 + (m2 >> 5)
    assert np.abs(iu - np.rint(u * 32)).max() <= 1 and np.abs(iv - np.rint(vv * 32)).max() <= 1
    assert (iu == np.rint(u * 32)).mean() > 0.999


def test_bilinear_table_sums_and_saturated_entry():
    t = po.bilinear_tab_i()
    assert (t.sum(1) == 32768).all()
    assert t[0].tolist() == [32767, 0, 0, 1]
    assert t[16 * 32 + 16].tolist() == [8192, 8192, 8192, 8192]
    assert t[31].tolist() == [1024, 31744, 0, 0]


def test_remap_identity_shift_half_pixel_and_border():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    xs, ys = np.meshgrid(np.arange(30), np.arange(20))
    m1 = np.stack([xs, ys], -1).astype(np.int16)
    m2 = np.zeros((20, 30), np.uint16)
    np.testing.assert_array_equal(po.remap_fixed(img, m1, m2), img)                     # (32767,0,0,1) weights: identity
    m1s = m1.copy()
    m1s[..., 0] += 3
    out = po.remap_fixed(img, m1s, m2)
    np.testing.assert_array_equal(out[:, :27], img[:, 3:])
    assert not out[:, 27:].any()                                                        # BORDER_CONSTANT 0
    half = np.full((20, 30), 16, np.uint16)                                             # fx = 16/32, fy = 0
    out = po.remap_fixed(img[..., 0], m1, half)
    a = img[..., 0].astype(int)
    want = (a[:, :-1] + a[:, 1:] + 1) >> 1
    np.testing.assert_array_equal(out[:, :-1], want)
    np.testing.assert_array_equal(out[:, -1], (a[:, -1] + 1) >> 1)                      # right tap outside -> 0


def test_bgr2gray_known_values():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 90]]], np.uint8)
    assert po.bgr2gray(px).tolist() == [[29, 150, 76, 255, 0, (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14]]


def test_fgs_identity_constant_and_smoothing():
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, (24, 31), dtype=np.uint8)
    ch, cv = po.fgs_weights(g, po.fgs_lut(10.0))
    assert (ch <= 0).all() and (ch[:, -1] == 0).all() and (cv[-1] == 0).all() and ch.min() >= -1
    src = rng.normal(size=(24, 31)).astype(np.float32)
    np.testing.assert_array_equal(po.fgs_filter(src, ch, cv, 0.0), src)                 # lambda 0: identity
    const = np.full((24, 31), 7.25, np.float32)
    np.testing.assert_allclose(po.fgs_filter(const, ch, cv, 500.0), const, rtol=2e-5)    # rows of (I + lam*A) sum to 1
    flat = np.zeros((24, 31), np.uint8)
    chf, cvf = po.fgs_weights(flat, po.fgs_lut(1.5))
    sm = po.fgs_filter(src, chf, cvf, 50.0)
    assert sm.std() < 0.2 * src.std() and abs(sm.mean() - src.mean()) < 0.05            # strong smoothing, mean kept


def test_fgs_pass_solves_its_tridiagonal_system():
    rng = np.random.default_rng(2)
    C = -rng.random((3, 17)).astype(np.float32)
    C[:, -1] = 0
    f = rng.normal(size=(3, 17)).astype(np.float32)
    lam = 30.0
    u = po._fgs_pass(f.copy(), C, lam).astype(np.float64)
    a = np.concatenate([np.zeros((3, 1)), lam * C[:, :-1].astype(np.float64)], 1)
    c = lam * C.astype(np.float64)
    lhs = (1 - a - c) * u
    lhs[:, 1:] += a[:, 1:] * u[:, :-1]
    lhs[:, :-1] += c[:, :-1] * u[:, 1:]
    np.testing.assert_allclose(lhs, f, atol=2e-4)


def test_wls_constant_planes_and_roi_fill():
    H, W, D, d0 = 40, 120, 32, 9
    dl = np.full((H, W), 16 * d0, np.int16)
    dr = np.full((H, W), -16 * d0, np.int16)
    g = np.random.default_rng(3).integers(0, 256, (H, W), dtype=np.uint8)
    out, conf = po.wls_filter(dl, g, dr, 0, D, 5, return_confidence=True)
    assert (out[:, :D] == -16).all()
    np.testing.assert_array_equal(out[:, D:], dl[:, D:])
    assert (conf[:, D:] == 255).all() and (conf[:, :D] == 0).all()


def test_wls_fills_inconsistent_pixels_from_confident_neighbours():
    H, W, D, d0 = 30, 100, 16, 5
    dl = np.full((H, W), 16 * d0, np.int16)
    dr = np.full((H, W), -16 * d0, np.int16)
    dl[10:14, 50:56] = -16                                    # a hole of invalid disparities
    g = np.full((H, W), 100, np.uint8)
    out, conf = po.wls_filter(dl, g, dr, 0, D, 5, return_confidence=True)
    assert (conf[10:14, 50:56] == 0).all()
    assert np.abs(out[10:14, 50:56].astype(int) - 16 * d0).max() <= 1      # filled from the surrounding plane
    assert np.abs(out[:, D:].astype(int) - 16 * d0).max() <= 1


def test_normalize_minmax_known_values():
    a = np.array([[0, 50, 100], [25, 75, 100]], np.int16)
    assert po.normalize_minmax(a).tolist() == [[0, 128, 255], [64, 191, 255]]          # 127.5 -> 128, 63.75, 191.25
    assert po.normalize_minmax(np.full((2, 2), 7, np.int16)).tolist() == [[0, 0], [0, 0]]
    b = np.array([-16, 0, 2032], np.int16)
    assert po.normalize_minmax(b).tolist() == [0, 2, 255]
