"""CPU tests of the SGBM oracle: known-answer tests + cross-check against an independent numpy restatement.
(No reference golden vectors exist for this path -- SURVEY.md section 8c: parity unpinned vs real OpenCV.)"""
import numpy as np
import pytest

from oracle import sgbm_oracle as so
from tests import sgbm_numpy_ref as ref

C2_KW = dict(minDisparity=0, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, uniquenessRatio=15,
             speckleWindowSize=0, speckleRange=2, preFilterCap=63)          # Calib_depth/depth2.py:139-158
D4_KW = dict(minDisparity=0, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, uniquenessRatio=10,
             speckleWindowSize=50, speckleRange=32, preFilterCap=63)        # Calib_depth/depth4.py:156-168


def test_constant_shift_known_answer(synth):
    D, d0 = 32, 11
    L, R = synth.constant_shift_pair(256, 80, d0, seed=3)
    disp = so.compute(L, R, so.make_params(numDisparities=D, **C2_KW))
    assert disp.dtype == np.int16 and disp.shape == L.shape
    assert (disp[:, :D] == -16).all()                       # columns x < minX1 are invalid = (minD-1)*16
    inner = disp[6:-6, D + 6:-6]
    assert (np.abs(inner.astype(int) - 16 * d0) <= 1).all()
    assert (inner == 16 * d0).mean() > 0.95


@pytest.mark.parametrize("kw,D,shape,seed", [
    (C2_KW, 16, (24, 48), 0), (C2_KW, 32, (21, 70), 1), (D4_KW, 16, (30, 52), 2),
    (dict(C2_KW, blockSize=3, P1=72, P2=288), 16, (17, 40), 3),
    (dict(C2_KW, blockSize=7, P1=8 * 3 * 49, P2=32 * 3 * 49, uniquenessRatio=0), 16, (26, 44), 4),
    (dict(C2_KW, minDisparity=5, blockSize=3, P1=216, P2=288, uniquenessRatio=0), 16, (23, 60), 5),   # minD >= 2: disp2 marker quirk
])
def test_c_oracle_equals_numpy_restatement(kw, D, shape, seed, synth):
    H, W = shape
    L, R, _ = synth.stereo_pair(W, H, D, seed=seed)
    got, got_raw = so.compute(L, R, so.make_params(numDisparities=D, **kw), return_raw=True)
    want, want_raw = ref.compute(L, R, numDisparities=D, return_raw=True, **kw)
    np.testing.assert_array_equal(got_raw, want_raw)
    np.testing.assert_array_equal(got, want)


def test_random_noise_images_match(synth):
    rng = np.random.default_rng(9)
    L = rng.integers(0, 256, (20, 44), dtype=np.uint8)
    R = rng.integers(0, 256, (20, 44), dtype=np.uint8)
    got = so.compute(L, R, so.make_params(numDisparities=16, **C2_KW))
    want = ref.compute(L, R, numDisparities=16, **C2_KW)
    np.testing.assert_array_equal(got, want)


def test_negative_min_disparity_right_matcher_geometry(synth):
    """createRightMatcher semantics: minD = -(D-1), compute(right, left)."""
    D = 16
    L, R, _ = synth.stereo_pair(60, 22, D, seed=7)
    kw = dict(C2_KW, minDisparity=-(0 + D) + 1, uniquenessRatio=0, disp12MaxDiff=1000000)
    got = so.compute(R, L, so.make_params(numDisparities=D, **kw))
    want = ref.compute(R, L, numDisparities=D, **kw)
    np.testing.assert_array_equal(got, want)
    assert (got[:, -(D - 1):] == (kw["minDisparity"] - 1) * 16).all()


def test_threads_do_not_change_result(synth):
    L, R, _ = synth.stereo_pair(160, 120, 32, seed=11)
    p = so.make_params(numDisparities=32, **C2_KW)
    np.testing.assert_array_equal(so.compute(L, R, p, nthreads=1), so.compute(L, R, p, nthreads=4))


def test_two_plane_step_edge(synth):
    D = 32
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, (64, 300)).astype(np.uint8)
    from scipy.ndimage import uniform_filter
    tex = np.clip(uniform_filter(tex.astype(float), 2) * 1.0, 0, 255).astype(np.uint8)
    W = 200
    L = tex[:, :W].copy()
    R = np.empty_like(L)
    R[:32] = tex[:32, 5:5 + W]          # top half: disparity 5
    R[32:] = tex[32:, 12:12 + W]        # bottom half: disparity 12
    disp = so.compute(L, R, so.make_params(numDisparities=D, **C2_KW))
    assert (disp[4:26, D + 8:-16] == 5 * 16).mean() > 0.97
    assert (disp[38:60, D + 8:-16] == 12 * 16).mean() > 0.97


def test_filter_speckles_matches_union_find():
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 6, (40, 50)) * 40).astype(np.int16)
    img[rng.random(img.shape) < 0.2] = -16
    got = so.filter_speckles(img, -16, 6, 32)
    want = ref.speckles(img, -16, 6, 32)
    np.testing.assert_array_equal(got, want)
    assert (got == -16).sum() > (img == -16).sum()


def test_bad_params_rejected():
    L = np.zeros((8, 40), np.uint8)
    with pytest.raises(RuntimeError):
        so.compute(L, L, so.make_params(numDisparities=24, blockSize=5))


def test_oracle_empty_matching_range_is_an_all_invalid_map():
    L = np.random.default_rng(0).integers(0, 256, (9, 100), dtype=np.uint8)
    d, raw = so.compute(L, L, so.make_params(numDisparities=112, blockSize=5, P1=600, P2=2400, preFilterCap=63), return_raw=True)
    assert (d == -16).all() and (raw == -16).all()


def test_quirk_small_image_stripes_shifts_clamped_stripes():
    """QUIRK_SMALL_IMAGE_STRIPES ([recalled] computeDisparity3WAY: private stripe buffers + assembly from row overlap + i % stripe_sz).
    H = 12, blockSize 5: stripe_sz 3, overlap 4; stripe 1 (rows 3-5) would start at row -1, is clamped to 0 with dst_offset 0, and
    the assembly hands out its run's rows 4, 5 and a row that was never written.  Stripes 2 and 3 (start 2, 5) are regular.  With the
    quirk OFF stripe 1's run also starts at row 0 and writes its own rows, so its rows 4, 5 are exactly what the quirk shifts up."""
    rng = np.random.default_rng(4)
    L = rng.integers(0, 256, (12, 60), dtype=np.uint8)
    R = np.roll(L, -5, axis=1)
    prm = so.make_params(numDisparities=16, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=15, preFilterCap=63)
    try:
        so.set_quirk_small_image_stripes(False)
        _, off = so.compute(L, R, prm, return_raw=True)
        assert not so.undefined_rows(12, prm).any()
        so.set_quirk_small_image_stripes(True)
        d_on, on = so.compute(L, R, prm, return_raw=True)
        assert list(np.nonzero(so.undefined_rows(12, prm))[0]) == [5]
    finally:
        so.set_quirk_small_image_stripes(True)
    assert np.array_equal(on[[0, 1, 2, 6, 7, 8, 9, 10, 11]], off[[0, 1, 2, 6, 7, 8, 9, 10, 11]])
    assert np.array_equal(on[3], off[4]) and np.array_equal(on[4], off[5]) and (on[5] == -16).all()
    assert (off[3:6, 16:] != -16).any()                         # the rows that move do hold disparities
    # a regular image is untouched by the switch, and blockSize 11 reaches the quirk at H = 24 (stripe_sz 6 < overlap 7)
    prm11 = so.make_params(numDisparities=16, blockSize=11, P1=100, P2=1000, preFilterCap=31)
    assert list(np.nonzero(so.undefined_rows(24, prm11))[0]) == [11] and not so.undefined_rows(28, prm11).any()
    assert not so.undefined_rows(13, prm).any() and not so.undefined_rows(2448, prm).any()
