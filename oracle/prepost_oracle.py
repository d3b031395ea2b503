"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the per-frame stages either side of the SGM matcher in
Calib_depth/depth*.py.  Imported by tests/, __graft_entry__.smoke() and nothing in the product path.

  init_undistort_rectify_map  cv2.initUndistortRectifyMap(K, dist, R, P, size, CV_16SC2)      depth2.py:125-128
  remap_fixed                 cv2.remap(frame, map1, map2, INTER_LINEAR)                      depth2.py:243-244
  bgr2gray                    cv2.cvtColor(img, COLOR_BGR2GRAY)                               depth2.py:247-248
  wls_filter                  createDisparityWLSFilter(matcher).filter(dl, view, None, dr)    depth2.py:164-166,255
  normalize_minmax            cv2.normalize(x, None, 0, 255, NORM_MINMAX)                     depth2.py:256

PARITY UNPINNED: OpenCV / opencv_contrib are a dependency of the reference that is absent from /root/reference and
from this image (cv2 is not importable), and the reference holds no recorded output of these calls.  Everything here
is a restatement from the published algorithms of OpenCV 4.x (imgproc: undistort, remap, color; ximgproc:
disparity_filters.cpp, fgs_filter.cpp), [recalled] where a constant or an operation order could not be checked.
remap / cvtColor / the maps are integer or fixed-point pipelines whose arithmetic is fully specified; the WLS filter
is float32 and follows the structure of DisparityWLSFilterImpl (confidence from box variance + LR consistency, fast
global smoother on disparity*confidence and on confidence, ratio).
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS

f32 = np.float32


def _inv3_cofactor(S):
    """cv::invert for a 3x3 double matrix with DECOMP_LU takes the closed form: cofactors times 1/det."""
    S = np.asarray(S, np.float64)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    d = 1.0 / d
    t = np.empty((3, 3))
    t[0, 0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[0, 1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[0, 2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[1, 0] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[1, 1] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[1, 2] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[2, 0] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[2, 1] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[2, 2] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def rectify_inverse(new_camera, R):
    """iR = (P[:, :3] @ R)^-1, the 3x3 matrix initUndistortRectifyMap walks the destination grid with."""
    A = np.asarray(new_camera, np.float64)[:3, :3]
    Rm = np.eye(3) if R is None else np.asarray(R, np.float64)
    M = np.empty((3, 3))
    for i in range(3):                       # plain triple loop: the product order of a 3x3 gemm, k ascending
        for j in range(3):
            M[i, j] = A[i, 0] * Rm[0, j] + A[i, 1] * Rm[1, j] + A[i, 2] * Rm[2, j]
    return _inv3_cofactor(M)


def dist14(dist):
    d = np.zeros(14)
    if dist is not None:
        v = np.asarray(dist, np.float64).ravel()
        assert v.size in (4, 5, 8, 12, 14)
        d[:v.size] = v
    return d


def init_undistort_rectify_map(camera, dist, R, new_camera, size):
    """-> (map1 int16 [H,W,2], map2 uint16 [H,W]).  Scalar loop of initUndistortRectifyMapComputer: per row
    _x,_y,_w start at i*ir[1]+ir[2] ... and are advanced by += ir[0], ir[3], ir[6] per column (running sums, fp64);
    rational + tangential + thin-prism distortion; tilt coefficients must be zero.  u,v scaled by 32 and rounded half
    to even (saturate_cast<int>), integer part -> map1, 5+5 fraction bits -> map2."""
    Wd, Hd = size
    K = np.asarray(camera, np.float64)
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4, tx, ty = dist14(dist)
    assert tx == 0 and ty == 0, "tilted sensor model not restated"
    ir = rectify_inverse(new_camera, R).ravel()
    u0, v0, fx, fy = K[0, 2], K[1, 2], K[0, 0], K[1, 1]
    i = np.arange(Hd, dtype=np.float64)

    def running(start, step):
        a = np.empty((Hd, Wd))
        a[:, 0] = start
        a[:, 1:] = step
        return np.add.accumulate(a, axis=1)      # sequential fp64 additions, like the C loop

    _x = running(i * ir[1] + ir[2], ir[0])
    _y = running(i * ir[4] + ir[5], ir[3])
    _w = running(i * ir[7] + ir[8], ir[6])
    w = 1.0 / _w
    x = _x * w
    y = _y * w
    x2 = x * x
    y2 = y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2)
    yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2)
    u = fx * 1.0 * xd + u0                        # invProj = 1 (identity tilt)
    v = fy * 1.0 * yd + v0
    lim = 2.0 ** 31
    iu = np.clip(np.rint(u * INTER_TAB_SIZE), -lim, lim - 1).astype(np.int64)
    iv = np.clip(np.rint(v * INTER_TAB_SIZE), -lim, lim - 1).astype(np.int64)
    map1 = np.empty((Hd, Wd, 2), np.int16)
    map1[..., 0] = (iu >> INTER_BITS).astype(np.int16)            # (short) cast wraps
    map1[..., 1] = (iv >> INTER_BITS).astype(np.int16)
    map2 = ((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1))).astype(np.uint16)
    return map1, map2


def bilinear_tab_i():
    """BilinearTab_i [1024][4] of initInterTab2D(INTER_LINEAR, fixpt): short(w*32768) of the four products; the one entry
    whose sum is not 32768 (fx=fy=0: 32768 saturates to 32767) gets the difference added to its last tap."""
    t = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), np.int32)
    for fy in range(INTER_TAB_SIZE):
        for fx in range(INTER_TAB_SIZE):
            wy = (f32(1) - f32(fy) / f32(INTER_TAB_SIZE), f32(fy) / f32(INTER_TAB_SIZE))
            wx = (f32(1) - f32(fx) / f32(INTER_TAB_SIZE), f32(fx) / f32(INTER_TAB_SIZE))
            e = [min(int(np.rint(f32(wy[a] * wx[b]) * f32(INTER_REMAP_COEF_SCALE))), 32767) for a in (0, 1) for b in (0, 1)]
            s = sum(e)
            if s != INTER_REMAP_COEF_SCALE:
                e[3] -= s - INTER_REMAP_COEF_SCALE
            t[fy * INTER_TAB_SIZE + fx] = e
    return t


_TAB = None


def remap_fixed(src, map1, map2, border_value=0):
    """remap(src, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT) for uint8 images with 1, 3 or 4
    channels: out = (sum_k tap_k * w_k + 2^14) >> 15 with the fixed-point table weights; taps outside the source take
    border_value; a destination pixel whose 2x2 footprint lies wholly outside gets border_value."""
    global _TAB
    if _TAB is None:
        _TAB = bilinear_tab_i()
    src = np.asarray(src)
    assert src.dtype == np.uint8
    s3 = src[:, :, None] if src.ndim == 2 else src
    Hs, Ws, cn = s3.shape
    sx = map1[..., 0].astype(np.int64)
    sy = map1[..., 1].astype(np.int64)
    wt = _TAB[(map2.astype(np.int64) & (INTER_TAB_SIZE * INTER_TAB_SIZE - 1))]            # [H,W,4]
    acc = np.zeros(map2.shape + (cn,), np.int64)
    k = 0
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = sx + dx, sy + dy
            inside = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
            tap = np.where(inside[..., None], s3[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)].astype(np.int64),
                           border_value)
            acc += tap * wt[..., k][..., None]
            k += 1
    out = ((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS)
    gone = (sx >= Ws) | (sx + 1 < 0) | (sy >= Hs) | (sy + 1 < 0)
    out = np.where(gone[..., None], border_value, out)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if src.ndim == 2 else out


def bgr2gray(img):
    """cvtColor(BGR2GRAY) for uint8: (B*1868 + G*9617 + R*4899 + 2^13) >> 14."""
    a = np.asarray(img).astype(np.int64)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ WLS filter

def fgs_lut(sigma_color, channels=1):
    """weights_LUT[i] = -exp(-sqrt(i)/sigma) for a squared colour distance i (fgs_filter.cpp); float32 of the fp64 value."""
    i = np.arange(channels * 255 * 255 + 1, dtype=np.float64)
    return (-np.exp(-np.sqrt(i) / float(sigma_color))).astype(np.float32)


def fgs_weights(guide, lut):
    g = np.asarray(guide)
    g3 = (g[:, :, None] if g.ndim == 2 else g).astype(np.int64)
    H, W = g3.shape[:2]
    ch = np.zeros((H, W), np.float32)
    cv = np.zeros((H, W), np.float32)
    ch[:, :-1] = lut[((g3[:, :-1] - g3[:, 1:]) ** 2).sum(-1)]
    cv[:-1, :] = lut[((g3[:-1] - g3[1:]) ** 2).sum(-1)]
    return ch, cv


def _fgs_pass(cur, C, lam):
    """One tridiagonal solve along axis 1 for every row (Thomas algorithm, float32, this operation order):
         a_j = lam*C[j-1], c_j = lam*C[j]   (C <= 0 holds -weight, C[-1] = 0)
         denom = (1 - a_j - c_j) - a_j*cc[j-1];  cc[j] = c_j/denom;  f[j] = (f[j] - a_j*f[j-1])/denom
         back substitution f[j] -= cc[j]*f[j+1]."""
    H, W = C.shape
    lam = f32(lam)
    one = f32(1)
    cc = np.empty((H, W), np.float32)
    c0 = lam * C[:, 0]
    denom = one - c0
    cc[:, 0] = c0 / denom
    cur[:, 0] = cur[:, 0] / denom
    for j in range(1, W):
        a = lam * C[:, j - 1]
        c = lam * C[:, j]
        denom = ((one - a) - c) - a * cc[:, j - 1]
        cc[:, j] = c / denom
        cur[:, j] = (cur[:, j] - a * cur[:, j - 1]) / denom
    for j in range(W - 2, -1, -1):
        cur[:, j] = cur[:, j] - cc[:, j] * cur[:, j + 1]
    return cur


def fgs_filter(src, ch, cv, lam, lambda_attenuation=0.25, num_iter=3):
    """FastGlobalSmootherFilter::filter for one float32 plane: num_iter x (horizontal solve, vertical solve), lambda
    multiplied by lambda_attenuation after each iteration."""
    cur = np.array(src, np.float32, copy=True)
    lam = f32(lam)
    cvt = np.ascontiguousarray(cv.T)
    for _ in range(num_iter):
        cur = _fgs_pass(cur, ch, lam)
        cur = np.ascontiguousarray(_fgs_pass(np.ascontiguousarray(cur.T), cvt, lam).T)
        lam = f32(lam * f32(lambda_attenuation))
    return cur


def _box_mean_reflect101(a_int, r):
    """boxFilter(normalize=true, BORDER_REFLECT_101) of an integer-valued float plane: sums are exact in the fp64
    accumulators OpenCV uses for 32F sources, result = float32(sum * (1/k^2))."""
    k = 2 * r + 1
    p = np.pad(a_int.astype(np.int64), r, mode="reflect")
    c = np.zeros((p.shape[0] + 1, p.shape[1] + 1), np.int64)
    c[1:, 1:] = p.cumsum(0).cumsum(1)
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return (s.astype(np.float64) * (1.0 / (k * k))).astype(np.float32)


def depth_discontinuity(disp_roi, radius, roll_off=0.001):
    d = disp_roi.astype(np.int64)
    mean = _box_mean_reflect101(d, radius)
    sqmean = _box_mean_reflect101(d * d, radius)
    var = sqmean - mean * mean
    v = f32(1) - f32(roll_off) * var
    return np.where(v > 0, v, f32(0)).astype(np.float32)


def wls_rois(W, H, min_disp, num_disp):
    """createDisparityWLSFilter for an SGBM matcher: left_offset = max(0, minD+D), right_offset = max(0, -minD), no
    top/bottom margin.  -> (x, y, w, h) of the left-view ROI and of the right-view ROI."""
    lo = max(0, min_disp + num_disp)
    ro = max(0, -min_disp)
    w = W - lo - ro
    left = (lo, 0, w, H)
    right = (W - (lo + w), 0, w, H)
    return left, right


def wls_confidence(disp_left, disp_right, min_disp, num_disp, radius, lrc_thresh=24):
    """computeConfidenceMap: depth-discontinuity maps of both views inside their ROIs, then the LR-consistency test;
    x255.  Pixels whose right-view partner falls outside the right ROI keep their discontinuity value [recalled]."""
    H, W = disp_left.shape
    (lx, ly, lw, lh), (rx, ry, rw, rh) = wls_rois(W, H, min_disp, num_disp)
    ddl = np.zeros((H, W), np.float32)
    ddr = np.zeros((H, W), np.float32)
    if lw > 0:
        ddl[:, lx:lx + lw] = depth_discontinuity(disp_left[:, lx:lx + lw], radius)
        ddr[:, rx:rx + rw] = depth_discontinuity(disp_right[:, rx:rx + rw], radius)
    conf = ddl.copy()
    j = np.arange(W)[None, :].repeat(H, 0)
    dl = disp_left.astype(np.int64)
    ridx = j - (dl >> 4)
    in_left = (j >= lx) & (j < lx + lw)
    in_right = (ridx >= rx) & (ridx < rx + rw)
    rsafe = np.clip(ridx, 0, W - 1)
    rows = np.arange(H)[:, None].repeat(W, 1)
    dr = disp_right.astype(np.int64)[rows, rsafe]
    agree = np.abs(dl + dr) < lrc_thresh
    both = in_left & in_right
    conf = np.where(both & agree, np.minimum(ddl, ddr[rows, rsafe]), conf)
    conf = np.where(both & ~agree, f32(0), conf)
    return (f32(255) * conf).astype(np.float32)


def wls_filter(disp_left, guide, disp_right, min_disp, num_disp, block_size, lam=8000.0, sigma_color=1.5,
               lrc_thresh=24, return_confidence=False):
    """DisparityWLSFilter::filter(disp_left, left_view, None, disp_right) of a filter made by
    createDisparityWLSFilter(sgbm_left): int16 in, int16 out; outside the ROI the output is 16*(minD-1)."""
    dl = np.asarray(disp_left, np.int16)
    dr = np.asarray(disp_right, np.int16)
    H, W = dl.shape
    radius = int(np.ceil(0.5 * block_size))
    (lx, ly, lw, lh), _ = wls_rois(W, H, min_disp, num_disp)
    out = np.full((H, W), 16 * (min_disp - 1), np.int16)
    conf = wls_confidence(dl, dr, min_disp, num_disp, radius, lrc_thresh)
    if lw > 0:
        g = np.asarray(guide)[:, lx:lx + lw]
        ch, cv = fgs_weights(g, fgs_lut(sigma_color, 1 if g.ndim == 2 else g.shape[2]))
        c = conf[:, lx:lx + lw]
        dm = c * dl[:, lx:lx + lw].astype(np.float32)
        dmf = fgs_filter(dm, ch, cv, lam)
        cf = fgs_filter(c, ch, cv, lam)
        q = dmf * (f32(1) / (cf + f32(0.00001)))
        out[:, lx:lx + lw] = np.clip(np.rint(q), -32768, 32767).astype(np.int16)
    return (out, conf) if return_confidence else out


def normalize_minmax(src, alpha=0.0, beta=255.0):
    """cv2.normalize(src, None, alpha, beta, NORM_MINMAX) keeping the source type: scale = (hi-lo)/(max-min) (0 when
    the image is constant), shift = lo - min*scale in fp64; convertTo applies them in float32 and rounds half to even."""
    a = np.asarray(src)
    lo, hi = min(alpha, beta), max(alpha, beta)
    smin, smax = float(a.min()), float(a.max())
    scale = (hi - lo) * (1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0)
    shift = lo - smin * scale
    v = a.astype(np.float32) * f32(scale) + f32(shift)
    if np.issubdtype(a.dtype, np.integer):
        info = np.iinfo(a.dtype)
        return np.clip(np.rint(v), info.min, info.max).astype(a.dtype)
    return v.astype(a.dtype)
