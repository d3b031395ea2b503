"""cv2-named wrappers for the per-frame stages either side of the matcher in the reference's depth viewers
(Calib_depth/depth2.py:102-129, 160-166, 243-257; same calls in depth1/3/4.py):

    map_x, map_y = cv2.initUndistortRectifyMap(mtx, dist, R, P, image_size, cv2.CV_16SC2)
    rectified    = cv2.remap(frame, map_x, map_y, cv2.INTER_LINEAR)
    gray         = cv2.cvtColor(rectified, cv2.COLOR_BGR2GRAY)
    wls_filter   = cv2.ximgproc.createDisparityWLSFilter(matcher_left=stereo_matcher)
    wls_filter.setLambda(8000); wls_filter.setSigmaColor(1.5)
    filtered     = wls_filter.filter(disparity_left, gray_left, None, disparity_right)
    filtered     = cv2.normalize(filtered, None, 0, 255, cv2.NORM_MINMAX)

All arithmetic runs in the HIP library (csrc/prepost.hip) through the C ABI in include/r3d.h; there is no CPU path.
Parity of this group is unpinned (OpenCV is absent here; see oracle/prepost_oracle.py)."""
import ctypes
import math

import numpy as np

from . import _lib

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32

CV_16SC2 = 11            # cv2.CV_16SC2
INTER_LINEAR = 1         # cv2.INTER_LINEAR
COLOR_BGR2GRAY = 6       # cv2.COLOR_BGR2GRAY
NORM_MINMAX = 32         # cv2.NORM_MINMAX
BORDER_CONSTANT = 0
SOLVER_PARTITIONED, SOLVER_SEQUENTIAL = 0, 1     # r3d_wls_params.solver


class WlsParams(ctypes.Structure):
    _fields_ = [("lambda_", ctypes.c_double), ("sigma_color", ctypes.c_double), ("lambda_attenuation", ctypes.c_double),
                ("discontinuity_roll_off", ctypes.c_double), ("min_disparity", _i32), ("num_disparities", _i32),
                ("discontinuity_radius", _i32), ("lrc_thresh", _i32), ("num_iter", _i32), ("solver", _i32)]


_lib.register({
    "r3d_init_undistort_rectify_map": ([_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp], ctypes.c_int),
    "r3d_remap_u8": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp], ctypes.c_int),
    "r3d_remap_u8_dev": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp], ctypes.c_int),
    "r3d_bgr2gray": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp], ctypes.c_int),
    "r3d_bgr2gray_dev": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp], ctypes.c_int),
    "r3d_wls_filter": ([_vp, ctypes.POINTER(WlsParams), _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp], ctypes.c_int),
    "r3d_wls_filter_dev": ([_vp, ctypes.POINTER(WlsParams), _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp], ctypes.c_int),
    "r3d_normalize_minmax_s16": ([_vp, _vp, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _vp], ctypes.c_int),
    "r3d_normalize_minmax_s16_dev": ([_vp, _vp, ctypes.c_int64, ctypes.c_double, ctypes.c_double, _vp], ctypes.c_int),
})


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type=CV_16SC2, ctx=None):
    """-> (map1 int16 [H,W,2], map2 uint16 [H,W]); size = (width, height) as in cv2.  Only the fixed-point map type the
    reference asks for (CV_16SC2) exists."""
    if m1type != CV_16SC2:
        raise ValueError("initUndistortRectifyMap: only m1type=CV_16SC2 is implemented (the type the reference uses)")
    ctx = ctx or _lib.default_context()
    K = np.ascontiguousarray(cameraMatrix, np.float64).reshape(3, 3)
    d = None if distCoeffs is None else np.ascontiguousarray(distCoeffs, np.float64).ravel()
    Rm = None if R is None else np.ascontiguousarray(R, np.float64).reshape(3, 3)
    P = np.ascontiguousarray(newCameraMatrix, np.float64)
    if P.shape not in ((3, 3), (3, 4)):
        raise ValueError("newCameraMatrix must be 3x3 or 3x4")
    w, h = int(size[0]), int(size[1])
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    ctx.call("r3d_init_undistort_rectify_map", _p(K), _p(d), 0 if d is None else d.size, _p(Rm), _p(P), P.shape[1], w, h,
             _p(m1), _p(m2))
    return m1, m2


def _check_maps(map1, map2):
    m1 = np.ascontiguousarray(map1)
    m2 = np.ascontiguousarray(map2)
    if m1.dtype != np.int16 or m1.ndim != 3 or m1.shape[2] != 2 or m2.dtype != np.uint16 or m2.shape != m1.shape[:2]:
        raise ValueError("remap: maps must be the CV_16SC2 + CV_16UC1 pair of initUndistortRectifyMap")
    return m1, m2


def remap(src, map1, map2, interpolation=INTER_LINEAR, borderValue=0, with_gray=False, ctx=None):
    """cv2.remap(src, map1, map2, INTER_LINEAR) for uint8 images with 1/3/4 channels and fixed-point maps.
    with_gray=True additionally returns cvtColor(result, COLOR_BGR2GRAY), produced by the same kernel."""
    if interpolation != INTER_LINEAR:
        raise ValueError("remap: only INTER_LINEAR is implemented (the mode the reference uses)")
    ctx = ctx or _lib.default_context()
    s = np.ascontiguousarray(src)
    if s.dtype != np.uint8 or s.ndim not in (2, 3):
        raise ValueError("remap: uint8 image expected")
    cn = 1 if s.ndim == 2 else s.shape[2]
    m1, m2 = _check_maps(map1, map2)
    dh, dw = m2.shape
    dst = np.empty((dh, dw) if s.ndim == 2 else (dh, dw, cn), np.uint8)
    gray = np.empty((dh, dw), np.uint8) if with_gray else None
    ctx.call("r3d_remap_u8", _p(s), s.shape[1], s.shape[0], s.strides[0], cn, _p(m1), _p(m2), dw, dh, int(borderValue),
             _p(dst), _p(gray))
    return (dst, gray) if with_gray else dst


def cvtColor(src, code=COLOR_BGR2GRAY, ctx=None):
    if code != COLOR_BGR2GRAY:
        raise ValueError("cvtColor: only COLOR_BGR2GRAY is implemented (the conversion the reference uses)")
    ctx = ctx or _lib.default_context()
    s = np.ascontiguousarray(src)
    if s.dtype != np.uint8 or s.ndim != 3 or s.shape[2] not in (3, 4):
        raise ValueError("cvtColor: uint8 BGR / BGRA image expected")
    out = np.empty(s.shape[:2], np.uint8)
    ctx.call("r3d_bgr2gray", _p(s), s.shape[1], s.shape[0], s.strides[0], s.shape[2], _p(out))
    return out


def normalize(src, dst=None, alpha=0, beta=255, norm_type=NORM_MINMAX, ctx=None):
    """cv2.normalize(src, None, 0, 255, cv2.NORM_MINMAX) for the int16 maps the WLS filter returns."""
    if norm_type != NORM_MINMAX:
        raise ValueError("normalize: only NORM_MINMAX is implemented")
    ctx = ctx or _lib.default_context()
    s = np.ascontiguousarray(src)
    if s.dtype != np.int16:
        raise ValueError("normalize: int16 input expected (a disparity map)")
    out = np.empty_like(s)
    if s.size:
        ctx.call("r3d_normalize_minmax_s16", _p(s), s.size, float(alpha), float(beta), _p(out))
    return out


class DisparityWLSFilter:
    """cv2.ximgproc.DisparityWLSFilter as created from an SGBM left matcher (confidence-aware mode)."""

    def __init__(self, min_disparity, num_disparities, block_size, device=0):
        self._min_disp = int(min_disparity)
        self._num_disp = int(num_disparities)
        self._lambda = 8000.0
        self._sigma = 1.0
        self._lrc = 24
        self._radius = int(math.ceil(0.5 * block_size))
        self._roll_off = 0.001
        self._device = device
        self.solver = SOLVER_PARTITIONED
        self._conf = None
        self._roi = None
        self._ctx = None

    # accessor protocol of the original
    def setLambda(self, v): self._lambda = float(v)
    def getLambda(self): return self._lambda
    def setSigmaColor(self, v): self._sigma = float(v)
    def getSigmaColor(self): return self._sigma
    def setLRCthresh(self, v): self._lrc = int(v)
    def getLRCthresh(self): return self._lrc
    def setDepthDiscontinuityRadius(self, v): self._radius = int(v)
    def getDepthDiscontinuityRadius(self): return self._radius
    def getConfidenceMap(self): return self._conf

    def getROI(self):
        return self._roi

    @property
    def context(self):
        if self._ctx is None:
            self._ctx = _lib.default_context(_lib.parse_device(self._device))
        return self._ctx

    def params_struct(self):
        return WlsParams(self._lambda, self._sigma, 0.25, self._roll_off, self._min_disp, self._num_disp, self._radius,
                         self._lrc, 3, int(self.solver))

    def filter(self, disparity_map_left, left_view, filtered_disparity_map=None, disparity_map_right=None):
        if disparity_map_right is None:
            raise ValueError("DisparityWLSFilter.filter: this filter was created from a matcher (confidence mode) and "
                             "needs disparity_map_right, as the original does")
        dl = np.ascontiguousarray(disparity_map_left)
        dr = np.ascontiguousarray(disparity_map_right)
        g = np.ascontiguousarray(left_view)
        if dl.dtype != np.int16 or dr.dtype != np.int16 or dl.ndim != 2 or dr.shape != dl.shape:
            raise ValueError("DisparityWLSFilter.filter: int16 disparity maps of equal size expected")
        if g.dtype != np.uint8 or g.shape[:2] != dl.shape or (g.ndim == 3 and g.shape[2] != 3) or g.ndim not in (2, 3):
            raise ValueError("DisparityWLSFilter.filter: uint8 guide (1 or 3 channels) of the disparity map's size expected")
        h, w = dl.shape
        out = np.empty((h, w), np.int16)
        conf = np.empty((h, w), np.float32)
        p = self.params_struct()
        self.context.call("r3d_wls_filter", ctypes.byref(p), _p(dl), _p(dr), _p(g), 1 if g.ndim == 2 else 3, g.strides[0], w, h,
                          _p(out), _p(conf))
        self._conf = conf
        lo, ro = max(0, self._min_disp + self._num_disp), max(0, -self._min_disp)
        self._roi = (lo, 0, w - lo - ro, h)
        return out

    def filter_device(self, d_disp_left, d_disp_right, d_guide, guide_cn, guide_stride, width, height, d_out, d_conf=None):
        """Device-pointer variant (ints): enqueued on the context stream, no host synchronisation."""
        p = self.params_struct()
        self.context.call("r3d_wls_filter_dev", ctypes.byref(p), _vp(d_disp_left), _vp(d_disp_right), _vp(d_guide), int(guide_cn),
                          int(guide_stride), int(width), int(height), _vp(d_out), _vp(d_conf) if d_conf else None)


def createDisparityWLSFilter(matcher_left):
    """cv2.ximgproc.createDisparityWLSFilter(matcher_left) (Calib_depth/depth2.py:164).  Like the original it
    re-configures the matcher it is given: disp12MaxDiff = 1000000, speckleWindowSize = 0, uniquenessRatio = 0 (the
    filter wants dense, unpruned disparities and does its own consistency check), and derives the valid ROI and the
    discontinuity radius ceil(0.5*blockSize) from it."""
    matcher_left.setDisp12MaxDiff(1000000)
    matcher_left.setSpeckleWindowSize(0)
    matcher_left.setUniquenessRatio(0)
    return DisparityWLSFilter(matcher_left.getMinDisparity(), matcher_left.getNumDisparities(), matcher_left.getBlockSize(),
                              device=getattr(matcher_left, "_device", 0))
