"""ctypes front-end of oracle/sgbm3way.c (the CPU restatement of OpenCV StereoSGBM MODE_SGBM_3WAY).

TEST INFRASTRUCTURE ONLY -- see the header of sgbm3way.c.  PARITY UNPINNED versus real OpenCV.
Reference call sites restated: Calib_depth/depth2.py:146-158 (StereoSGBM_create kwargs) and :251 (.compute).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libr3d_oracle.so")


class SgbmParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
        "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange")]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("sgbm3way.c", "graph.c", "normals.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        s16p = ctypes.POINTER(ctypes.c_int16)
        L.sgbm_oracle_compute.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(SgbmParams), s16p, s16p, ctypes.c_int]
        L.sgbm_oracle_compute.restype = ctypes.c_int
        L.sgbm_oracle_cost_rows.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(SgbmParams), ctypes.c_int, ctypes.c_int, ctypes.c_int, s16p]
        L.sgbm_oracle_cost_rows.restype = ctypes.c_int
        L.sgbm_oracle_filter_speckles.argtypes = [s16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int]
        L.sgbm_oracle_filter_speckles.restype = ctypes.c_int
        L.sgbm_oracle_set_quirk_small_image_stripes.argtypes = [ctypes.c_int]
        L.sgbm_oracle_set_quirk_small_image_stripes.restype = None
        L.sgbm_oracle_undefined_rows.argtypes = [ctypes.c_int, ctypes.POINTER(SgbmParams), u8p]
        L.sgbm_oracle_undefined_rows.restype = ctypes.c_int
        _lib = L
    return _lib


def set_quirk_small_image_stripes(on):
    """QUIRK_SMALL_IMAGE_STRIPES of sgbm3way.c (default on = [recalled] OpenCV: a stripe whose warm-up start is clamped to row 0
    hands out rows from further down the image; rows the original leaves uninitialised read INVALID here)."""
    lib().sgbm_oracle_set_quirk_small_image_stripes(int(bool(on)))


def undefined_rows(H, params):
    """bool [H]: rows whose content is undefined in the original (uninitialised stripe-buffer rows) under the quirk."""
    m = np.zeros(H, np.uint8)
    lib().sgbm_oracle_undefined_rows(int(H), ctypes.byref(params), m.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return m.astype(bool)


def make_params(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=None):
    """Same keyword names and defaults as cv2.StereoSGBM_create; `mode` is accepted and must be 3WAY (2) or None."""
    if mode not in (None, 2):
        raise ValueError("oracle restates MODE_SGBM_3WAY only")
    return SgbmParams(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                      uniquenessRatio, speckleWindowSize, speckleRange)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def compute(left, right, params, nthreads=1, return_raw=False):
    """left/right: uint8 [H,W].  Returns int16 [H,W] disparity x16 (invalid = (minD-1)*16)."""
    left, lp = _u8(left)
    right, rp = _u8(right)
    assert left.ndim == 2 and left.shape == right.shape
    H, W = left.shape
    disp = np.empty((H, W), np.int16)
    raw = np.empty((H, W), np.int16) if return_raw else None
    s16p = ctypes.POINTER(ctypes.c_int16)
    rc = lib().sgbm_oracle_compute(lp, rp, W, H, W, W, ctypes.byref(params), disp.ctypes.data_as(s16p),
                                   raw.ctypes.data_as(s16p) if return_raw else None, int(nthreads))
    if rc:
        raise RuntimeError(f"sgbm_oracle_compute failed rc={rc}")
    return (disp, raw) if return_raw else disp


def cost_rows(left, right, params, band_start, y0, y1):
    left, lp = _u8(left)
    right, rp = _u8(right)
    H, W = left.shape
    minD, D = params.minDisparity, params.numDisparities
    W1 = (W + min(minD, 0)) - max(minD + D, 0)
    out = np.empty((y1 - y0, W1, D), np.int16)
    rc = lib().sgbm_oracle_cost_rows(lp, rp, W, H, W, W, ctypes.byref(params), band_start, y0, y1,
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)))
    if rc:
        raise RuntimeError(f"sgbm_oracle_cost_rows failed rc={rc}")
    return out


def filter_speckles(img, new_val, max_speckle_size, max_diff):
    img = np.ascontiguousarray(img, dtype=np.int16).copy()
    H, W = img.shape
    rc = lib().sgbm_oracle_filter_speckles(img.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)), W, H, W,
                                           int(new_val), int(max_speckle_size), int(max_diff))
    if rc:
        raise RuntimeError(f"filter_speckles failed rc={rc}")
    return img
