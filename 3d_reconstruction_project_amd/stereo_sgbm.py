"""Drop-in for the cv2.StereoSGBM object protocol the reference's depth viewers use.

Reference surface (Calib_depth/depth1.py:202-214,240-265,331; depth2.py:146-158,251):
    matcher = cv2.StereoSGBM_create(minDisparity=..., numDisparities=..., blockSize=..., P1=..., P2=...,
                                    disp12MaxDiff=..., uniquenessRatio=..., speckleWindowSize=..., speckleRange=...,
                                    preFilterCap=..., mode=cv2.STEREO_SGBM_MODE_SGBM_3WAY)
    disp = matcher.compute(gray_left, gray_right)        # int16 [H,W], disparity x16, invalid = (minD-1)*16
    matcher.setBlockSize(b); matcher.getNumDisparities(); ...
Same keyword names, defaults, getters/setters and error behaviour (an exception on bad input).  All arithmetic
runs in the HIP library (csrc/sgm.hip) through the C ABI in include/r3d.h.
"""
import ctypes
import os

import numpy as np

from . import _lib

STEREO_SGBM_MODE_SGBM = 0
STEREO_SGBM_MODE_HH = 1
STEREO_SGBM_MODE_SGBM_3WAY = 2
STEREO_SGBM_MODE_HH4 = 3

_FIELDS = ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
           "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")


class StereoSGBM:
    def __init__(self, minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                 uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=STEREO_SGBM_MODE_SGBM_3WAY, device=0):
        self._p = dict(minDisparity=minDisparity, numDisparities=numDisparities, blockSize=blockSize, P1=P1, P2=P2,
                       disp12MaxDiff=disp12MaxDiff, preFilterCap=preFilterCap, uniquenessRatio=uniquenessRatio,
                       speckleWindowSize=speckleWindowSize, speckleRange=speckleRange, mode=mode)
        self._device = _lib.parse_device(device)
        self._ctx = None

    # cv2-style accessors (depth1.py:240-265 tunes blockSize / numDisparities from key presses)
    def __getattr__(self, name):
        for pre in ("get", "set"):
            if name.startswith(pre):
                key = name[3:]
                key = key[0].lower() + key[1:]
                if key == "mode":
                    key = "mode"
                if key in self._p or key in ("p1", "p2"):
                    key = {"p1": "P1", "p2": "P2"}.get(key, key)
                    if pre == "get":
                        return lambda: self._p[key]
                    return lambda v: self._p.__setitem__(key, int(v))
        raise AttributeError(name)

    @property
    def context(self):
        if self._ctx is None:
            self._ctx = _lib.default_context(self._device)
        return self._ctx

    def params_struct(self):
        return _lib.SgbmParams(*[int(self._p[k]) for k in _FIELDS])

    def compute(self, left, right):
        """left, right: uint8 [H,W] rectified grayscale.  Returns int16 [H,W]."""
        left = np.asarray(left)
        right = np.asarray(right)
        if left.dtype != np.uint8 or right.dtype != np.uint8 or left.ndim != 2 or left.shape != right.shape:
            raise ValueError("StereoSGBM.compute expects two uint8 single-channel images of equal size "
                             "(the reference converts with cv2.cvtColor(..., COLOR_BGR2GRAY) first)")
        left = np.ascontiguousarray(left)
        right = np.ascontiguousarray(right)
        H, W = left.shape
        self._last_shape = (H, W)
        disp = np.empty((H, W), np.int16)
        p = self.params_struct()
        vp = ctypes.c_void_p
        self.context.call("r3d_sgbm_compute", ctypes.byref(p), left.ctypes.data_as(vp), right.ctypes.data_as(vp),
                          W, H, W, disp.ctypes.data_as(vp))
        return disp

    def compute_device(self, d_left, d_right, width, height, stride, d_disp):
        """Device-pointer variant (ints): enqueues on the context stream, no synchronisation."""
        p = self.params_struct()
        vp = ctypes.c_void_p
        self.context.call("r3d_sgbm_compute_dev", ctypes.byref(p), vp(d_left), vp(d_right), int(width), int(height),
                          int(stride), vp(d_disp))

    def compute_batch_device(self, d_lefts, d_rights, width, height, stride, d_disps, done_events=None):
        """Device-pointer batch (lists of ints): maps are pipelined over the library's internal lanes.  done_events (list of
        Context.event() handles or None entries): map i's completion is recorded into done_events[i], so a consumer on another
        context can wait for it (Context.wait_event) while later maps are still running."""
        n = len(d_lefts)
        assert len(d_rights) == n and len(d_disps) == n and (done_events is None or len(done_events) == n)
        if n > 1:   # three lanes + the context stream (+ the consumers' streams): more than HIP's default four hardware queues
            _lib.warn_if_few_hw_queues(int(os.environ.get("R3D_SGM_LANES", "3")) + 2)
        arr = ctypes.c_void_p * n
        p = self.params_struct()
        if done_events is None:
            self.context.call("r3d_sgbm_compute_batch_dev", ctypes.byref(p), n, arr(*d_lefts), arr(*d_rights), int(width),
                              int(height), int(stride), arr(*d_disps))
        else:
            self.context.call("r3d_sgbm_compute_batch_events_dev", ctypes.byref(p), n, arr(*d_lefts), arr(*d_rights), int(width),
                              int(height), int(stride), arr(*d_disps), arr(*done_events))

    def compute_batch(self, lefts, rights):
        """lefts / rights: sequences of uint8 [H,W] images of equal size -> list of int16 disparity maps."""
        ctx = self.context
        H, W = np.asarray(lefts[0]).shape
        dl = [ctx.to_device(np.ascontiguousarray(a, dtype=np.uint8)) for a in lefts]
        dr = [ctx.to_device(np.ascontiguousarray(a, dtype=np.uint8)) for a in rights]
        dd = [ctx.alloc(W * H * 2) for _ in lefts]
        try:
            self.compute_batch_device(dl, dr, W, H, W, dd)
            ctx.sync()
            out = []
            for d in dd:
                a = np.empty((H, W), np.int16)
                ctx.d2h(a, d)
                out.append(a)
        finally:
            for q in dl + dr + dd:
                ctx.free(q)
        self._last_shape = (H, W)
        return out

    def debug_fetch(self, want_cost=False, want_hsum=False, want_raw=True):
        """Stage outputs of the last compute (parity tests): dict of int16 arrays."""
        ctx = self.context
        D = self._p["numDisparities"]
        minD = self._p["minDisparity"]
        H, W = self._last_shape
        W1 = (W + min(minD, 0)) - max(minD + D, 0)
        # slots per cost-volume column: the smallest of 32 / 64 / 128 / 256 that holds D (128 / 256 for the v1 / v3 kernels)
        import os
        legacy = os.environ.get("R3D_SGM_IMPL") in ("v1", "v3")
        dp = next(c for c in ((128, 256) if legacy else (32, 64, 128, 256)) if D <= c)
        out = {}
        cost = np.empty((H, W1, dp), np.int16) if want_cost else None
        hsum = np.empty((H, W1, dp), np.int16) if want_hsum else None
        raw = np.empty((H, W), np.int16) if want_raw else None
        vp = ctypes.c_void_p
        ctx.call("r3d_sgbm_debug_fetch", cost.ctypes.data_as(vp) if want_cost else None,
                 hsum.ctypes.data_as(vp) if want_hsum else None, raw.ctypes.data_as(vp) if want_raw else None)
        if want_cost:
            out["cost"] = cost[:, :, :D]
        if want_hsum:
            out["hsum"] = hsum[:, :, :D]
        if want_raw:
            out["raw"] = raw
        return out


def filterSpeckles(img, newVal, maxSpeckleSize, maxDiff, device=0):
    """cv2.filterSpeckles on an int16 image; returns the filtered copy."""
    a = np.ascontiguousarray(img, dtype=np.int16).copy()
    H, W = a.shape
    _lib.default_context(_lib.parse_device(device)).call("r3d_filter_speckles", a.ctypes.data_as(ctypes.c_void_p), W, H,
                                                         int(newVal), int(maxSpeckleSize), int(maxDiff))
    return a


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                      uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=STEREO_SGBM_MODE_SGBM, device=0):
    """Factory with cv2.StereoSGBM_create's keyword names and defaults.  Only MODE_SGBM_3WAY is implemented (the
    mode every reference call site passes); other modes raise at compute()."""
    return StereoSGBM(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
                      speckleWindowSize, speckleRange, mode, device)


def createRightMatcher(matcher_left):
    """cv2.ximgproc.createRightMatcher(matcher_left) (Calib_depth/depth1.py:215, depth2.py:161): a StereoSGBM with
    minDisparity = -(minD + D) + 1, the same D / blockSize / P1 / P2 / preFilterCap / mode, uniquenessRatio 0,
    disp12MaxDiff 1000000 and no speckle filter; it is called as compute(right, left) (depth2.py:252)."""
    p = matcher_left._p
    return StereoSGBM(minDisparity=-(p["minDisparity"] + p["numDisparities"]) + 1, numDisparities=p["numDisparities"],
                      blockSize=p["blockSize"], P1=p["P1"], P2=p["P2"], disp12MaxDiff=1000000, preFilterCap=p["preFilterCap"],
                      uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=p["mode"], device=matcher_left._device)


def reference_matcher(numDisparities=128, blockSize=5, family="depth2", device=0):
    """The two parameter families the reference uses (Calib_depth/depth2.py:139-158, depth4.py:147-168)."""
    kw = dict(minDisparity=0, numDisparities=numDisparities, blockSize=blockSize, P1=8 * 3 * blockSize ** 2,
              P2=32 * 3 * blockSize ** 2, disp12MaxDiff=1, preFilterCap=63, mode=STEREO_SGBM_MODE_SGBM_3WAY)
    if family in ("depth1", "depth2", "depth3"):
        kw.update(uniquenessRatio=15, speckleWindowSize=0, speckleRange=2)
    elif family in ("depth4", "depth_test"):
        kw.update(uniquenessRatio=10, speckleWindowSize=50, speckleRange=32)
    else:
        raise ValueError(family)
    return StereoSGBM_create(device=device, **kw)


def depth(left, right, **params):
    """north_star alias: one-shot disparity map with reference (depth2.py) defaults overridable by kwargs."""
    fam = params.pop("family", "depth2")
    m = reference_matcher(params.pop("numDisparities", 128), params.pop("blockSize", 5), fam, params.pop("device", 0))
    for k, v in params.items():
        m._p[k] = int(v)
    return m.compute(left, right)
