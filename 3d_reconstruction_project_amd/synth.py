"""Seeded synthetic inputs for the hot path.

The reference ships no stereo pair (only calibration .npz files) and its scanner needs a live RealSense,
so every benchmark / parity input is generated here: rectified stereo pairs with a known disparity field,
and surface-sampled cloud pairs with a known rigid offset.
"""
import numpy as np


def _gauss_blur(a, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(a, sigma, mode="reflect")


def stereo_truth(width, height, num_disparities):
    """Ground-truth disparity field: smooth sinusoid + a fronto-parallel foreground rectangle (float64 [H,W])."""
    D = num_disparities
    x = np.arange(width)[None, :]
    y = np.arange(height)[:, None]
    d = 0.5 * D + (40.0 / 128.0) * D * np.sin(2 * np.pi * x / width) * np.cos(2 * np.pi * y / height)
    y0, y1 = int(0.4 * height), int(0.6 * height)
    x0, x1 = int(0.4 * width), int(0.6 * width)
    d[y0:y1, x0:x1] = (110.0 / 128.0) * D
    return np.clip(d, 4.0 * D / 128.0, 123.0 * D / 128.0)


def stereo_pair(width=3264, height=2448, num_disparities=128, seed=20241008, noise=1.5):
    """Band-limited textured rectified pair.  Returns (left u8 [H,W], right u8 [H,W], truth f64 [H,W]).

    The left image is a window of a wider texture; the right image samples the same texture displaced by the
    ground-truth disparity (linear interpolation) plus N(0, noise) sensor noise.
    """
    rng = np.random.default_rng(seed)
    D = num_disparities
    Ww = width + D
    tex = np.zeros((height, Ww))
    for sigma in (1.0, 4.0, 16.0):
        f = _gauss_blur(rng.standard_normal((height, Ww)), sigma)
        tex += f / f.std()
    tex = 128.0 + 40.0 * tex
    left = np.clip(np.rint(tex[:, :width]), 0, 255).astype(np.uint8)     # left(x) = tex(x)
    d = stereo_truth(width, height, D)
    # right(x, y) shows the scene point that the left camera sees at x + d: sample the texture there
    # disparity convention: left(x) == right(x - d).  Rendered with the approximation right(xr) = left(xr + d(xr)),
    # adequate for a smooth field (the parity tests never compare against `d`, only GPU against oracle).
    xs = np.arange(width)[None, :] + d
    x0 = np.floor(xs).astype(np.int64)
    fr = xs - x0
    x0 = np.clip(x0, 0, Ww - 2)
    rows = np.arange(height)[:, None]
    right = (1.0 - fr) * tex[rows, x0] + fr * tex[rows, x0 + 1]
    right += rng.normal(0.0, noise, right.shape)
    right = np.clip(np.rint(right), 0, 255).astype(np.uint8)
    return left, right, d


def constant_shift_pair(width, height, shift, seed=1):
    """Known-answer pair: right = left shifted by a constant integer disparity on random texture."""
    rng = np.random.default_rng(seed)
    tex = np.zeros((height, width + shift))
    for sigma in (0.7, 2.0):
        f = _gauss_blur(rng.standard_normal(tex.shape), sigma)
        tex += f / f.std()
    tex = np.clip(np.rint(128 + 45 * tex), 0, 255).astype(np.uint8)
    left = tex[:, :width].copy()          # left(x)  = tex(x)
    right = tex[:, shift:].copy()         # right(x) = tex(x + shift) = left(x + shift)  => disparity = shift
    return left, right


def _sample_surface(n, rng):
    """Area-uniform rejection sampling of r(theta, phi) = 1 + 0.1 sin(4 theta) sin(3 phi)."""
    out = np.empty((0, 3))
    while out.shape[0] < n:
        m = int((n - out.shape[0]) * 1.6) + 1024
        v = rng.standard_normal((m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        theta = np.arccos(np.clip(v[:, 2], -1, 1))
        phi = np.arctan2(v[:, 1], v[:, 0])
        r = 1.0 + 0.1 * np.sin(4 * theta) * np.sin(3 * phi)
        keep = rng.random(m) < (r / 1.1) ** 2          # area element ~ r^2 (slope term neglected)
        out = np.concatenate([out, (v * r[:, None])[keep]], axis=0)
    return out[:n]


def rigid(axis, angle_deg, t):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    a = np.deg2rad(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


T_STAR = rigid((1, 1, 1), 1.5, (0.010, -0.005, 0.008))


def cloud_pair(n=1_000_000, seed=7, noise=0.001, scale=1.0):
    """Target and source samples of the same bumpy sphere; source = inv(T*) applied to an independent noisy
    sample, so that registration source->target should recover T* (SURVEY.md section 8d, config C3).
    `scale` shrinks the surface so that smaller n keeps the ~3.5 mm mean spacing of the 1M-point case."""
    rng = np.random.default_rng(seed)
    tgt = _sample_surface(n, rng) * scale
    src = _sample_surface(n, rng) * scale + rng.normal(0.0, noise, (n, 3))
    Ti = np.linalg.inv(T_STAR)
    src = src @ Ti[:3, :3].T + Ti[:3, 3]
    return src.astype(np.float32), tgt.astype(np.float32), T_STAR.copy()
