"""CPU oracle for the point-cloud half of the hot path -- TEST INFRASTRUCTURE ONLY (numpy / scipy, float64).

Restates the Open3D (legacy, float64) functions the reference calls.  Open3D is a third-party dependency that is
NOT vendored in /root/reference, is un-pinned there (no requirements file; most likely 0.18) and is not installed
in this image, so each function follows the library's published algorithm and is anchored on the reference's
call sites:
    voxel_down_sample          pointcloud_alignment.py:22-23, test/check84.py:180
    estimate_normals(Hybrid)   pointcloud_alignment.py:27-28, test/GICP1.py:77, test/check84.py:181-182
    remove_statistical_outlier pointcloud_processing.py:35, test/check_lama1.py:175
    remove_radius_outlier      pointcloud_processing.py:39
    create_from_rgbd_image     test/check84.py:155-159,172-178
    registration_icp (P2P)     pointcloud_alignment.py:35-39
    registration_generalized_icp  test/GICP1.py:99-102
    point-to-plane             test/check2.py:151-154
PINNED by the reference's recorded runs (tests/golden, from test/output84 and test/output): back-projection, voxel
down-sampling, statistical outlier removal and hybrid-search PCA normals reproduce the recorded PLY files
(tests/test_cloud_oracle.py).  ICP / GICP / point-to-plane are PARITY UNPINNED (the reference recorded no
transforms); they are checked by analytic known-answer tests only.
"""
import json
import struct

import numpy as np
from scipy.spatial import cKDTree


# ---------------------------------------------------------------------------------------------- file formats
def read_ply(path):
    """Binary little-endian PLY as Open3D writes it (double x,y,z[,nx,ny,nz], uchar r,g,b)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        props, n = [], 0
        while True:
            line = f.readline().strip().split()
            if line[0] == b"end_header":
                break
            if line[0] == b"format":
                assert line[1] == b"binary_little_endian"
            elif line[0] == b"element":
                if line[1] == b"vertex":
                    n = int(line[2])
                    in_vertex = True
                else:
                    in_vertex = False
            elif line[0] == b"property" and in_vertex:
                props.append((line[2].decode(), {b"double": "<f8", b"float": "<f4", b"uchar": "u1"}[line[1]]))
        data = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
    out = {"points": np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float64)}
    if "nx" in data.dtype.names:
        out["normals"] = np.stack([data["nx"], data["ny"], data["nz"]], 1).astype(np.float64)
    if "red" in data.dtype.names:
        out["colors"] = np.stack([data["red"], data["green"], data["blue"]], 1)
    return out


def read_png16(path):
    from PIL import Image
    return np.asarray(Image.open(path)).astype(np.uint16)


def read_intrinsics(path):
    with open(path) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------ create_from_rgbd_image
DEPTH_SCALE_F32 = np.float32(1.0) / np.float32(0.001)      # check84.py:158: 1.0/self.depth_scale = 999.99994


def backproject(depth_u16, intr, depth_scale=DEPTH_SCALE_F32, depth_trunc=3.0, flip=True):
    """Appendix C of SURVEY.md (fixture-verified): z = f32(raw)/f32(scale), z > trunc or z == 0 dropped,
    x=(u-ppx)*z/fx, y=(v-ppy)*z/fy in float64, then (x,-y,-z); row-major pixel order.  Returns (points, (v,u))."""
    z = (depth_u16.astype(np.float32) / np.float32(depth_scale)).astype(np.float32)
    z[z > np.float32(depth_trunc)] = 0
    v, u = np.nonzero(z > 0)
    zz = z[v, u].astype(np.float64)
    x = (u - intr["ppx"]) * zz / intr["fx"]
    y = (v - intr["ppy"]) * zz / intr["fy"]
    pts = np.stack([x, y, zz], 1)
    if flip:
        pts = pts * np.array([1.0, -1.0, -1.0])
    return pts, (v, u)


# ---------------------------------------------------------------------------------------- voxel_down_sample
def voxel_keys(points, voxel):
    origin = points.min(axis=0) - 0.5 * voxel
    return np.floor((points - origin) / voxel).astype(np.int64)


def voxel_down_sample(points, voxel, colors=None, normals=None):
    """Legacy PointCloud::VoxelDownSample: mean of points / colors / normals per occupied voxel.  Output order of
    the original is hash-map order (unspecified); here voxels are emitted in lexicographic key order."""
    keys = voxel_keys(points, voxel)
    _, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)

    def mean(a):
        out = np.zeros((cnt.shape[0], a.shape[1]))
        np.add.at(out, inv, a)
        return out / cnt[:, None]

    res = [mean(points)]
    if colors is not None:
        res.append(mean(colors))
    if normals is not None:
        res.append(mean(normals))
    return res[0] if len(res) == 1 else tuple(res)


# ------------------------------------------------------------------------------------------ outlier removal
def statistical_outlier_mask(points, nb_neighbors, std_ratio):
    """keep_i <=> mean of the k smallest distances (self included) < mean + ratio * std(ddof=1)."""
    d, _ = cKDTree(points).query(points, k=nb_neighbors)
    a = d.mean(axis=1)
    return a < a.mean() + std_ratio * a.std(ddof=1)


def radius_outlier_mask(points, nb_points, radius):
    """keep_i <=> #neighbours within radius (self included) > nb_points   [recalled Open3D semantics]."""
    cnt = cKDTree(points).query_ball_point(points, radius, return_length=True)
    return cnt > nb_points


# --------------------------------------------------------------------------------------------------- normals
def _canon_sign(v):
    """Sign convention shared with the HIP kernel (an eigenvector's sign is solver-dependent; the reference's own
    sign is Open3D's solver's, unknowable here): the largest-magnitude component is positive."""
    a = np.abs(v)
    lead = v[0] if (a[0] >= a[1] and a[0] >= a[2]) else (v[1] if a[1] >= a[2] else v[2])
    return -v if lead < 0 else v


def _d2(points, idx, q):
    d = points[idx] - q
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _pca_normals(points, idx_lists):
    n = points.shape[0]
    normals = np.zeros((n, 3))
    covs = np.zeros((n, 3, 3))
    for i in range(n):
        idx = idx_lists[i]
        if len(idx) < 3:
            normals[i] = (0.0, 0.0, 1.0)
            continue
        q = points[idx]
        c = np.cov(q.T, bias=True)
        covs[i] = c
        w, v = np.linalg.eigh(c)
        normals[i] = _canon_sign(v[:, 0])
    return normals, covs


def _nearest_total_order(points, q, k, extra=12):
    """k nearest of every query under the TOTAL ORDER (squared distance, index): exact-distance ties are common on
    voxelised clouds and a kd-tree breaks them arbitrarily.  Returns (idx [m,k], d2 [m,k])."""
    kk = min(k + extra, points.shape[0])
    _, idx = cKDTree(points).query(q, k=kk)
    if kk == 1:
        idx = idx[:, None]
    d2 = _d2(points, idx, q[:, None, :])
    order = np.lexsort((idx, d2), axis=1)
    idx = np.take_along_axis(idx, order, 1)[:, :k]
    d2 = np.take_along_axis(d2, order, 1)[:, :k]
    return idx, d2


def hybrid_neighbors(points, radius, max_nn, queries=None):
    """<= max_nn nearest neighbours with distance < radius (self included), nearest first."""
    q = points if queries is None else queries
    idx, d2 = _nearest_total_order(points, q, min(max_nn, points.shape[0]))
    keep = d2 < radius * radius
    return [idx[i][keep[i]] for i in range(q.shape[0])]


def estimate_normals_hybrid(points, radius, max_nn):
    """Legacy estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)): population covariance of the neighbours,
    eigenvector of the smallest eigenvalue, (0,0,1) when fewer than 3 neighbours; sign is arbitrary."""
    return _pca_normals(points, hybrid_neighbors(points, radius, max_nn))[0]


def estimate_normals_knn(points, k):
    idx, _ = _nearest_total_order(points, points, min(k, points.shape[0]))
    return _pca_normals(points, list(idx))[0]


# ----------------------------------------------------------------------------------------------- registration
def transform_points(T, p):
    return p @ T[:3, :3].T + T[:3, 3]


def euler_zyx_to_matrix(x6):
    """TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x3..5 (Euler composition, not so(3) exp)."""
    a, b, c = x6[0], x6[1], x6[2]
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = x6[3:]
    return T


def solve_6x6(JTJ, JTr):
    """SolveJacobianSystemAndObtainExtrinsicMatrix: x = LDLT(JTJ) \\ (-JTr); identity when |det| < 1e-6."""
    det = np.linalg.det(JTJ)
    if not np.isfinite(det) or abs(det) < 1e-6:
        return np.eye(4), False
    return euler_zyx_to_matrix(np.linalg.solve(JTJ, -JTr)), True


def umeyama(src, dst):
    """Eigen::umeyama(src, dst, with_scaling=false)."""
    ms, md = src.mean(0), dst.mean(0)
    sigma = (dst - md).T @ (src - ms) / src.shape[0]
    U, d, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = md - R @ ms
    return T


def covariances_from_normals(normals, eps=1e-3):
    """GeneralizedICP InitializePointCloudForGeneralizedICP: C = R diag(eps,1,1) R^T with R = GetRotationFromE1ToX(n)
    = I - (1-eps) n n^T for unit n.  QUIRK: when n.e1 < -0.99 the original returns R = I, i.e. uses e1 as normal."""
    n = normals.copy()
    n[n[:, 0] < -0.99] = (1.0, 0.0, 0.0)
    return np.eye(3)[None] - (1.0 - eps) * n[:, :, None] * n[:, None, :]


def _evaluate(src, tree, max_dist):
    """GetRegistrationResultAndCorrespondences: 1-NN with dist < max_dist; fitness, inlier RMSE (Euclidean)."""
    tgt = tree.data
    _, cand = tree.query(src, k=min(4, tgt.shape[0]))
    if cand.ndim == 1:
        cand = cand[:, None]
    d2 = _d2(tgt, cand, src[:, None, :])
    order = np.lexsort((cand, d2), axis=1)[:, 0]                 # total order (d2, index), as the HIP kernel
    j = np.take_along_axis(cand, order[:, None], 1)[:, 0]
    d2 = np.take_along_axis(d2, order[:, None], 1)[:, 0]
    ok = d2 < max_dist * max_dist
    i = np.nonzero(ok)[0]
    if i.size == 0:
        return i, j[ok], 0.0, 0.0
    return i, j[ok], i.size / src.shape[0], float(np.sqrt(d2[ok].sum() / i.size))


def _inv_sqrt_sym(M):
    w, v = np.linalg.eigh(M)
    return (v * (1.0 / np.sqrt(w))[:, None, :]) @ np.swapaxes(v, 1, 2)


def registration(source, target, max_dist, init=None, mode="p2p", max_iteration=30, relative_fitness=1e-6,
                 relative_rmse=1e-6, target_normals=None, source_cov=None, target_cov=None, history=None):
    """registration_icp / registration_generalized_icp loop (Open3D Registration.cpp RegistrationICP).
    Returns dict(T, fitness, inlier_rmse, iterations, correspondences)."""
    T = np.eye(4) if init is None else np.array(init, float)
    P = transform_points(T, np.asarray(source, float))
    tgt = np.asarray(target, float)
    Cs = None if source_cov is None else T[:3, :3] @ source_cov @ T[:3, :3].T
    tree = cKDTree(tgt)
    i, j, fit, rmse = _evaluate(P, tree, max_dist)
    it = 0
    for it in range(1, max_iteration + 1):
        if i.size == 0:
            U = np.eye(4)
        elif mode == "p2p":
            U = umeyama(P[i], tgt[j])
        else:
            s, t = P[i], tgt[j]
            if mode == "p2plane":
                n = target_normals[j]
                r = ((s - t) * n).sum(1)
                J = np.concatenate([np.cross(s, n), n], 1)                      # [n,6]
                JTJ, JTr = J.T @ J, J.T @ r
            elif mode == "gicp":
                W = _inv_sqrt_sym(target_cov[j] + Cs[i])                        # [n,3,3]
                d = s - t
                r = np.einsum("nij,nj->ni", W, d)
                sk = np.zeros((s.shape[0], 3, 3))
                sk[:, 0, 1], sk[:, 0, 2] = -s[:, 2], s[:, 1]
                sk[:, 1, 0], sk[:, 1, 2] = s[:, 2], -s[:, 0]
                sk[:, 2, 0], sk[:, 2, 1] = -s[:, 1], s[:, 0]
                Jb = np.concatenate([-sk, np.broadcast_to(np.eye(3), sk.shape)], 2)   # [n,3,6]
                J = np.einsum("nij,njk->nik", W, Jb)
                JTJ = np.einsum("nri,nrj->ij", J, J)
                JTr = np.einsum("nri,nr->i", J, r)
            else:
                raise ValueError(mode)
            U, _ = solve_6x6(JTJ, JTr)
        T = U @ T
        P = transform_points(U, P)
        if Cs is not None:
            Cs = U[:3, :3] @ Cs @ U[:3, :3].T
        pf, pr = fit, rmse
        i, j, fit, rmse = _evaluate(P, tree, max_dist)
        if history is not None:
            history.append((fit, rmse))
        if abs(pf - fit) < relative_fitness and abs(pr - rmse) < relative_rmse:
            break
    return dict(T=T, fitness=fit, inlier_rmse=rmse, iterations=it, correspondences=int(i.size))
