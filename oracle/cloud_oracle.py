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
    orient_normals_consistent_tangent_plane   normal_estimation.py:21
    tensor voxel_down_sample   pointcloud_processing.py:27, test/GICP1.py:71-72
    scanning loops             main.py:34-54, test/GICP1.py:134-155 (fuse_loop)
PINNED by the reference's recorded runs (tests/golden, from test/output84 and test/output): back-projection, voxel
down-sampling, statistical outlier removal and hybrid-search PCA normals reproduce the recorded PLY files
(tests/test_cloud_oracle.py).  ICP / GICP / point-to-plane are PARITY UNPINNED (the reference recorded no
transforms); they are checked by analytic known-answer tests only.  Normal orientation and the tensor (float32,
origin 0) voxel grid are PARITY UNPINNED as well (nothing recorded by the reference; Open3D semantics [recalled]).
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
    """Appendix C of SURVEY.md (fixture-verified): z = f32(raw)/f32(scale), z >= trunc or z == 0 dropped,
    x=(u-ppx)*z/fx, y=(v-ppy)*z/fy in float64, then (x,-y,-z); row-major pixel order.  Returns (points, (v,u)).
    [recalled] Image::ConvertDepthToFloatImage clips `*p >= depth_trunc` with the float promoted to the double parameter; the
    recorded frames agree with both >= and > (their scale puts raw 3000 at 3.0000002)."""
    z = (depth_u16.astype(np.float32) / np.float32(depth_scale)).astype(np.float32)
    z[z.astype(np.float64) >= float(depth_trunc)] = 0
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
    """RemoveStatisticalOutliers [recalled]: a_i = mean of the k smallest distances (self included); the cloud mean sums
    only a_i > 0 but divides by the number of points, the deviation sums only a_i > 0 and divides by n - 1 (Bessel);
    keep_i <=> 0 < a_i < mean + ratio * std.  (a_i == 0 needs >= k coincident points; without them this is the plain
    mean / std(ddof=1) rule the recorded PLY files verify.)"""
    n = points.shape[0]
    if n == 0:
        return np.zeros(0, bool)
    d, _ = cKDTree(points).query(points, k=min(nb_neighbors, n))
    d = d.reshape(n, -1)
    a = d.mean(axis=1)
    pos = a > 0
    mean = a[pos].sum() / n
    std = np.sqrt(((a[pos] - mean) ** 2).sum() / (n - 1)) if n > 1 else 0.0
    return pos & (a < mean + std_ratio * std)


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


def _normals_lib():
    import ctypes
    from . import sgbm_oracle
    L = ctypes.CDLL(sgbm_oracle.build())
    dp, i64p, i32p = (ctypes.POINTER(t) for t in (ctypes.c_double, ctypes.c_int64, ctypes.c_int32))
    L.r3d_oracle_cumulant_cov.argtypes = [dp, i64p, i32p, ctypes.c_int64, ctypes.c_int32, i32p, dp]
    L.r3d_oracle_cumulant_cov.restype = None
    L.r3d_oracle_fast_eigen3x3.argtypes = [dp, ctypes.c_int64, i32p, dp]
    L.r3d_oracle_fast_eigen3x3.restype = None
    return L, dp, i64p, i32p


def cumulant_covariances(points, idx_lists, cfg=None):
    """utility::ComputeCovariance of every neighbour list (oracle/normals.c: one pass of nine cumulants over the raw
    coordinates, nearest neighbour first); identity where a list has fewer than 3 entries (EstimatePerPointCovariances)."""
    L, dp, i64p, i32p = _normals_lib()
    n = len(idx_lists)
    cnt = np.fromiter((len(ix) for ix in idx_lists), np.int32, n)
    k = max(int(cnt.max()) if n else 0, 1)
    pad = np.zeros((n, k), np.int64)
    for i, ix in enumerate(idx_lists):
        pad[i, :len(ix)] = ix
    pts = np.ascontiguousarray(points, np.float64)
    cov = np.empty((n, 3, 3))
    c = None if cfg is None else np.ascontiguousarray(cfg, np.int32)
    L.r3d_oracle_cumulant_cov(pts.ctypes.data_as(dp), pad.ctypes.data_as(i64p), cnt.ctypes.data_as(i32p), n, k,
                              None if c is None else c.ctypes.data_as(i32p), cov.ctypes.data_as(dp))
    return cov, cnt


def fast_eigen3x3(covs, cfg=None):
    """FastEigen3x3 of every covariance (oracle/normals.c): eigenvector of the smallest eigenvalue with the sign the closed
    form produces; the zero vector for an all-zero matrix."""
    L, dp, _, i32p = _normals_lib()
    covs = np.ascontiguousarray(covs, np.float64).reshape(-1, 9)
    out = np.empty((len(covs), 3))
    c = None if cfg is None else np.ascontiguousarray(cfg, np.int32)
    L.r3d_oracle_fast_eigen3x3(covs.ctypes.data_as(dp), len(covs), None if c is None else c.ctypes.data_as(i32p), out.ctypes.data_as(dp))
    return out


def _pca_normals(points, idx_lists, prev_normals=None, centers=None, cfg=None):
    """Legacy PointCloud::EstimateNormals on given neighbour lists (nearest first): cumulant covariance -> FastEigen3x3
    (oracle/normals.c, PINNED by the recorded frames sign included); fewer than 3 neighbours (identity covariance) or an all-zero
    covariance -> (0,0,1) (the previous normal if the cloud has normals); a cloud that already carries normals keeps their
    orientation (each new normal is turned towards the old one).  `centers` is accepted for the callers that restrict the
    queries to a subset; the covariance itself only depends on the neighbour lists.  Returns (normals, covariances)."""
    covs, cnt = cumulant_covariances(points, idx_lists, cfg)
    normals = fast_eigen3x3(covs, cfg)
    zero = (normals == 0).all(1)
    if prev_normals is not None:
        prev = np.asarray(prev_normals, float)
        normals[zero] = prev[zero]
        flip = (normals * prev).sum(1) < 0
        normals[flip] = -normals[flip]
    else:
        normals[zero] = (0.0, 0.0, 1.0)
    return normals, covs


def _nearest_total_order(points, q, k, extra=12):
    """k nearest of every query under the TOTAL ORDER (squared distance, index): exact-distance ties are common on
    voxelised clouds and a kd-tree breaks them arbitrarily.  Returns (idx [m,k], d2 [m,k])."""
    kk = min(k + extra, points.shape[0])
    _, idx = cKDTree(points).query(q, k=kk, workers=-1)
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


def estimate_normals_hybrid(points, radius, max_nn, prev_normals=None):
    """Legacy estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)): population covariance of the neighbours,
    eigenvector of the smallest eigenvalue, (0,0,1) when fewer than 3 neighbours; sign is arbitrary unless the cloud
    already carries normals (then each new normal is turned towards the old one, test/GICP1.py:148 relies on it)."""
    return _pca_normals(points, hybrid_neighbors(points, radius, max_nn), prev_normals)[0]


def estimate_normals_knn(points, k, queries=None):
    """kNN normals; `queries` (indices into points) restricts the output to a subset (spot checks on large clouds)."""
    if queries is None:
        idx, _ = _nearest_total_order(points, points, min(k, points.shape[0]))
        return _pca_normals(points, list(idx))[0]
    queries = np.asarray(queries, np.int64)
    idx, _ = _nearest_total_order(points, points[queries], min(k, points.shape[0]))
    return _pca_normals(points, list(idx), centers=points[queries])[0]


# ------------------------------------------------------------------------------- tensor (o3d.t) voxel grid
def voxel_down_sample_tensor(points, voxel, colors=None, normals=None):
    """o3d.t.geometry.PointCloud.voxel_down_sample(voxel) on a Float32 cloud (pointcloud_processing.py:26-27,
    test/GICP1.py:71-72) [recalled, Open3D 0.18 reduction="mean"]: key = floor(p / voxel) in float32 (grid origin 0, NOT
    the legacy min_bound - voxel/2), attributes reduced as float32 sums / float32 count.  The original's summation order is
    that of atomic adds on the device (unspecified); here members are added in their original order, in float32.
    Voxels come out in lexicographic key order (the original's is hash-map order).  Inputs are rounded to float32 first
    (from_legacy(..., Float32)); returns float64 arrays holding float32 values (to_legacy)."""
    p32 = np.asarray(points, np.float64).astype(np.float32)
    keys = np.floor(p32 / np.float32(voxel)).astype(np.int64)
    _, inv, cnt = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(inv, kind="stable")
    starts = np.concatenate([[0], np.cumsum(cnt)])

    def mean32(a):
        a32 = np.asarray(a, np.float64).astype(np.float32)[order]
        out = np.zeros((cnt.shape[0], 3), np.float32)
        pos = np.zeros(cnt.shape[0], np.int64)
        live = np.arange(cnt.shape[0])
        while live.size:                                   # member j of every voxel that still has one: sequential fp32 adds
            out[live] = out[live] + a32[starts[live] + pos[live]]
            pos[live] += 1
            live = live[pos[live] < cnt[live]]
        return (out / cnt.astype(np.float32)[:, None]).astype(np.float64)

    res = [mean32(p32)]
    if colors is not None:
        res.append(mean32(colors))
    if normals is not None:
        res.append(mean32(normals))
    return res[0] if len(res) == 1 else tuple(res)


# ------------------------------------------------------------------------------------- normal orientation
def delaunay_edges(points, qhull_options="Qbb Qt"):
    """Unique undirected edges (a < b) of the Delaunay tetrahedralisation, lexicographically sorted.  Open3D builds it with
    Qhull ("d Qbb Qt", TetraMesh::CreateFromPointCloud [recalled]); scipy wraps the same library."""
    from scipy.spatial import Delaunay
    tet = Delaunay(np.asarray(points, np.float64), qhull_options=qhull_options).simplices.astype(np.int64)
    pairs = np.concatenate([tet[:, [a, b]] for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))], 0)
    pairs.sort(axis=1)
    return np.unique(pairs, axis=0)


def kruskal(n, v0, v1, w):
    """Minimum spanning forest, edges visited in the total order (weight, v0, v1) (the original's std::sort leaves equal
    weights in unspecified order).  Returns the boolean mask of kept edges.  The union-find loop is oracle/graph.c."""
    import ctypes
    from . import sgbm_oracle
    L = ctypes.CDLL(sgbm_oracle.build())
    i64p = ctypes.POINTER(ctypes.c_int64)
    L.r3d_oracle_kruskal.argtypes = [ctypes.c_int64, ctypes.c_int64, i64p, i64p, i64p, ctypes.POINTER(ctypes.c_uint8)]
    L.r3d_oracle_kruskal.restype = ctypes.c_int64
    v0 = np.ascontiguousarray(v0, np.int64)
    v1 = np.ascontiguousarray(v1, np.int64)
    order = np.ascontiguousarray(np.lexsort((v1, v0, np.asarray(w, np.float64))), np.int64)
    kept = np.zeros(len(v0), np.uint8)
    rc = L.r3d_oracle_kruskal(int(n), len(v0), order.ctypes.data_as(i64p), v0.ctypes.data_as(i64p), v1.ctypes.data_as(i64p),
                              kept.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    if rc < 0:
        raise RuntimeError("kruskal oracle failed")
    return kept.astype(bool)


def orient_normals(points, normals, k, edges=None, delaunay_blocks_knn=True):
    """PointCloud::OrientNormalsConsistentTangentPlane(k) (legacy; the tensor method of normal_estimation.py:21 converts
    to legacy and calls it) [recalled, Open3D 0.17 / 0.18 with lambda = 0, cos_alpha_tol = 1]:
      1. Delaunay tetrahedralisation; Kruskal on its edges weighted by squared length -> Euclidean MST.
      2. Riemannian graph = EMST edges re-weighted 1 - |n_a . n_b|, plus for every point its k nearest neighbours
         (SearchKNN: the point itself is one of the k) as edges with the same weight -- QUIRK: an edge is skipped when it
         is already in `graph_edges`, and that set holds EVERY Delaunay edge, not only the EMST ones
         (delaunay_blocks_knn=False adds them, i.e. what the comment in the original says it does).
      3. Kruskal on that graph; breadth-first propagation over the tree from the point of maximum z (first one on ties),
         whose normal is turned towards +z; a child is flipped when n_parent . n_child < 0.
    Sign propagation over a tree does not depend on the visiting order, so the result is defined by the two spanning trees.
    `edges` lets a test pass a precomputed Delaunay edge list (the product receives the same list)."""
    p = np.asarray(points, np.float64)
    nrm = np.array(normals, np.float64)
    n = p.shape[0]
    if n < 4:
        raise ValueError("Not enough points to create a tetrahedral mesh.")          # TetraMesh::CreateFromPointCloud
    de = delaunay_edges(p) if edges is None else np.asarray(edges, np.int64).reshape(-1, 2)
    d = p[de[:, 0]] - p[de[:, 1]]
    keep = kruskal(n, de[:, 0], de[:, 1], (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    emst = de[keep]

    def weight(a, b):
        return 1.0 - np.abs((nrm[a, 0] * nrm[b, 0] + nrm[a, 1] * nrm[b, 1]) + nrm[a, 2] * nrm[b, 2])

    idx, _ = _nearest_total_order(p, p, min(k, n))
    a = np.repeat(np.arange(n, dtype=np.int64), idx.shape[1])
    b = idx.reshape(-1).astype(np.int64)
    sel = a != b
    kn = np.stack([np.minimum(a[sel], b[sel]), np.maximum(a[sel], b[sel])], 1)
    kn = np.unique(kn, axis=0)
    blocked = de if delaunay_blocks_knn else emst
    code = lambda e: e[:, 0] * n + e[:, 1]                                           # noqa: E731  EdgeIndex
    kn = kn[~np.isin(code(kn), code(blocked))]
    g = np.concatenate([emst, kn], 0)
    tree = g[kruskal(n, g[:, 0], g[:, 1], weight(g[:, 0], g[:, 1]))]
    # adjacency (CSR) and breadth-first propagation
    both = np.concatenate([tree, tree[:, ::-1]], 0)
    both = both[np.argsort(both[:, 0], kind="stable")]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(both[:, 0], minlength=n))])
    root = int(np.argmax(p[:, 2]))
    if nrm[root, 2] < 0:
        nrm[root] = -nrm[root]
    seen = np.zeros(n, bool)
    seen[root] = True
    frontier = np.array([root], np.int64)
    while frontier.size:
        reps = ptr[frontier + 1] - ptr[frontier]
        par = np.repeat(frontier, reps)
        slot = np.repeat(ptr[frontier], reps) + (np.arange(int(reps.sum())) - np.repeat(np.cumsum(reps) - reps, reps))
        child = both[slot, 1]
        new = ~seen[child]
        par, child = par[new], child[new]
        flip = ((nrm[par] * nrm[child]).sum(1)) < 0
        nrm[child[flip]] = -nrm[child[flip]]
        seen[child] = True
        frontier = child
    return nrm


# ----------------------------------------------------------------------------------------------- registration
def transform_points(T, p, rotate_only=False):
    """PointCloud::Transform (pointcloud_alignment.py:42).  Written out per coordinate as ((r0 x + r1 y) + r2 z) + t, without
    fused multiply-adds: the order Eigen uses for a 4x4 times a point is an implementation detail of the original; fixing one
    here (the HIP kernel's) keeps the loops that feed a transformed cloud back into a neighbour search comparable bit for bit
    instead of up to rounding (an ill-conditioned PCA normal amplifies a 1e-16 coordinate difference to 1e-2)."""
    T = np.asarray(T, float)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    out = np.empty_like(p, dtype=float)
    for i in range(3):
        out[:, i] = ((T[i, 0] * x + T[i, 1] * y) + T[i, 2] * z) + (0.0 if rotate_only else T[i, 3])
    return out


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
    _, cand = tree.query(src, k=min(4, tgt.shape[0]), workers=-1)
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


# -------------------------------------------------------------------------------------------- scanning loops
def fuse_loop(frames, flavour="icp", threshold=0.02, voxel_size=0.01, max_iter=100, log=None, hook=None):
    """The reference's scanning loops on a list of frames (None / empty = failed capture, skipped: main.py:39,53-54).
    flavour "icp"  -- main.py:34-54 with pointcloud_alignment.py:6-43: the first valid frame becomes the model; every later
        frame: voxel_down_sample(voxel_size) of frame AND model, registration_icp(PointToPoint, threshold, identity,
        criteria(1e-6, 1e-6, max_iter)), the DOWN-SAMPLED transformed frame is appended to the model.  frames: [n,3] arrays.
    flavour "gicp" -- test/GICP1.py:134-155 with :81-104: frames are (points, normals); registration_generalized_icp
        (threshold, identity, default criteria = 30 iterations) of the frame against the whole model, points and normals
        transformed and appended, then estimate_normals(Hybrid(0.05, 30)) on the whole model (:148), which keeps the
        orientation of the normals already there.
    Returns (model points, model normals or None); `log` (list) receives one registration result per aligned frame.
    `hook(k, T)` (sensitivity experiments only, tools/cpu_gicp_sensitivity.py) may return a replacement for the k-th
    registration's transform before it is applied."""
    model = model_n = None
    k = 0
    for f in frames:
        if f is None:
            continue
        fp, fn = (f, None) if flavour == "icp" else f
        if len(fp) == 0:
            continue
        if model is None:
            model, model_n = np.array(fp, float), (None if fn is None else np.array(fn, float))
            continue
        if flavour == "icp":
            s = voxel_down_sample(fp, voxel_size)
            t = voxel_down_sample(model, voxel_size)
            res = registration(s, t, threshold, mode="p2p", max_iteration=max_iter)
            model = np.concatenate([model, transform_points(res["T"], s)], 0)
        else:
            res = registration(fp, model, threshold, mode="gicp", max_iteration=30, target_normals=model_n,
                               target_cov=covariances_from_normals(model_n), source_cov=covariances_from_normals(fn))
            T = res["T"]
            if hook is not None:
                T2 = hook(k, T)
                T = T if T2 is None else T2
            model = np.concatenate([model, transform_points(T, fp)], 0)
            model_n = np.concatenate([model_n, transform_points(T, fn, rotate_only=True)], 0)
            model_n = estimate_normals_hybrid(model, 0.05, 30, prev_normals=model_n)
        k += 1
        if log is not None:
            log.append(res)
    return model, model_n
