"""orient_normals_consistent_tangent_plane(k) (normal_estimation.py:21) -- SURVEY.md section 8 row c2 / next-row f-3.

Open3D [recalled] (the tensor method converts to a legacy cloud and calls the legacy one, so this is a host algorithm in
the reference as well): Delaunay tetrahedralisation of the cloud by Qhull -> Euclidean MST of its edges; Riemannian graph =
EMST edges + k-nearest-neighbour edges that are not Delaunay edges, weight 1 - |n_i . n_j|; minimum spanning tree of that
graph; propagation from the highest point (its normal is turned towards +z), flipping a child that disagrees with its parent.

Split here: the tetrahedralisation comes from Qhull on the host exactly as in Open3D (scipy.spatial.Delaunay wraps the
same library, options "Qbb Qt"); the k-NN graph comes from the HIP kernel; both spanning trees and the (inherently
sequential) propagation run in C++ inside the library (r3d_orient_normals_graph).  PARITY UNPINNED (the reference recorded
no oriented normals); checked sign for sign against oracle/cloud_oracle.orient_normals in tests/test_cloud_gpu.py.
delaunay=False selects the k-NN-graph-only variant (no Qhull step; a deviation from Open3D, see include/r3d.h).
"""
import numpy as np

from . import cloud_ops


def delaunay_edges(points, qhull_options="Qbb Qt"):
    """Unique undirected edges [m,2] (int32) of the Delaunay tetrahedralisation (TetraMesh::CreateFromPointCloud)."""
    from scipy.spatial import Delaunay            # Qhull, as in Open3D; an ImportError here is loud on purpose
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(p) < 4:
        raise ValueError("Not enough points to create a tetrahedral mesh.")
    tet = Delaunay(p, qhull_options=qhull_options).simplices
    e = np.concatenate([tet[:, [a, b]] for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))], 0)
    e.sort(axis=1)
    code = np.unique(e[:, 0].astype(np.int64) * len(p) + e[:, 1])
    return np.stack([code // len(p), code % len(p)], 1).astype(np.int32)


def orient_normals_consistent_tangent_plane(points, normals, k=100, delaunay=True, ctx=None):
    edges = delaunay_edges(points) if delaunay else None
    return cloud_ops.orient_normals(points, normals, k, delaunay_edges=edges, ctx=ctx)
