"""orient_normals_consistent_tangent_plane(k) (normal_estimation.py:21) -- SURVEY.md section 8 row c2 / next-row f-3.

Open3D [recalled]: Riemannian graph = Euclidean MST edges + k-nearest-neighbour edges, edge weight 1 - |n_i . n_j|;
minimum spanning tree of that graph; propagation from the highest point (its normal is turned towards +z), flipping a
child's normal when it disagrees with its parent.  Here: the k-NN graph comes from the HIP kernel, the (inherently
sequential) Kruskal tree and breadth-first propagation run in C++ inside the library (r3d_orient_normals).
PARITY UNPINNED (the reference recorded no oriented normals); known difference from Open3D: the Euclidean-MST edges
come from the kNN graph itself (no Delaunay step), so disconnected kNN components are oriented independently.
"""
from . import cloud_ops


def orient_normals_consistent_tangent_plane(points, normals, k=100, ctx=None):
    return cloud_ops.orient_normals(points, normals, k, ctx=ctx)
