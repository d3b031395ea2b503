"""orient_normals_consistent_tangent_plane(k) (normal_estimation.py:21) -- SURVEY.md section 8 row c2 / next-row f-3.

Open3D [recalled]: Riemannian graph = Euclidean MST edges + k-nearest-neighbour edges, edge weight 1 - |n_i . n_j|;
minimum spanning tree of that graph; depth-first propagation from the highest point (its normal is turned towards
+z), flipping a child's normal when it disagrees with its parent.  The spanning-tree work is inherently sequential and
stays on the host; the k-NN graph (the only O(N k) geometry work) comes from the HIP library (r3d_knn_graph).
PARITY UNPINNED (the reference recorded no oriented normals); differences from Open3D: the Euclidean-MST edges come
from the kNN graph itself (no Delaunay step), so disconnected kNN components are oriented independently.
"""
import numpy as np

from . import cloud_ops


def orient_normals_consistent_tangent_plane(points, normals, k=100, ctx=None):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import breadth_first_order, connected_components, minimum_spanning_tree
    p = np.asarray(points, float).reshape(-1, 3)
    n = np.array(normals, float).reshape(-1, 3)
    N = len(p)
    if N < 2:
        return n
    nbr, _ = cloud_ops.knn_graph(p, min(k + 1, N), want_d2=False, ctx=ctx)
    i = np.repeat(np.arange(N), nbr.shape[1] - 1)
    j = nbr[:, 1:].reshape(-1)
    ok = j >= 0
    i, j = i[ok], j[ok]
    w = 1.0 - np.abs((n[i] * n[j]).sum(1)) + 1e-12          # strictly positive so that csgraph keeps the edge
    g = coo_matrix((w, (i, j)), shape=(N, N)).tocsr()
    g = g.maximum(g.T)
    mst = minimum_spanning_tree(g)
    mst = mst.maximum(mst.T).tocsr()
    ncomp, lab = connected_components(mst, directed=False)
    for c in range(ncomp):
        members = np.nonzero(lab == c)[0]
        root = members[np.argmax(p[members, 2])]
        if n[root, 2] < 0:
            n[root] = -n[root]
        order, pred = breadth_first_order(mst, root, directed=False, return_predecessors=True)
        for v in order[1:]:
            if np.dot(n[v], n[pred[v]]) < 0:
                n[v] = -n[v]
    return n
