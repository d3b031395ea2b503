/* CPU oracle helper -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Kruskal's minimum spanning forest as Open3D's geometry/EstimateNormals.cpp uses it inside
 * PointCloud::OrientNormalsConsistentTangentPlane (the call behind /root/reference/normal_estimation.py:21):
 * edges are visited in ascending weight, an edge is kept when its end points lie in different sets of a
 * disjoint-set forest.  Open3D is un-vendored and un-pinned in the reference; this restates its published
 * algorithm [recalled].  The caller passes the visiting order (numpy lexsort by (weight, v0, v1)): the original's
 * std::sort leaves the order of equal weights unspecified, so a fixed total order is the oracle's convention.
 */
#include <stdint.h>
#include <stdlib.h>

static int64_t find_root(int64_t *parent, int64_t v) {
    while (parent[v] != v) {
        parent[v] = parent[parent[v]];
        v = parent[v];
    }
    return v;
}

/* order[m]: edge indices in visiting order; v0/v1[m]: end points; kept[m] (out): 1 when the edge is in the forest.
 * Returns the number of kept edges, or -1 on allocation failure / bad vertex index. */
int64_t r3d_oracle_kruskal(int64_t n_vertices, int64_t m, const int64_t *order, const int64_t *v0, const int64_t *v1,
                           uint8_t *kept) {
    int64_t *parent = (int64_t *)malloc((size_t)(n_vertices > 0 ? n_vertices : 1) * sizeof(int64_t));
    if (!parent) return -1;
    for (int64_t i = 0; i < n_vertices; i++) parent[i] = i;
    for (int64_t e = 0; e < m; e++) kept[e] = 0;
    int64_t n_kept = 0;
    for (int64_t k = 0; k < m; k++) {
        const int64_t e = order[k];
        if (e < 0 || e >= m || v0[e] < 0 || v0[e] >= n_vertices || v1[e] < 0 || v1[e] >= n_vertices) {
            free(parent);
            return -1;
        }
        const int64_t a = find_root(parent, v0[e]), b = find_root(parent, v1[e]);
        if (a == b) continue;
        parent[a] = b;
        kept[e] = 1;
        if (++n_kept == n_vertices - 1) break;
    }
    free(parent);
    return n_kept;
}
