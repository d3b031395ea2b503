// cloud.hip -- point-cloud half of the hot path for gfx950: uniform-grid neighbour search, voxel down-sampling,
// hybrid / kNN PCA normals and the fused nearest-neighbour + Gauss-Newton reduction of ICP / point-to-plane / GICP.
//
// Replaces the Open3D (legacy, float64) calls the reference makes; semantics are those pinned in
// oracle/cloud_oracle.py (fixture-verified for voxel / normals / SOR, [recalled] Open3D for registration):
//   voxel_down_sample          pointcloud_alignment.py:22-23
//   estimate_normals(Hybrid)   pointcloud_alignment.py:27-28, test/GICP1.py:77, normal_estimation.py:20
//   registration_icp           pointcloud_alignment.py:35-39
//   registration_generalized_icp   test/GICP1.py:99-102     point-to-plane: test/check2.py:151-154
//   source.transform(T)        pointcloud_alignment.py:42
// All geometry is float64 like the legacy Open3D classes (MI355X runs fp64 vector math at half the fp32 rate,
// and these kernels are latency / LDS bound, not flop bound), which keeps voxel membership and neighbour sets
// identical to the oracle's instead of "within tolerance".

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "r3d_internal.h"

namespace {

struct GridView {
    double ox, oy, oz, cell, inv_cell;
    int nx, ny, nz;
    // Cell table, two levels.  cstart(c) = number of points whose cell index is below c = first sorted slot of cell c (x is the
    // fastest key digit, so cells adjacent in x hold ONE contiguous run of points).  The linear cell index is cut into LINES of 32
    // cells; l1[line] >= 0: no point in cells [32 line, 32 line + 36): all those cells share that cstart value;
    // l1[line] < 0: the line is materialised as l2[(-l1 - 1) * 36 + w], w = 0 .. 35 (the last four entries repeat the first four
    // cells of the next line, so a 16-byte read of four consecutive entries never leaves its line).  A surface occupies a few per
    // cent of a dense grid: at 1 M points the dense int table was 144 MB and most of a query's L2 misses; l1 is 4.5 MB, l2 ~15 MB.
    const int *l1, *l2;
    // ... or, when `dense` is set, the plain table dense[c] = cstart(c), c = 0 .. ncells + 4 (no dependent second load: what the
    // registration loop's search uses; its set-up pays three passes over the whole table instead)
    const int *dense;
    const double *pts;     // [n][3] points in cell-sorted order
    const int *idx;        // [n] sorted slot -> original index
    // optional float32 structure-of-arrays copy of pts (n + 4 entries each) for the two-stage 1-NN search of the
    // registration loop; fe = 2 x (bound on |float distance - exact distance|), see nn_block_top4
    const float *fx, *fy, *fz;
    float fe;
    // optional packed copy for the default search (nn_block_q10): one dword per sorted point = 10-bit offsets inside its own
    // cell (x | y << 10 | z << 20) and the low two bits of its cell x index (<< 30); n + 4 entries
    const unsigned *q10;
    // optional (registration loop): reach[coarse cell] != 0 iff some point lies in the 3x3x3 block of COARSE cells (rs fine cells
    // on a side) around it.  A query whose coarse cell reads 0 has no point within rs fine cells in any direction, i.e. none within
    // the correspondence radius (rs >= ceil(radius / cell)): its search is skipped outright, with the result it would have had.
    const unsigned char *reach;
    int rs, rnx, rny, rnz;
    int pool;              // registration loop: the packed search pools the lanes' row lists per wave (nn_block_q10's pooled walk)
};

__device__ __forceinline__ int cell_coord(double v, double o, double inv) { return (int)floor((v - o) * inv); }

constexpr int CL_SHIFT = 5, CL_STRIDE = 36;
typedef int cs_int4 __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte global load
// loads are issued unconditionally from a clamped address and selected afterwards (a branch around a load costs a wait per load)
__device__ __forceinline__ int cs1(const GridView &g, int64_t c) {
    if (g.dense) return g.dense[c];   // wave-uniform branch
    const int v = g.l1[c >> CL_SHIFT];
    const int e = g.l2[(int64_t)(v < 0 ? -v - 1 : 0) * CL_STRIDE + (int)(c & 31)];
    return v < 0 ? e : v;
}
// cstart of the four consecutive cells c .. c + 3
__device__ __forceinline__ cs_int4 cs4(const GridView &g, int64_t c) {
    if (g.dense) return *(const cs_int4 *)(g.dense + c);   // wave-uniform branch
    const int v = g.l1[c >> CL_SHIFT];
    const cs_int4 e = *(const cs_int4 *)(g.l2 + (int64_t)(v < 0 ? -v - 1 : 0) * CL_STRIDE + (int)(c & 31));
    cs_int4 r;
    r.x = v < 0 ? e.x : v; r.y = v < 0 ? e.y : v; r.z = v < 0 ? e.z : v; r.w = v < 0 ? e.w : v;
    return r;
}

// ------------------------------------------------------------------------------------------------ bbox
__global__ void __launch_bounds__(256) k_bbox_partial(const double *__restrict__ p, int64_t n, double *__restrict__ part) {
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        for (int a = 0; a < 3; a++) { double v = p[i * 3 + a]; mn[a] = fmin(mn[a], v); mx[a] = fmax(mx[a], v); }
    __shared__ double sm[6][256];
    for (int a = 0; a < 3; a++) { sm[a][threadIdx.x] = mn[a]; sm[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; a++) {
                sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + s]);
                sm[3 + a][threadIdx.x] = fmax(sm[3 + a][threadIdx.x], sm[3 + a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = sm[threadIdx.x][0];
}

// the same over a point count that is still on the device (n = *c0 + *c1: the last entries of a compaction's scan and flags):
// the bounding box is enqueued before the host knows how many points there are, and both come back in ONE read
__global__ void __launch_bounds__(256) k_bbox_partial_dn(const double *__restrict__ p, const int *__restrict__ c0, const int *__restrict__ c1,
                                                         double *__restrict__ part) {
    const int64_t n = (int64_t)*c0 + *c1;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        for (int a = 0; a < 3; a++) { double v = p[i * 3 + a]; mn[a] = fmin(mn[a], v); mx[a] = fmax(mx[a], v); }
    __shared__ double sm[6][256];
    for (int a = 0; a < 3; a++) { sm[a][threadIdx.x] = mn[a]; sm[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; a++) {
                sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + s]);
                sm[3 + a][threadIdx.x] = fmax(sm[3 + a][threadIdx.x], sm[3 + a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = sm[threadIdx.x][0];
}

// second stage: one workgroup folds the per-workgroup boxes into one (6 doubles), so the host reads 48 bytes
__global__ void __launch_bounds__(256) k_bbox_final(const double *__restrict__ part, int nb, double *__restrict__ out6) {
    __shared__ double sm[6][256];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int b = threadIdx.x; b < nb; b += 256)
        for (int a = 0; a < 3; a++) { mn[a] = fmin(mn[a], part[b * 6 + a]); mx[a] = fmax(mx[a], part[b * 6 + 3 + a]); }
    for (int a = 0; a < 3; a++) { sm[a][threadIdx.x] = mn[a]; sm[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; a++) {
                sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + s]);
                sm[3 + a][threadIdx.x] = fmax(sm[3 + a][threadIdx.x], sm[3 + a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out6[threadIdx.x] = sm[threadIdx.x][0];
}

// ------------------------------------------------------------------------------------------------ keys
// KEY = unsigned (the key space fits 32 bits: half the bytes through every radix pass) or unsigned long long
template <class KEY>
__global__ void __launch_bounds__(256) k_cell_keys(const double *__restrict__ p, int64_t n, double ox, double oy, double oz,
                                                   double cell, int nx, int ny, int nz, int key_order,
                                                   KEY *__restrict__ keys, int *__restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    if (key_order == 1) {  // voxel keys: the legacy index is floor((p - origin) / voxel) with a true division
        cx = (int)floor((p[i * 3] - ox) / cell); cy = (int)floor((p[i * 3 + 1] - oy) / cell); cz = (int)floor((p[i * 3 + 2] - oz) / cell);
    } else if (key_order == 3) {  // tensor voxel keys: floor(float(p) / float(voxel)) in float32, grid origin 0; ox..oz = smallest key
        const float v = (float)cell;
        cx = (int)((long long)floorf((float)p[i * 3] / v) - (long long)ox);
        cy = (int)((long long)floorf((float)p[i * 3 + 1] / v) - (long long)oy);
        cz = (int)((long long)floorf((float)p[i * 3 + 2] / v) - (long long)oz);
    } else {               // search grid: must agree with cell_coord() used by the queries
        const double inv = 1.0 / cell;
        cx = cell_coord(p[i * 3], ox, inv); cy = cell_coord(p[i * 3 + 1], oy, inv); cz = cell_coord(p[i * 3 + 2], oz, inv);
    }
    cx = min(max(cx, 0), nx - 1); cy = min(max(cy, 0), ny - 1); cz = min(max(cz, 0), nz - 1);
    // key_order 0: x fastest (search grid); 1: z fastest (voxel output in lexicographic (kx,ky,kz) order);
    // 2: Morton code of the cell coordinates (query ordering: consecutive points stay compact in all three axes)
    unsigned long long k;
    if (key_order == 2) {
        auto spread = [](unsigned long long v) {  // 21 bits -> every third bit
            v &= 0x1fffffull;
            v = (v | v << 32) & 0x1f00000000ffffull;
            v = (v | v << 16) & 0x1f0000ff0000ffull;
            v = (v | v << 8) & 0x100f00f00f00f00full;
            v = (v | v << 4) & 0x10c30c30c30c30c3ull;
            v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        k = spread((unsigned long long)cx) | spread((unsigned long long)cy) << 1 | spread((unsigned long long)cz) << 2;
    } else {
        k = key_order == 0 ? ((unsigned long long)cz * ny + cy) * nx + cx : ((unsigned long long)cx * ny + cy) * nz + cz;   // 1 and 3: z fastest
    }
    keys[i] = (KEY)k;
    vals[i] = (int)i;
}

// population of every cell: the last point of a run of equal (sorted) keys knows the run length
template <class KEY>
__global__ void __launch_bounds__(256) k_cell_counts(const KEY *__restrict__ keys, int64_t n, int *__restrict__ run_start,
                                                     int *__restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const KEY k = keys[i];
    if (i == 0 || keys[i - 1] != k) run_start[k] = (int)i;
}
template <class KEY>
__global__ void __launch_bounds__(256) k_cell_counts2(const KEY *__restrict__ keys, int64_t n, const int *__restrict__ run_start,
                                                      int *__restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const KEY k = keys[i];
    if (i == n - 1 || keys[i + 1] != k) counts[k] = (int)i + 1 - run_start[k];
}

__global__ void __launch_bounds__(256) k_gather3(const double *__restrict__ src, const int *__restrict__ idx, int64_t n, double *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int j = idx[i];
    dst[i * 3] = src[(int64_t)j * 3]; dst[i * 3 + 1] = src[(int64_t)j * 3 + 1]; dst[i * 3 + 2] = src[(int64_t)j * 3 + 2];
}

// ================================================================================== hand-written exclusive scan (int32 / packed 64)
// Reduce-then-scan in two launches: k_scan_sums leaves one sum per 4096-element tile; k_scan_apply first adds up the tile sums in
// front of its own tile (the list is short: 2 000 tiles for an 8 MP image, read from L2) and then scans its tile.  Integer adds:
// the result does not depend on the order, so it is deterministic.  T = int (flags, cell populations) or unsigned long long
// (two counters packed: points in the low word, runs in the high word; both stay below 2^31 so no carry crosses).
constexpr int SCAN_T = 256, SCAN_I = 16, SCAN_TILE = SCAN_T * SCAN_I;
// MAXOP: the running maximum instead of the running sum (values >= 0; used to carry "points so far" across empty table lines)
template <class T, bool MAXOP = false>
__device__ __forceinline__ T scan_op(T a, T b) { return MAXOP ? (a > b ? a : b) : a + b; }
template <class T, bool MAXOP = false>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(v, o);
        if (lane >= o) v = scan_op<T, MAXOP>(v, t);
    }
    return v;
}
// one value per thread of a 256-thread workgroup -> exclusive prefix; *total = workgroup sum
template <class T, bool MAXOP = false>
__device__ __forceinline__ T block_excl_scan(T v, T *total, T *ws /* 4 entries of LDS */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T incl = wave_incl_scan<T, MAXOP>(v, lane);
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { if (i < w) off = scan_op<T, MAXOP>(off, ws[i]); tot = scan_op<T, MAXOP>(tot, ws[i]); }
    __syncthreads();
    *total = tot;
    if (MAXOP) {   // exclusive = the inclusive value of the lane before (0 for the first lane of the wave), joined with the waves before
        T prev = __shfl_up(incl, 1);
        if (lane == 0) prev = 0;
        return scan_op<T, true>(off, prev);
    }
    return off + incl - v;
}
// 16 consecutive elements of a thread as 16-byte accesses when the whole group is inside the array (the arrays come from the arena:
// 256-byte aligned, and base is a multiple of 16 elements)
template <class T>
__device__ __forceinline__ void scan_load16(const T *__restrict__ in, int64_t base, int64_t n, T v[SCAN_I]) {
    if (base + SCAN_I <= n) {
        constexpr int PER = 16 / sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(PER)));
        const vec_t *p = reinterpret_cast<const vec_t *>(in + base);
#pragma unroll
        for (int q = 0; q < SCAN_I / PER; q++) {
            const vec_t t = p[q];
#pragma unroll
            for (int c = 0; c < PER; c++) v[q * PER + c] = t[c];
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_I; j++) v[j] = base + j < n ? in[base + j] : (T)0;
    }
}
template <class T, bool MAXOP = false>
__global__ void __launch_bounds__(SCAN_T) k_scan_sums(const T *__restrict__ in, int64_t n, T *__restrict__ part) {
    __shared__ T ws[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    T v[SCAN_I], s = 0;
    scan_load16<T>(in, base, n, v);
#pragma unroll
    for (int j = 0; j < SCAN_I; j++) s = scan_op<T, MAXOP>(s, v[j]);
    T tot;
    (void)block_excl_scan<T, MAXOP>(s, &tot, ws);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// hi_max (T = unsigned long long only): the largest HIGH word of any input element (runs per bucket), by atomicMax
template <class T, bool MAXOP = false>
__global__ void __launch_bounds__(SCAN_T) k_scan_apply(const T *__restrict__ in, int64_t n, const T *__restrict__ part, T *__restrict__ out,
                                                       int *__restrict__ lo_out, int *__restrict__ hi_max) {
    __shared__ T ws[4];
    T acc = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_T) acc = scan_op<T, MAXOP>(acc, part[b]);
    T tile_off;
    (void)block_excl_scan<T, MAXOP>(acc, &tile_off, ws);
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    T v[SCAN_I], s = 0;
    int hm = 0;
    scan_load16<T>(in, base, n, v);
#pragma unroll
    for (int j = 0; j < SCAN_I; j++) {
        if (sizeof(T) == 8) hm = max(hm, (int)((unsigned long long)v[j] >> 32));
        s = scan_op<T, MAXOP>(s, v[j]);
    }
    T tot;
    T run = scan_op<T, MAXOP>(tile_off, block_excl_scan<T, MAXOP>(s, &tot, ws));
    if (base + SCAN_I <= n && !lo_out) {
        constexpr int PER = 16 / sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(PER)));
        vec_t *p = reinterpret_cast<vec_t *>(out + base);
#pragma unroll
        for (int q = 0; q < SCAN_I / PER; q++) {
            vec_t t;
#pragma unroll
            for (int c = 0; c < PER; c++) { t[c] = run; run = scan_op<T, MAXOP>(run, v[q * PER + c]); }
            p[q] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SCAN_I; j++) {
            if (base + j < n) {
                out[base + j] = run;
                if (lo_out) lo_out[base + j] = (int)(unsigned)run;
            }
            run = scan_op<T, MAXOP>(run, v[j]);
        }
    }
    if (sizeof(T) == 8 && hi_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hm = max(hm, __shfl_xor(hm, o));
        if ((threadIdx.x & 63) == 0 && hm > 0) atomicMax(hi_max, hm);
    }
}

// ================================================================================== stable counting sort by cell key, run based
// Sorts point indices by (key, index) WITHOUT a radix pass.  Consecutive points with the same key (the pixels of an image row that
// fall into one voxel; the members of a voxelised cloud in key order) form a RUN; a run is one record (key, first index,
// length).  k_cs_runs detects runs inside every wave (64 consecutive points), counts per BUCKET -- a coarse prefix of the key: the
// (kx, ky) column of a z-fastest voxel key, the high bits of a Morton or x-fastest key -- how many runs and points it receives
// (one 64-bit atomic per run head: runs << 32 | points) and hands every run a slot in a global list.  After a scan of the bucket
// table, k_cs_place files every run under its bucket, k_cs_rank gives every run its final position by counting, among the runs
// of ITS bucket, the points of those that sort before it by (key, first index) -- the only quadratic step, in runs per bucket
// (tens) -- and k_cs_emit writes every point to final base + offset inside its run.  Arrival order (the atomics) only picks
// scratch slots: the result is the unique stable order, identical to a stable radix sort.  The bucket table is small (L2
// resident); the point arrays are read and written once each.
template <class KEY>
struct CsRun { KEY key; int start, len; unsigned bucket; int gid; };
struct CsGeom { double ox, oy, oz, cell; int nx, ny, nz, key_order, shift; unsigned long long div; };   // bucket = key >> shift, or key / div when shift < 0

template <class KEY>
__device__ __forceinline__ void cs_key(const double *__restrict__ p, int64_t i, const CsGeom &g, KEY &full, unsigned &bucket) {
    int cx, cy, cz;
    if (g.key_order == 1) {  // voxel keys: the legacy index is floor((p - origin) / voxel) with a true division
        cx = (int)floor((p[i * 3] - g.ox) / g.cell); cy = (int)floor((p[i * 3 + 1] - g.oy) / g.cell); cz = (int)floor((p[i * 3 + 2] - g.oz) / g.cell);
    } else if (g.key_order == 3) {  // tensor voxel keys: floor(float(p) / float(voxel)) in float32, grid origin 0; ox..oz = smallest key
        const float v = (float)g.cell;
        cx = (int)((long long)floorf((float)p[i * 3] / v) - (long long)g.ox);
        cy = (int)((long long)floorf((float)p[i * 3 + 1] / v) - (long long)g.oy);
        cz = (int)((long long)floorf((float)p[i * 3 + 2] / v) - (long long)g.oz);
    } else {               // search grid: must agree with cell_coord() used by the queries
        const double inv = 1.0 / g.cell;
        cx = cell_coord(p[i * 3], g.ox, inv); cy = cell_coord(p[i * 3 + 1], g.oy, inv); cz = cell_coord(p[i * 3 + 2], g.oz, inv);
    }
    cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
    unsigned long long k;
    if (g.key_order == 2) {
        auto spread = [](unsigned long long v) {  // 21 bits -> every third bit
            v &= 0x1fffffull;
            v = (v | v << 32) & 0x1f00000000ffffull;
            v = (v | v << 16) & 0x1f0000ff0000ffull;
            v = (v | v << 8) & 0x100f00f00f00f00full;
            v = (v | v << 4) & 0x10c30c30c30c30c3ull;
            v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        k = spread((unsigned long long)cx) | spread((unsigned long long)cy) << 1 | spread((unsigned long long)cz) << 2;
        bucket = (unsigned)(k >> g.shift);
    } else if (g.key_order == 0) {
        k = ((unsigned long long)cz * g.ny + cy) * g.nx + cx;
        bucket = (unsigned)(k >> g.shift);
    } else {   // 1 and 3: z fastest; bucket = a stretch of g.div consecutive keys (a piece of a (kx, ky) column)
        k = ((unsigned long long)cx * g.ny + cy) * g.nz + cz;
        bucket = sizeof(KEY) == 4 ? (unsigned)k / (unsigned)g.div : (unsigned)(k / g.div);
    }
    full = (KEY)k;
}

template <class KEY>
__global__ void __launch_bounds__(256) k_cs_runs(const double *__restrict__ p, int64_t n, CsGeom g, KEY *__restrict__ keys, int *__restrict__ gid_of,
                                                 unsigned long long *__restrict__ cnt, CsRun<KEY> *__restrict__ runs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < n;
    KEY key = 0;
    unsigned bucket = 0;
    if (valid) cs_key<KEY>(p, i, g, key, bucket);
    const KEY prev = __shfl_up(key, 1);
    const bool head = valid && (lane == 0 || prev != key);
    const unsigned long long hm = __ballot(head), vm = __ballot(valid);
    if (vm == 0) return;
    const unsigned long long upto = (2ull << lane) - 1ull;           // bits 0 .. lane (lane 63: all ones)
    const int hp = 63 - __clzll((long long)(hm & upto));              // lane of my run's head (lane 0 is always a head)
    const unsigned long long above = hm & ~upto;
    const int nxt = above ? __ffsll((long long)above) - 1 : __popcll(vm);   // valid lanes are a prefix of the wave
    // a run is named by the index of its first point (no global counter: one address hit by every wave of the grid serialises)
    if (head) {
        const int len = nxt - lane;
        const unsigned long long old = atomicAdd(&cnt[bucket], (1ull << 32) | (unsigned long long)len);
        CsRun<KEY> r;
        r.key = key; r.start = (int)i; r.len = len; r.bucket = bucket;
        r.gid = (int)(old >> 32);      // arrival number of the run inside its bucket (picks a scratch slot only) until k_cs_place
        runs[i] = r;
    }
    if (valid) { keys[i] = key; gid_of[i] = (int)i - (lane - hp); }
}
// files every run under its bucket: placed[run_start(bucket) + arrival number]; one thread per point, run heads act
template <class KEY>
__global__ void __launch_bounds__(256) k_cs_place(const CsRun<KEY> *__restrict__ runs, const int *__restrict__ gid_of, int64_t n,
                                                  const unsigned long long *__restrict__ start, CsRun<KEY> *__restrict__ placed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || gid_of[i] != (int)i) return;
    CsRun<KEY> r = runs[i];
    const int slot = (int)(start[r.bucket] >> 32) + r.gid;
    r.gid = (int)i;
    placed[slot] = r;
}
// final position of every run: points of its bucket's runs that sort before it by (key, first index).  Wave-cooperative: the 64
// runs of a wave lie in consecutive buckets, so the union of their buckets' run lists is ONE contiguous stretch of `placed`; the
// wave reads it 64 records at a time (one coalesced load) and every lane tests all 64 through v_readlane (uniform index), counting
// only the records of its own bucket.  Cost per wave: union length x ~8 instructions, whatever the split into buckets.
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return (unsigned long long)hi << 32 | lo;
}
template <class KEY>
__global__ void __launch_bounds__(256) k_cs_rank(const CsRun<KEY> *__restrict__ placed, int nruns, const unsigned long long *__restrict__ start,
                                                 int *__restrict__ final_base, const unsigned long long *__restrict__ d_total,
                                                 const int *__restrict__ d_maxruns, int max_runs_allowed) {
    // deferred mode (d_total != NULL): the host has NOT read the run count and the fallback test yet -- they travel with the
    // caller's next read-back -- so the grid covers the upper bound (one run per point), the count comes from device memory, and
    // a bucket population the quadratic step below must not be asked to rank makes the whole kernel a no-op (the host then
    // discards this sort and runs the radix sort)
    if (d_total) {
        if (*d_maxruns > max_runs_allowed) return;
        nruns = (int)(*d_total >> 32);
    }
    const int s = blockIdx.x * 256 + threadIdx.x;   // nruns = high word of the scan total, read by the host together with the fallback test
    if ((s & ~63) >= nruns) return;                  // whole wave beyond the list
    const bool valid = s < nruns;
    const CsRun<KEY> me = placed[valid ? s : nruns - 1];
    const unsigned long long b0 = start[me.bucket], b1 = start[me.bucket + 1];
    const int rb = (int)(b0 >> 32), re = (int)(b1 >> 32);
    int acc = (int)(unsigned)b0;
    const int u0 = __builtin_amdgcn_readfirstlane(rb);
    int u1 = re;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u1 = max(u1, __shfl_xor(u1, o));
    u1 = __builtin_amdgcn_readfirstlane(u1);
    const int lane = threadIdx.x & 63;
    for (int c = u0; c < u1; c += 64) {
        const CsRun<KEY> o = placed[min(c + lane, nruns - 1)];
        const int lim = min(64, u1 - c);
        for (int t = 0; t < lim; t++) {
            KEY ok;
            if (sizeof(KEY) == 8) ok = (KEY)readlane_u64((unsigned long long)o.key, t);
            else ok = (KEY)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)o.key, t);
            const int os = __builtin_amdgcn_readlane(o.start, t), ol = __builtin_amdgcn_readlane(o.len, t);
            const int j = c + t;
            const bool less = ok < me.key || (ok == me.key && os < me.start);
            acc += (j >= rb && j < re && less) ? ol : 0;
        }
    }
    if (valid) final_base[me.gid] = acc;
}
// every point to its final slot; optionally the gathered coordinates as well (the cell-sorted copy a search grid needs)
template <class KEY>
__global__ void __launch_bounds__(256) k_cs_emit(const KEY *__restrict__ keys, const int *__restrict__ gid_of, const int *__restrict__ final_base, int64_t n, KEY *__restrict__ keys_out, int *__restrict__ idx_out,
                                                 const double *__restrict__ pts, double *__restrict__ sorted, const int *__restrict__ d_maxruns, int max_runs_allowed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (d_maxruns && *d_maxruns > max_runs_allowed) return;           // deferred mode: k_cs_rank did not run, final_base holds nothing
    const int g = gid_of[i];                                         // = index of the first point of my run
    const int64_t o = (int64_t)final_base[g] + ((int)i - g);
    keys_out[o] = keys[i];
    idx_out[o] = (int)i;
    if (sorted) { sorted[o * 3] = pts[i * 3]; sorted[o * 3 + 1] = pts[i * 3 + 1]; sorted[o * 3 + 2] = pts[i * 3 + 2]; }
}

// ================================================================================== two-level cell table (GridView::l1 / l2)
// Built from the SORTED keys.  k_line_mark: which lines hold a point in their 36-cell reach, and, per line, one past the sorted
// slot of its last point (so that an exclusive running maximum over the lines gives "points before this line").  k_line_l1: the
// first level.  k_line_fill: every run start i (first slot of a cell c, previous occupied cell p) owns the cells (p, c]: their
// cstart is i; it writes that value into the materialised lines that reach into (p, c] -- those are the lines that contain p or
// c in their 36-cell reach (a materialised line holds an occupied cell q; for an entry cc <= q the owner's c lies in [cc, q], for
// cc > q its p lies in [q, cc): inside the line either way), at most four lines per run start.  Slot n is the virtual last
// run start (c = +infinity).
template <class KEY>
__global__ void __launch_bounds__(256) k_line_mark(const KEY *__restrict__ keys, int64_t n, int *__restrict__ mark, int *__restrict__ last1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long c = (long long)keys[i];
    if (i == 0 || (long long)keys[i - 1] != c) {
        mark[c >> CL_SHIFT] = 1;
        if ((c & 31) < CL_STRIDE - 32 && c >= 32) mark[(c >> CL_SHIFT) - 1] = 1;
    }
    if (i == n - 1 || ((long long)keys[i + 1] >> CL_SHIFT) != (c >> CL_SHIFT)) last1[c >> CL_SHIFT] = (int)i + 1;
}
__global__ void __launch_bounds__(256) k_line_l1(const int *__restrict__ mark, const int *__restrict__ ids, const int *__restrict__ before,
                                                 int64_t nl, int *__restrict__ l1) {
    const int64_t L = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (L >= nl) return;
    l1[L] = mark[L] ? -(ids[L] + 1) : before[L];
}
template <class KEY>
__global__ void __launch_bounds__(256) k_line_fill(const KEY *__restrict__ keys, int64_t n, const int *__restrict__ l1, int64_t nl, int *__restrict__ l2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    const long long p = i > 0 ? (long long)keys[i - 1] : -1, c = i < n ? (long long)keys[i] : (long long)nl * 32 + CL_STRIDE;
    if (i < n && i > 0 && p == c) return;            // not a run start
    long long cand[4] = {(p >> CL_SHIFT) - 1, p >> CL_SHIFT, (c >> CL_SHIFT) - 1, c >> CL_SHIFT};   // (p = -1: lines -2 and -1, skipped)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const long long L = cand[q];
        bool dup = false;
#pragma unroll
        for (int r = 0; r < 4; r++) dup = dup || (r < q && cand[r] == L);
        if (dup || L < 0 || L >= nl) continue;
        const int v = l1[L];
        if (v >= 0) continue;                        // not materialised
        int *e = l2 + (int64_t)(-v - 1) * CL_STRIDE;
        const long long c0 = L * 32;
        const int w0 = (int)max(p + 1 - c0, 0ll), w1 = (int)min(c - c0, (long long)CL_STRIDE - 1);
        for (int w = w0; w <= w1; w++) e[w] = (int)i;
    }
}

// ------------------------------------------------------------------------------------------------ voxel
// segment heads of the sorted key array -> compact list of segment starts (one output voxel per segment)
template <class KEY>
__global__ void __launch_bounds__(256) k_seg_flags(const KEY *__restrict__ keys, int64_t n, int *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i - 1] != keys[i]) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_seg_starts(const int *__restrict__ flags, const int *__restrict__ scan, int64_t n, int *__restrict__ starts) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) starts[scan[i]] = (int)i;
}
// Voxel means: sequential sums in ORIGINAL point order (the sort is stable), exactly like the accumulate-then-divide of the legacy
// VoxelDownSample (T = double) / of the tensor method (T = float: members rounded to float32, float32 sums, float32 count).
// One thread per voxel keeps the order, and the kernel then lasts as long as its most populated voxel's chain of round trips
// (1 400 members in the near field of an 8 MP view: 175 rounds of 8, 0.30 ms for a kernel that moves 0.2 GB).  So voxels with more
// than VM_BIG members are only LISTED by k_voxel_mean_t (one wave-aggregated atomic per wave) and summed by k_voxel_mean_big, a wave
// per voxel: 64 members per round trip (coalesced index read, 64 gathers in flight), added in order through v_readlane -- every
// lane carries the same running sum, the additions are the voxel's own sequential chain, nothing else waits on them.
// (A form that staged 64 voxels' members through LDS windows was measured 3x SLOWER: where voxels are larger than the window only
// one lane walks at a time; 16 members per thread and round trip instead of 8: 2x slower.)
constexpr int VM_BIG = 192;
template <class T>
__global__ void __launch_bounds__(256) k_voxel_mean_t(const double *__restrict__ a, const int *__restrict__ idx, const int *__restrict__ starts,
                                                      int64_t nseg, int64_t n, double *__restrict__ out, int *__restrict__ big_list,
                                                      int *__restrict__ big_count) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = s < nseg;
    const int b = live ? starts[s] : 0, e = live ? (s + 1 < nseg ? starts[s + 1] : (int)n) : 0;
    const bool big = e - b > VM_BIG;
    const unsigned long long bm = __ballot(big);
    if (bm) {   // wave-uniform
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(big_count, __popcll(bm));
        base = __shfl(base, 0);
        if (big) big_list[base + __popcll(bm & ((1ull << lane) - 1ull))] = (int)s;
    }
    if (!live || big) return;
    T x = 0, y = 0, z = 0;
    for (int i0 = b; i0 < e; i0 += 8) {
        int64_t j[8];
        T X[8], Y[8], Z[8];
#pragma unroll
        for (int u = 0; u < 8; u++) j[u] = idx[min(i0 + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) { X[u] = (T)a[j[u] * 3]; Y[u] = (T)a[j[u] * 3 + 1]; Z[u] = (T)a[j[u] * 3 + 2]; }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u < e) { x += X[u]; y += Y[u]; z += Z[u]; }
    }
    const T c = (T)(e - b);
    out[s * 3] = (double)(x / c); out[s * 3 + 1] = (double)(y / c); out[s * 3 + 2] = (double)(z / c);
}
// the same sums over the points IN SORTED ORDER (the sort wrote them, k_cs_emit): a voxel's members are one contiguous stretch, so
// a lane streams whole cache lines instead of gathering 24 bytes from each; the next eight members are requested before the
// current eight are added
template <class T, int R>
__global__ void __launch_bounds__(256) k_voxel_mean_sorted(const double *__restrict__ sp, const int *__restrict__ starts, int64_t nseg, int64_t n,
                                                           double *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const int b = starts[s], e = s + 1 < nseg ? starts[s + 1] : (int)n, m = e - b;
    const double *__restrict__ p = sp + (int64_t)b * 3;
    T x = 0, y = 0, z = 0;
    double cur[3 * R], nxt[3 * R];
#pragma unroll
    for (int u = 0; u < R; u++) {
        const int64_t q = (int64_t)min(u, m - 1) * 3;
        cur[u * 3] = p[q]; cur[u * 3 + 1] = p[q + 1]; cur[u * 3 + 2] = p[q + 2];
    }
    for (int i0 = 0; i0 < m; i0 += R) {
#pragma unroll
        for (int u = 0; u < R; u++) {
            const int64_t q = (int64_t)min(i0 + R + u, m - 1) * 3;
            nxt[u * 3] = p[q]; nxt[u * 3 + 1] = p[q + 1]; nxt[u * 3 + 2] = p[q + 2];
        }
#pragma unroll
        for (int u = 0; u < R; u++)
            if (i0 + u < m) { x += (T)cur[u * 3]; y += (T)cur[u * 3 + 1]; z += (T)cur[u * 3 + 2]; }
#pragma unroll
        for (int u = 0; u < 3 * R; u++) cur[u] = nxt[u];
    }
    const T c = (T)m;
    out[s * 3] = (double)(x / c); out[s * 3 + 1] = (double)(y / c); out[s * 3 + 2] = (double)(z / c);
}
__device__ __forceinline__ double readlane_t(double v, int l) { return __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), l)); }
__device__ __forceinline__ float readlane_t(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <class T>
__global__ void __launch_bounds__(64) k_voxel_mean_big(const double *__restrict__ a, const int *__restrict__ idx, const int *__restrict__ starts,
                                                       int64_t nseg, int64_t n, double *__restrict__ out, const int *__restrict__ big_list,
                                                       const int *__restrict__ big_count) {
    const int lane = threadIdx.x, nbig = *big_count;
    for (int q = blockIdx.x; q < nbig; q += gridDim.x) {
        const int64_t s = big_list[q];
        const int b = starts[s], e = s + 1 < nseg ? starts[s + 1] : (int)n;
        T x = 0, y = 0, z = 0;
        int64_t j = idx[min(b + lane, e - 1)];
        T X = (T)a[j * 3], Y = (T)a[j * 3 + 1], Z = (T)a[j * 3 + 2];
        for (int c = b; c < e; c += 64) {
            const int64_t jn = idx[min(c + 64 + lane, e - 1)];                      // next chunk in flight while this one is added
            const T Xn = (T)a[jn * 3], Yn = (T)a[jn * 3 + 1], Zn = (T)a[jn * 3 + 2];
            const int lim = min(64, e - c);
            for (int t = 0; t < lim; t++) { x += readlane_t(X, t); y += readlane_t(Y, t); z += readlane_t(Z, t); }
            X = Xn; Y = Yn; Z = Zn;
        }
        if (lane == 0) {
            const T cnt = (T)(e - b);
            out[s * 3] = (double)(x / cnt); out[s * 3 + 1] = (double)(y / cnt); out[s * 3 + 2] = (double)(z / cnt);
        }
    }
}
// ------------------------------------------------------------------------------------------------ search
// visits the cells of Chebyshev shell s around (cx,cy,cz); F(slot_begin, slot_end) is called per non-empty cell
template <class F>
__device__ __forceinline__ void for_shell(const GridView &g, int cx, int cy, int cz, int s, F &&f) {
    for (int dz = -s; dz <= s; dz++) {
        int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -s; dy <= s; dy++) {
            int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            const bool face = (dz == -s || dz == s || dy == -s || dy == s);
            const int step = face ? 1 : 2 * s;  // interior rows of the shell only touch dx = -s and dx = +s
            for (int dx = -s; dx <= s; dx += (step > 0 ? step : 1)) {
                int x = cx + dx;
                if (x < 0 || x >= g.nx) continue;
                int64_t c = ((int64_t)z * g.ny + y) * g.nx + x;
                int b = cs1(g, c), e = cs1(g, c + 1);
                if (e > b) f(b, e);
            }
        }
    }
}

// the 3x3x3 block around (cx,cy,cz) as NINE runs: the three cells of an x-row are contiguous in the sorted point array
template <class F>
__device__ __forceinline__ void for_block3(const GridView &g, int cx, int cy, int cz, F &&f) {
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    if (x0 > x1) return;
    for (int dz = -1; dz <= 1; dz++) {
        const int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; dy++) {
            const int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            const int64_t row = ((int64_t)z * g.ny + y) * g.nx;
            const int b = cs1(g, row + x0), e = cs1(g, row + x1 + 1);
            if (e > b) f(b, e);
        }
    }
}

// Shell s >= 2 walked by x-rows with pruning.  A row on a face of the shell is ONE contiguous run of the sorted point array
// (two table entries instead of 2 (2s+1)), trimmed in x to the cells that can still hold a useful point; a row inside the
// shell only contributes its two end cells; rows and cells whose nearest face is further than bound() are skipped.
// bound() is the squared distance beyond which a candidate is useless to the caller (re-read for every row, so it tightens
// as the caller's result improves); the comparison is strict and shrunk by more than the rounding of the cell faces, so a
// point at exactly the bound is still visited (ties are decided by the caller's total order).
template <class Bound, class F>
__device__ __forceinline__ void for_shell_rows(const GridView &g, double px, double py, double pz, int cx, int cy, int cz, int s,
                                               Bound &&bound, F &&visit) {
    constexpr double SLK = 1.0 - 1e-9;
    auto slab = [&](double q, double o, int c) {   // distance from coordinate q to the slab of cell c along one axis (0 inside)
        const double lo = o + (double)c * g.cell, hi = lo + g.cell;
        return q < lo ? lo - q : (q > hi ? q - hi : 0.0);
    };
    // rows are taken centre-out (0, -1, +1, -2, ...): near rows tighten the bound before the far ones are looked at
    for (int iz = 0; iz <= 2 * s; iz++) {
        const int dz = (iz & 1) ? -((iz + 1) >> 1) : (iz >> 1);
        const int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        const double sz = slab(pz, g.oz, z);
        if (sz * sz * SLK > bound()) continue;
        for (int iy = 0; iy <= 2 * s; iy++) {
            const int dy = (iy & 1) ? -((iy + 1) >> 1) : (iy >> 1);
            const int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            const double sy = slab(py, g.oy, y), syz = sy * sy + sz * sz, bnd = bound();
            if (syz * SLK > bnd) continue;
            const int64_t row = ((int64_t)z * g.ny + y) * g.nx;
            if (dz == -s || dz == s || dy == -s || dy == s) {
                // cells further than kx from the query's cell lie at least (kx - 1) * cell away in x
                const double rx = sqrt(fmax(bnd - syz * SLK, 0.0)) * g.inv_cell;
                const int kx = rx < (double)s ? (int)rx + 1 : s;
                const int x0 = max(cx - min(kx, s), 0), x1 = min(cx + min(kx, s), g.nx - 1);
                if (x0 > x1) continue;
                const int b = cs1(g, row + x0), e = cs1(g, row + x1 + 1);
                if (e > b) visit(b, e);
            } else {
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int x = side ? cx + s : cx - s;
                    if (x < 0 || x >= g.nx) continue;
                    const double sx = slab(px, g.ox, x);
                    if ((sx * sx + syz) * SLK > bound()) continue;
                    const int b = cs1(g, row + x), e = cs1(g, row + x + 1);
                    if (e > b) visit(b, e);
                }
            }
        }
    }
}

// max shells needed to cover the whole grid from a clamped centre
__device__ __forceinline__ int max_shell(const GridView &g, int cx, int cy, int cz) {
    int m = max(max(cx, g.nx - 1 - cx), max(max(cy, g.ny - 1 - cy), max(cz, g.nz - 1 - cz)));
    return max(m, 0);
}

// ---- k nearest (<= k, distance < radius if radius > 0), sorted ascending, in LDS columns [slot][thread]
// Selection keeps the list as a binary MAX-HEAP in the total order (d2, original index) while candidates stream by: the root
// is the current k-th neighbour, an accepted candidate replaces it and sifts down (<= log2 k levels).  A sorted insertion
// costs every lane of the wave the longest shift chain of any lane at every candidate (PMC of that form: 71 k instructions
// per 64 queries, 53 % of them scalar loop control; k_normals 2.9 ms at 1 M points, 1.96 ms with a re-scanned unsorted list).
// The list is sorted once at the end, so callers see the same ascending order (and summation order) as before.
constexpr int KNN_BLOCK = 64;
template <bool unused = true>
__device__ __forceinline__ int knn_query(const GridView &g, double qx, double qy, double qz, int k, double radius,
                                         double *sd /*[k][KNN_BLOCK] column tid*/, int *si) {
    const int t = threadIdx.x;
    const double r2 = radius > 0 ? radius * radius : 1e300;
    int cnt = 0;
    double maxd = 0.0;      // valid when cnt == k: the heap's root (the largest (d2, index) of the list)
    int maxslot = 0;
    const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
    const int smax = max_shell(g, min(max(cx, 0), g.nx - 1), min(max(cy, 0), g.ny - 1), min(max(cz, 0), g.nz - 1)) + 1;
    // total order (d2, original index): exact-distance ties are common on voxelised / fp32-rounded clouds, and the oracle
    // breaks them the same way; the index loads stay off the hot path (only on an exact tie)
    auto greater = [&](double da, int sa, double db, int sb) {
        bool gt = da > db;
        if (da == db) gt = g.idx[sa] > g.idx[sb];
        return gt;
    };
    // puts (d2, i) at heap position pos and lets it sink to its place among the first n entries
    auto sift_down = [&](int pos, double d2, int i, int n) {
        for (;;) {
            int c = 2 * pos + 1;
            if (c >= n) break;
            double cd = sd[c * KNN_BLOCK + t];
            int cs = si[c * KNN_BLOCK + t];
            if (c + 1 < n) {
                const double rd = sd[(c + 1) * KNN_BLOCK + t];
                const int rs = si[(c + 1) * KNN_BLOCK + t];
                if (greater(rd, rs, cd, cs)) { c++; cd = rd; cs = rs; }
            }
            if (!greater(cd, cs, d2, i)) break;
            sd[pos * KNN_BLOCK + t] = cd;
            si[pos * KNN_BLOCK + t] = cs;
            pos = c;
        }
        sd[pos * KNN_BLOCK + t] = d2;
        si[pos * KNN_BLOCK + t] = i;
    };
    // candidates are fetched four at a time from clamped slots (no branch around a load)
    auto visit = [&](int b, int e) {
        for (int i0 = b; i0 < e; i0 += 4) {
            double X[4], Y[4], Z[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t ii = min(i0 + u, e - 1);
                X[u] = g.pts[ii * 3]; Y[u] = g.pts[ii * 3 + 1]; Z[u] = g.pts[ii * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u;
                if (i >= e) break;
                const double dx = X[u] - qx, dy = Y[u] - qy, dz = Z[u] - qz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < r2)) continue;
                if (cnt < k) {
                    sd[cnt * KNN_BLOCK + t] = d2;
                    si[cnt * KNN_BLOCK + t] = i;
                    if (++cnt == k) {                       // list full: make it a heap
                        for (int p = k / 2 - 1; p >= 0; p--) sift_down(p, sd[p * KNN_BLOCK + t], si[p * KNN_BLOCK + t], k);
                        maxd = sd[t]; maxslot = si[t];
                    }
                } else if (greater(maxd, maxslot, d2, i)) {
                    sift_down(0, d2, i, k);
                    maxd = sd[t]; maxslot = si[t];
                }
            }
        }
    };
    for (int s = 1; s <= smax; s++) {
        if (s == 1) for_block3(g, cx, cy, cz, visit);      // shells 0 and 1 as nine contiguous runs
        else for_shell_rows(g, qx, qy, qz, cx, cy, cz, s, [&]() { return cnt == k ? fmin(maxd, r2) : r2; }, visit);
        // everything not visited yet is at least s*cell away
        const double reach = s * g.cell;
        if (radius > 0 && reach >= radius) break;
        if (cnt == k && maxd <= reach * reach) break;
    }
    // ascending (d2, index): a full list is a heap, so heap-sort it in place (pop the maximum to the end, k - 1 times);
    // a list that never filled is in arrival order: heapify it first
    if (cnt < k)
        for (int p = cnt / 2 - 1; p >= 0; p--) sift_down(p, sd[p * KNN_BLOCK + t], si[p * KNN_BLOCK + t], cnt);
    for (int n = cnt - 1; n > 0; n--) {
        const double ld = sd[n * KNN_BLOCK + t], rd = sd[t];
        const int ls = si[n * KNN_BLOCK + t], rs = si[t];
        sd[n * KNN_BLOCK + t] = rd;
        si[n * KNN_BLOCK + t] = rs;
        sift_down(0, ld, ls, n);
    }
    return cnt;
}

// Normal of a neighbourhood covariance as legacy Open3D computes it: FastEigen3x3 [recalled: geometry/EstimateNormals.cpp, the
// closed form of geometrictools' RobustEigenSymmetric3x3] -- eigenvalues from the trigonometric solution of the characteristic
// cubic of A = cov / max coefficient, the eigenvector of the better-conditioned extreme eigenvalue from the largest cross product
// of two rows of A - lambda I, the middle one from the 2x2 problem in its orthogonal complement, the third as their cross
// product; returns the eigenvector of the smallest eigenvalue WITH THE SIGN THE FORMULAS PRODUCE (the reference's recorded
// normals carry exactly that sign).  The fused multiply-adds are placed explicitly (the library is built with
// -ffp-contract=off): they are the ones of the build that recorded the reference's frames, found by matching its 2.14 M recorded
// normals (oracle/normals.c header, profiles/r03_pin_normals.json: 89 % bit-equal sign included, all within 2e-12).  Cheaper
// than the cyclic Jacobi sweeps it replaces, and a degenerate neighbourhood now has the answer the original's formulas give.
__device__ __forceinline__ double fe_mulsub(double a, double b, double c, double d) { return fma(a, b, -(c * d)); }   // a*b - c*d
__device__ __forceinline__ void fe_cross(const double a[3], const double b[3], double o[3]) {
    o[0] = fe_mulsub(a[1], b[2], a[2], b[1]);
    o[1] = fe_mulsub(a[2], b[0], a[0], b[2]);
    o[2] = fe_mulsub(a[0], b[1], a[1], b[0]);
}
__device__ __forceinline__ void fe_eigenvector0(const double A[6], double ev, double o[3]) {
    const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    fe_cross(r0, r1, c01);
    fe_cross(r0, r2, c02);
    fe_cross(r1, r2, c12);
    const double d0 = fma(c01[2], c01[2], c01[0] * c01[0] + c01[1] * c01[1]);    // Eigen's 3-vector dot: packet of two + scalar tail
    const double d1 = fma(c02[2], c02[2], c02[0] * c02[0] + c02[1] * c02[1]);
    const double d2 = fma(c12[2], c12[2], c12[0] * c12[0] + c12[1] * c12[1]);
    // imax as the original: d1 replaces d0 if larger, d2 replaces the running maximum if larger (static selects only)
    const bool s1 = d1 > d0;
    const double dm = s1 ? d1 : d0;
    const bool s2 = d2 > dm;
    const double x = s2 ? c12[0] : (s1 ? c02[0] : c01[0]), y = s2 ? c12[1] : (s1 ? c02[1] : c01[1]), z = s2 ? c12[2] : (s1 ? c02[2] : c01[2]);
    const double l = sqrt(s2 ? d2 : dm);
    o[0] = x / l; o[1] = y / l; o[2] = z / l;
}
__device__ __forceinline__ void fe_eigenvector1(const double A[6], const double e0[3], double ev1, double o[3]) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        const double il = 1.0 / sqrt(fma(e0[0], e0[0], e0[2] * e0[2]));
        U[0] = -e0[2] * il; U[1] = 0.0; U[2] = e0[0] * il;
    } else {
        const double il = 1.0 / sqrt(fma(e0[1], e0[1], e0[2] * e0[2]));
        U[0] = 0.0; U[1] = e0[2] * il; U[2] = -e0[1] * il;
    }
    fe_cross(e0, U, V);
#define FE_SUM3(x0, y0, x1, y1, x2, y2) fma((x2), (y2), fma((x0), (y0), (x1) * (y1)))
    const double AU[3] = {FE_SUM3(A[0], U[0], A[1], U[1], A[2], U[2]), FE_SUM3(A[1], U[0], A[3], U[1], A[4], U[2]),
                          FE_SUM3(A[2], U[0], A[4], U[1], A[5], U[2])};
    const double AV[3] = {FE_SUM3(A[0], V[0], A[1], V[1], A[2], V[2]), FE_SUM3(A[1], V[0], A[3], V[1], A[4], V[2]),
                          FE_SUM3(A[2], V[0], A[4], V[1], A[5], V[2])};
    double m00 = FE_SUM3(U[0], AU[0], U[1], AU[1], U[2], AU[2]) - ev1;
    double m01 = FE_SUM3(U[0], AV[0], U[1], AV[1], U[2], AV[2]);
    double m11 = FE_SUM3(V[0], AV[0], V[1], AV[1], V[2], AV[2]) - ev1;
#undef FE_SUM3
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    double cu, cv;   // result = cu * U - cv * V
    if (a00 >= a11) {
        if (!(fmax(a00, a01) > 0)) { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; return; }
        if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(fma(m01, m01, 1.0)); m01 *= m00; }
        else { m00 /= m01; m01 = 1.0 / sqrt(fma(m00, m00, 1.0)); m00 *= m01; }
        cu = m01; cv = m00;
    } else {
        if (!(fmax(a11, a01) > 0)) { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; return; }
        if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(fma(m01, m01, 1.0)); m01 *= m11; }
        else { m11 /= m01; m01 = 1.0 / sqrt(fma(m11, m11, 1.0)); m11 *= m01; }
        cu = m11; cv = m01;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = fe_mulsub(cu, U[i], cv, V[i]);
}
// cov as c00 c01 c02 c11 c12 c22; n = (0,0,0) for an all-zero matrix (the caller substitutes (0,0,1) / the previous normal)
__device__ __forceinline__ void fast_eigen3x3(double c00, double c01, double c02, double c11, double c12, double c22, double n[3]) {
    const double mx = fmax(fmax(fmax(c00, c01), fmax(c02, c11)), fmax(c12, c22));
    n[0] = n[1] = n[2] = 0.0;
    if (mx == 0.0) return;
    const double A[6] = {c00 / mx, c01 / mx, c02 / mx, c11 / mx, c12 / mx, c22 / mx};
    const double norm = (A[1] * A[1] + A[2] * A[2]) + A[4] * A[4];
    if (!(norm > 0.0)) {   // diagonal matrix: the axis of the strictly smallest diagonal entry, z otherwise
        const double a00 = A[0] * mx, a11 = A[3] * mx, a22 = A[5] * mx;
        if (a00 < a11 && a00 < a22) n[0] = 1.0;
        else if (a11 < a00 && a11 < a22) n[1] = 1.0;
        else n[2] = 1.0;
        return;
    }
    const double q = (A[0] + A[3] + A[5]) / 3.0;
    const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
    const double p = sqrt(fma(norm, 2.0, fma(b22, b22, fma(b00, b00, b11 * b11))) / 6.0);
    const double k00 = fe_mulsub(b11, b22, A[4], A[4]);
    const double k01 = fe_mulsub(A[1], b22, A[4], A[2]);
    const double k02 = fe_mulsub(A[1], A[4], b11, A[2]);
    const double det = fma(A[2], k02, fma(b00, k00, -(A[1] * k01))) / (p * p * p);
    const double half_det = fmin(fmax(det * 0.5, -1.0), 1.0);
    const double angle = acos(half_det) / 3.0;
    const double beta2 = cos(angle) * 2.0;
    const double beta0 = cos(angle + 2.09439510239319549) * 2.0;
    const double beta1 = -(beta0 + beta2);
    const double e0 = fma(p, beta0, q), e1 = fma(p, beta1, q), e2 = fma(p, beta2, q);
    double va[3], vb[3];
    if (half_det >= 0.0) {
        fe_eigenvector0(A, e2, va);                                    // evec2
        if (e2 < e0 && e2 < e1) { n[0] = va[0]; n[1] = va[1]; n[2] = va[2]; return; }
        fe_eigenvector1(A, va, e1, vb);                                // evec1
        if (e1 < e0 && e1 < e2) { n[0] = vb[0]; n[1] = vb[1]; n[2] = vb[2]; return; }
        fe_cross(vb, va, n);                                           // evec0 = evec1 x evec2
    } else {
        fe_eigenvector0(A, e0, va);                                    // evec0
        if (e0 < e1 && e0 < e2) { n[0] = va[0]; n[1] = va[1]; n[2] = va[2]; return; }
        fe_eigenvector1(A, va, e1, vb);                                // evec1
        if (e1 < e0 && e1 < e2) { n[0] = vb[0]; n[1] = vb[1]; n[2] = vb[2]; return; }
        fe_cross(va, vb, n);                                           // evec2 = evec0 x evec1
    }
}

// k_normals: one thread per (cell-sorted) point: hybrid / kNN search, population covariance, smallest eigenvector.
// < 3 neighbours -> (0,0,1).  If prev != NULL the sign follows the previous normal (legacy EstimateNormals).
__global__ void __launch_bounds__(KNN_BLOCK) k_normals(GridView g, int64_t n, int k, double radius, const double *__restrict__ prev,
                                                       double *__restrict__ normals, int *__restrict__ nn_count) {
    extern __shared__ double lds_d[];
    double *sd = lds_d;
    int *si = (int *)(lds_d + (size_t)k * KNN_BLOCK);
    int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int t = threadIdx.x;
    const double qx = g.pts[i * 3], qy = g.pts[i * 3 + 1], qz = g.pts[i * 3 + 2];
    const int cnt = knn_query(g, qx, qy, qz, k, radius, sd, si);
    // utility::ComputeCovariance [recalled]: ONE pass of nine cumulants over the RAW coordinates in neighbour order (nearest
    // first), products accumulated with fused multiply-adds, divided by the count, cov = E[ab] - E[a] E[b] (fused); fewer than
    // 3 neighbours: identity covariance, whose "normal" is (0,0,1).  Pinned by the recorded frames (oracle/normals.c).
    double nrm[3] = {0.0, 0.0, 0.0};
    if (cnt >= 3) {
        double sx = 0, sy = 0, sz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (int j0 = 0; j0 < cnt; j0 += 4) {       // four neighbours per round trip, summed in list order
            double X[4], Y[4], Z[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t s = si[min(j0 + u, cnt - 1) * KNN_BLOCK + t];
                X[u] = g.pts[s * 3]; Y[u] = g.pts[s * 3 + 1]; Z[u] = g.pts[s * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j0 + u >= cnt) break;
                const double x = X[u], y = Y[u], z = Z[u];
                sx += x; sy += y; sz += z;
                xx = fma(x, x, xx); xy = fma(x, y, xy); xz = fma(x, z, xz); yy = fma(y, y, yy); yz = fma(y, z, yz); zz = fma(z, z, zz);
            }
        }
        const double dn = (double)cnt;
        sx /= dn; sy /= dn; sz /= dn; xx /= dn; xy /= dn; xz /= dn; yy /= dn; yz /= dn; zz /= dn;
        fast_eigen3x3(fma(-sx, sx, xx), fma(-sx, sy, xy), fma(-sx, sz, xz), fma(-sy, sy, yy), fma(-sy, sz, yz), fma(-sz, sz, zz), nrm);
    }
    const int64_t o = g.idx[i];
    const bool zero = nrm[0] == 0.0 && nrm[1] == 0.0 && nrm[2] == 0.0;     // < 3 neighbours or an all-zero covariance
    if (prev) {   // a cloud that already carries normals: a zero result keeps the old normal, every other one is turned towards it
        const double px = prev[o * 3], py = prev[o * 3 + 1], pz = prev[o * 3 + 2];
        if (zero) { nrm[0] = px; nrm[1] = py; nrm[2] = pz; }
        if ((nrm[0] * px + nrm[1] * py) + nrm[2] * pz < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    } else if (zero) nrm[2] = 1.0;
    normals[o * 3] = nrm[0]; normals[o * 3 + 1] = nrm[1]; normals[o * 3 + 2] = nrm[2];
    if (nn_count) nn_count[o] = cnt;
}

// k_knn_dist: mean of the k smallest distances (self included) -- statistical outlier removal's per-point score;
// with count_radius > 0 instead counts the neighbours within that radius (radius outlier removal)
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_score(GridView g, int64_t n, int k, double count_radius, double *__restrict__ score) {
    extern __shared__ double lds_d[];
    double *sd = lds_d;
    int *si = (int *)(lds_d + (size_t)k * KNN_BLOCK);
    int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int t = threadIdx.x;
    const double qx = g.pts[i * 3], qy = g.pts[i * 3 + 1], qz = g.pts[i * 3 + 2];
    if (count_radius > 0) {
        const double r2 = count_radius * count_radius;
        const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
        int c = 0;
        const int smax = (int)ceil(count_radius * g.inv_cell);
        auto count = [&](int b, int e) {
            for (int j = b; j < e; j++) {
                double dx = g.pts[(int64_t)j * 3] - qx, dy = g.pts[(int64_t)j * 3 + 1] - qy, dz = g.pts[(int64_t)j * 3 + 2] - qz;
                if (dx * dx + dy * dy + dz * dz <= r2) c++;
            }
        };
        for_block3(g, cx, cy, cz, count);
        for (int s = 2; s <= smax; s++) for_shell(g, cx, cy, cz, s, count);
        score[g.idx[i]] = (double)c;
        return;
    }
    const int cnt = knn_query(g, qx, qy, qz, k, -1.0, sd, si);
    double a = 0;
    for (int j = 0; j < cnt; j++) a += sqrt(sd[j * KNN_BLOCK + t]);
    score[g.idx[i]] = cnt ? a / cnt : 0.0;
}

// k_knn_graph: indices (original numbering) and squared distances of the k nearest points, nearest first
// (the point itself comes first); rows shorter than k are padded with -1 / +inf
__global__ void __launch_bounds__(KNN_BLOCK) k_knn_graph(GridView g, int64_t n, int k, double radius, int *__restrict__ nbr,
                                                         double *__restrict__ d2) {
    extern __shared__ double lds_d[];
    double *sd = lds_d;
    int *si = (int *)(lds_d + (size_t)k * KNN_BLOCK);
    int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int t = threadIdx.x;
    const int cnt = knn_query(g, g.pts[i * 3], g.pts[i * 3 + 1], g.pts[i * 3 + 2], k, radius, sd, si);
    const int64_t o = g.idx[i];
    for (int j = 0; j < k; j++) {
        nbr[o * k + j] = j < cnt ? g.idx[si[j * KNN_BLOCK + t]] : -1;
        if (d2) d2[o * k + j] = j < cnt ? sd[j * KNN_BLOCK + t] : 1e300;
    }
}

// ------------------------------------------------------------------------------------------------ ICP
// One launch = one "GetRegistrationResultAndCorrespondences" + the sums the NEXT ComputeTransformation needs:
// for every source point p = T s: nearest target point within max_dist (strict <), then
//   slots [0] n  [1] sum d^2
//   P2P    : [2..4] sum p, [5..7] sum t, [8..16] sum p t^T (row-major p_i t_j)
//   P2PLANE: [2..22] upper triangle of J^T J (J = [p x n_t ; n_t]), [23..28] J^T r, r = (p - t).n_t
//   GICP   : same slots with J^T J = Jb^T M^-1 Jb, J^T r = Jb^T M^-1 d, Jb = [-[p]x | I], d = p - t,
//            M = C_t + R C_s R^T with C = I - (1-eps) n n^T   (== W^T W with W = M^-1/2 of the original)
constexpr int ICP_SLOTS = 29;
constexpr int ICP_BLOCK = 256;
enum { MODE_P2P = 0, MODE_P2PLANE = 1, MODE_GICP = 2 };

struct Rigid { double r[9], t[3]; };

// Device-resident state of one registration loop: the evaluation kernels read the pose from here and k_icp_step advances it,
// so consecutive iterations are enqueued back to back without a host round trip (the host looks at `done` once per batch).
struct IcpState {
    double T[16];             // pose used by the next evaluation (row-major 4x4)
    double fit, rmse, corr;   // statistics of the latest evaluation
    int evals;                // evaluations consumed so far
    int done, converged, iterations;
    // device clock (wall_clock64, 100 MHz) at the first and the latest step: the GPU-side duration of the loop, to tell a stall
    // of the device from a completion that reached the host late (R3D_ICP_DEBUG=1 prints both when they disagree)
    unsigned long long t_first, t_last;
    int seq;                  // host mirror only: evaluations published so far (written LAST, after a system-scope fence)
};
constexpr int ICP_SEQ_DONE = 1 << 30;   // bit of the mirror's `seq` word: the loop has ended (published with the sequence number)
__device__ __forceinline__ Rigid load_rigid(const IcpState *__restrict__ st) {
    Rigid T;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) T.r[i * 3 + j] = st->T[i * 4 + j];
        T.t[i] = st->T[i * 4 + 3];
    }
    return T;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ bool inv3_sym(const double m[6] /*xx xy xz yy yz zz*/, double a[6]) {
    const double c00 = m[3] * m[5] - m[4] * m[4], c01 = m[2] * m[4] - m[1] * m[5], c02 = m[1] * m[4] - m[2] * m[3];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (fabs(det) < 1e-300) return false;
    const double id = 1.0 / det;
    a[0] = c00 * id; a[1] = c01 * id; a[2] = c02 * id;
    a[3] = (m[0] * m[5] - m[2] * m[2]) * id; a[4] = (m[1] * m[2] - m[0] * m[4]) * id; a[5] = (m[0] * m[3] - m[1] * m[1]) * id;
    return true;
}

// nearest target point inside the 3x3x3 cell block around the query (shells 0 and 1), straight from global memory, as nine
// contiguous runs.  All 18 run bounds are requested before any is used, and the candidates of a run are fetched four at a
// time (clamped index, no branch around a load): the search is a chain of dependent gathers otherwise.
__device__ __forceinline__ void nn_block_global(const GridView &g, double px, double py, double pz, int cx, int cy, int cz, double &best,
                                                int &bi) {
    const int xa = cx - 1, xb = cx + 1;
    const bool xok = xb >= 0 && xa <= g.nx - 1;
    const int x0 = min(max(xa, 0), g.nx - 1), x1 = min(max(xb, 0), g.nx - 1);
    int rb[9], re[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool ok = xok && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
        const int64_t row = ((int64_t)min(max(z, 0), g.nz - 1) * g.ny + min(max(y, 0), g.ny - 1)) * g.nx;
        // both bounds of the run with ONE 16-byte request (the end entry is at most three behind the start entry; the table
        // is allocated with four spare entries); the search is bound by the number of cache-line requests, not by latency
        const cs_int4 cs = cs4(g, row + x0);
        const int b = cs.x, de = x1 + 1 - x0, e = de == 3 ? cs.w : de == 2 ? cs.z : cs.y;
        rb[r] = b;
        re[r] = b + ((e - b) & (ok ? -1 : 0));
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int e = re[r];
        for (int j0 = rb[r]; j0 < e; j0 += 4) {
            double X[4], Y[4], Z[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t jj = min(j0 + u, e - 1);
                X[u] = g.pts[jj * 3]; Y[u] = g.pts[jj * 3 + 1]; Z[u] = g.pts[jj * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                const double dx = X[u] - px, dy = Y[u] - py, dz = Z[u] - pz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (j < e && d2 <= best) {  // ties are rare: the index loads of the total order (d2, index) stay off the hot path
                    if (d2 < best || (bi >= 0 && g.idx[j] < g.idx[bi])) { best = d2; bi = j; }
                }
            }
        }
    }
}

// Two-stage form of the same search.  The loop is bound by the texture-addresser / L1 path (PMC: 4.4 M gather instructions
// per iteration at 1 M points, the L2 sees only 4.6 M requests), and a float64 candidate costs two gather instructions
// (24 bytes = dwordx4 + dwordx2).  Stage 1 scans float32 structure-of-arrays copies (three 16-byte gathers per FOUR
// candidates) and keeps the four smallest float distances with their slots -- branch-free, so no lane waits for another
// lane's exact arithmetic.  Stage 2 evaluates the best three exactly (float64, the update rule and tie order of
// nn_block_global), every lane doing the same three steps.  With |d_float - d_exact| <= fe/2 for every candidate (coordinate
// rounding of target and query; bound from the bounding box, computed on the host) every candidate that can be the exact
// nearest point, or tie with it, has a float distance within fe of the smallest float distance; if the FOURTH smallest is
// within that margin too, the three may not contain them all and the lane falls back to the all-float64 scan (rare: four
// candidates within micrometres of the same distance).  Result: identical to nn_block_global.
__device__ __forceinline__ void nn_block_top4(const GridView &g, double px, double py, double pz, int cx, int cy, int cz, double &best,
                                              int &bi) {
    const int xa = cx - 1, xb = cx + 1;
    const bool xok = xb >= 0 && xa <= g.nx - 1;
    const int x0 = min(max(xa, 0), g.nx - 1), x1 = min(max(xb, 0), g.nx - 1);
    typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    int rb[9], re[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool ok = xok && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
        const int64_t row = ((int64_t)min(max(z, 0), g.nz - 1) * g.ny + min(max(y, 0), g.ny - 1)) * g.nx;
        const cs_int4 cs = cs4(g, row + x0);
        const int b = cs.x, de = x1 + 1 - x0, e = de == 3 ? cs.w : de == 2 ? cs.z : cs.y;
        rb[r] = b;
        re[r] = b + ((e - b) & (ok ? -1 : 0));
    }
    const float qx = (float)px, qy = (float)py, qz = (float)pz;
    const float INF = __builtin_huge_valf();
    float t1 = INF, t2 = INF, t3 = INF, t4 = INF;
    int i1 = -1, i2 = -1, i3 = -1;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int e = re[r];
        for (int j0 = rb[r]; j0 < e; j0 += 4) {
            const float4_a4 X = *(const float4_a4 *)(g.fx + j0), Y = *(const float4_a4 *)(g.fy + j0), Z = *(const float4_a4 *)(g.fz + j0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                const float fdx = X[u] - qx, fdy = Y[u] - qy, fdz = Z[u] - qz;
                float f = fdx * fdx + fdy * fdy + fdz * fdz;
                f = j < e ? f : INF;
                // insert (f, j) into the ascending quadruple; the slot of the fourth is never needed
                const bool c3 = f < t3, c2 = f < t2, c1 = f < t1;
                t4 = c3 ? t3 : fminf(t4, f);
                i3 = c2 ? i2 : (c3 ? j : i3);
                t3 = c2 ? t2 : (c3 ? f : t3);
                i2 = c1 ? i1 : (c2 ? j : i2);
                t2 = c1 ? t1 : (c2 ? f : t2);
                i1 = c1 ? j : i1;
                t1 = c1 ? f : t1;
            }
        }
    }
    if (!(t1 < INF)) return;                       // empty block: nothing to evaluate (the caller goes on to the outer shells)
    const float tt = sqrtf(t1) * 1.000001f + g.fe;  // the relative term covers the float32 evaluation of the distance itself
    const float thr2 = tt * tt * 1.000001f;
    if (t4 <= thr2) {                              // a fourth contender: the triple may be incomplete
        nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
        return;
    }
    const int cand[3] = {i1, i2, i3};
    const float ct[3] = {t1, t2, t3};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int j = max(cand[q], 0);             // clamped slot: no branch around the loads
        const double dx = g.pts[(int64_t)j * 3] - px, dy = g.pts[(int64_t)j * 3 + 1] - py, dz = g.pts[(int64_t)j * 3 + 2] - pz;
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (cand[q] >= 0 && ct[q] <= thr2 && d2 <= best) {
            if (d2 < best || (bi >= 0 && g.idx[j] < g.idx[bi])) { best = d2; bi = j; }
        }
    }
}

// reach bitmap of a search grid (GridView::reach): coarse occupancy, then a 3-tap maximum along x, y and z
__global__ void __launch_bounds__(256) k_reach_mark(GridView g, int64_t n, int rs, int rnx, int rny, int rnz, unsigned char *__restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cx = min(max(cell_coord(g.pts[i * 3], g.ox, g.inv_cell), 0), g.nx - 1) / rs;
    const int cy = min(max(cell_coord(g.pts[i * 3 + 1], g.oy, g.inv_cell), 0), g.ny - 1) / rs;
    const int cz = min(max(cell_coord(g.pts[i * 3 + 2], g.oz, g.inv_cell), 0), g.nz - 1) / rs;
    occ[((int64_t)cz * rny + cy) * rnx + cx] = 1;
}
__global__ void __launch_bounds__(256) k_reach_dilate(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int64_t n, int64_t stride, int dim) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t pos = (i / stride) % dim;            // coordinate along the filtered axis
    unsigned char v = in[i];
    if (pos > 0) v |= in[i - stride];
    if (pos + 1 < dim) v |= in[i + stride];
    out[i] = v;
}

__global__ void __launch_bounds__(256) k_soa_f32(const double *__restrict__ pts, int64_t n, float *__restrict__ fx, float *__restrict__ fy,
                                                 float *__restrict__ fz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n + 4) return;
    const bool in = i < n;
    const int64_t ic = in ? i : n - 1;
    fx[i] = in ? (float)pts[ic * 3] : 0.f;
    fy[i] = in ? (float)pts[ic * 3 + 1] : 0.f;
    fz[i] = in ? (float)pts[ic * 3 + 2] : 0.f;
}

// Packed form of the two-stage search (the default).  PMC on the float32 form: 51 M VALU wave-instructions per evaluation
// at 1 M points (3 300 per 64 queries) against 0.5 M gathers -- the search is bound by the vector ALU, not by memory: 37
// instructions per candidate (most of them the sorted insertion of (distance, slot) into a quadruple), every one of the nine
// rows walked by the whole wave for as long as its longest lane.  Here
//  (1) a candidate is ONE dword: its position inside its own cell quantised to 10 bits per axis, x together with the low two
//      bits of its cell x index (k_pack_q10): one 16-byte gather brings four candidates, and the query is expressed in the
//      same unit (cell / 1024) relative to the corner of its 3x3x3 block, where every coordinate is below 4096 and float32
//      is exact to 2^-11 units;
//  (2) distance and candidate travel as ONE 32-bit key: the float bits of the squared distance with the low 10 mantissa
//      bits replaced by the candidate's ordinal in this query's scan, so the sorted insertion into the best four is
//      v_min + 3 x v_med3 (20 instructions per candidate in all);
//  (3) the centre row is scanned first; a row whose slab distance to the query exceeds the margin threshold of the best
//      distance so far (or the correspondence radius) is dropped: it cannot hold the nearest point, a tie with it, or a
//      contender that the fallback test would have to count; the surviving rows go to a per-lane list (LDS, column per
//      thread) that is walked in ONE loop, so a wave iterates max-over-lanes of the per-lane total instead of the sum over
//      rows of the per-row maximum.
// Error budget of stage 1 in units: reconstruction of a candidate 0.5 per axis (0.867 in distance), the query's float
// coordinates 2.5e-4, key truncation 2^-13 relative in d^2 (<= 0.22 at the far corner of the block): |d_stage1 - d_exact| <=
// 1.1, FEQ ("fe" of nn_block_top4) = 2.25 units = cell / 455.  Stage 2 (exact float64 evaluation of the best three, fallback
// to nn_block_global on a fourth contender within the margin) is unchanged, and so is the result: bit-identical to the
// all-float64 search (tests/test_cloud_gpu.py).  Queries whose nine rows hold more than 1024 candidates use the fallback.
constexpr float FEQ = 2.25f;
// pooled walk (nn_block_q10<1>): queue capacity per wave, and words per wave of the pool buffer: keys [Q][4], items [Q][2], the
// lanes' query parameters [4][64], the queue length
constexpr int POOL_Q = 256, POOL_Q_SEARCH = 176 /* k_icp_search: four workgroups' LDS still fit a CU */;
constexpr int pool_words(int q) { return 6 * q + 256 + 4; }

__global__ void __launch_bounds__(256) k_pack_q10(GridView g, int64_t n, unsigned *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n + 4) return;
    unsigned w = 0;
    if (i < n) {
        const double u[3] = {(g.pts[i * 3] - g.ox) * g.inv_cell, (g.pts[i * 3 + 1] - g.oy) * g.inv_cell, (g.pts[i * 3 + 2] - g.oz) * g.inv_cell};
        const int dims[3] = {g.nx, g.ny, g.nz};
        unsigned q[3];
        int c0 = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int c = min(max((int)floor(u[a]), 0), dims[a] - 1);   // the cell the point was sorted into (k_cell_keys)
            q[a] = (unsigned)min(max((int)floor((u[a] - c) * 1024.0), 0), 1023);
            if (a == 0) c0 = c;
        }
        w = q[0] | (unsigned)(c0 & 3) << 10 | q[1] << 12 | q[2] << 22;
    }
    out[i] = w;
}

__device__ __forceinline__ int med3_i32(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// G: groups of four candidates requested together before any of them is used (1: one gather per step of the scan, the form for
// large clouds, where three waves per SIMD hide the gathers; 3: for clouds so small that a SIMD holds one wave -- a scanning-loop
// frame of 40 k points is 160 workgroups on 256 CUs -- and every dependent gather is a full memory round trip: the same candidates
// in the same order, so the same keys and the same result)
#ifdef R3D_ICP_STATS   // diagnostic build only (csrc/build.sh -DR3D_ICP_STATS, tools/gpu_icp_stats.py): trip counts of the packed search per
                       // wave, as the maximum over lanes (what the wave executes) and the sum over lanes (what its lanes need)
__device__ unsigned long long g_icp_stats[16];
__device__ __forceinline__ void icp_stat(int slot, int v) {
    int m = v, sum = v;
    for (int o = 32; o > 0; o >>= 1) { m = max(m, __shfl_xor(m, o)); sum += __shfl_xor(sum, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_icp_stats[slot], (unsigned long long)m); atomicAdd(&g_icp_stats[slot + 1], (unsigned long long)sum); }
}
#endif
template <int G, int PQ = POOL_Q>
__device__ __forceinline__ void nn_block_q10(const GridView &g, double px, double py, double pz, int cx, int cy, int cz, double r2,
                                             double &best, int &bi, int *__restrict__ sRun /* [18][ICP_BLOCK] */,
                                             int *__restrict__ sPool = nullptr /* G == 1: [ICP_BLOCK / 64][pool_words(PQ)], 16-byte aligned */) {
    constexpr int B = 256;   // = ICP_BLOCK (declared below)
    const int tid = threadIdx.x;
    const int xa = cx - 1, xb = cx + 1;
    const bool xok = xb >= 0 && xa <= g.nx - 1;
    const int x0 = min(max(xa, 0), g.nx - 1), x1 = min(max(xb, 0), g.nx - 1);
    typedef unsigned uint4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int ORD[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};   // centre row, the four rows sharing a face with it, the four corners
    int rb[9], re[9], rm1[9], rm2[9], total = 0;   // run bounds and the two cell boundaries inside a full three-cell run
    // table index of the run's first cell: the table has at most 2^26 entries (grid_build), so 32-bit arithmetic, two vector
    // multiplies for all nine rows (the row offsets are wave-uniform); rows outside the grid read entry 0 and are emptied
    const unsigned base = ((unsigned)(cz - 1) * (unsigned)g.ny + (unsigned)(cy - 1)) * (unsigned)g.nx + (unsigned)x0;
    const int de = x1 + 1 - x0;
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int r = ORD[q], rz = r / 3, ry = r % 3;
        const bool ok = xok && (unsigned)(cz + rz - 1) < (unsigned)g.nz && (unsigned)(cy + ry - 1) < (unsigned)g.ny;
        const unsigned off = ((unsigned)rz * (unsigned)g.ny + (unsigned)ry) * (unsigned)g.nx;
        const cs_int4 cs = cs4(g, (int64_t)(ok ? base + off : 0u));
        const int b = cs.x, e = de == 3 ? cs.w : de == 2 ? cs.z : cs.y;
        rb[q] = b;
        re[q] = b + ((e - b) & (ok ? -1 : 0));
        rm1[q] = cs.y;
        rm2[q] = cs.z;
        total += (re[q] - rb[q] + 3) & ~3;
    }
    if (total > 1024) {                            // the ordinal field of the keys would overflow (very dense cells)
        nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
        return;
    }
    // the query in units of cell / 1024 relative to the block corner (cx-1, cy-1, cz-1): [1024, 2048) on every axis
    const double ux = (px - g.ox) * g.inv_cell, uy = (py - g.oy) * g.inv_cell, uz = (pz - g.oz) * g.inv_cell;
    const float qxc = (float)((ux - (double)(cx - 1)) * 1024.0), qyc = (float)((uy - (double)(cy - 1)) * 1024.0),
                qzc = (float)((uz - (double)(cz - 1)) * 1024.0);
    const float qx = qxc - 0.5f;                   // candidates are reconstructed at the centre of their quantisation step
    const unsigned xsub = (unsigned)(cx - 1) << 10;
    constexpr int KINF = 0x7f800000;
    auto thr_of = [&](float t) { const float tt = sqrtf(t) * 1.000001f + FEQ; return tt * tt * 1.000001f; };
    const float thr_r = thr_of((float)(r2 * g.inv_cell * g.inv_cell * (1024.0 * 1024.0)) * 1.000001f);
    int k1 = KINF, k2 = KINF, k3 = KINF, k4 = KINF;   // the four smallest keys, ascending
    int ord = 0;
    // four candidates per 16-byte gather; the distance arithmetic runs two candidates per instruction (v_pk_add / v_pk_mul /
    // v_pk_fma_f32: 6 packed instructions per pair instead of 12 scalar ones; lane-wise IEEE, so the keys are unchanged)
    typedef float v2f __attribute__((ext_vector_type(2)));
    // keys of four packed candidates for a query given by (qx_, xsub_, qyr, qzr), ordinals ord_ ..; the lanes of a wave also run this
    // for EACH OTHER's queries (pooled walk below): same expressions, same operand values, so the same keys whoever computes them
    auto keys4 = [&](const uint4_a4 W, int rem, float qx_, unsigned xsub_, float qyr, float qzr, int ord_, int (&keys)[4]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned w0 = W[2 * h], w1 = W[2 * h + 1];
            v2f cx2 = {(float)((w0 - xsub_) & 4095u), (float)((w1 - xsub_) & 4095u)};
            v2f cy2 = {__uint_as_float((w0 & 0x003ff000u) | 0x45000000u), __uint_as_float((w1 & 0x003ff000u) | 0x45000000u)};
            v2f cz2 = {(float)(w0 >> 22), (float)(w1 >> 22)};
            const v2f qx2 = {qx_, qx_}, qy2 = {qyr + 2048.f, qyr + 2048.f}, qz2 = {qzr, qzr};
            const v2f fdx = cx2 - qx2, fdy = cy2 - qy2, fdz = cz2 - qz2;
            const v2f f = __builtin_elementwise_fma(fdz, fdz, __builtin_elementwise_fma(fdy, fdy, fdx * fdx));
            keys[2 * h] = (__float_as_int(f.x) & ~1023) | (ord_ + 2 * h);
            keys[2 * h + 1] = (__float_as_int(f.y) & ~1023) | (ord_ + 2 * h + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) keys[u] = u < rem ? keys[u] : KINF;
    };
    auto fold4 = [&](const int (&keys)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int key = keys[u];
            k4 = med3_i32(k3, k4, key);
            k3 = med3_i32(k2, k3, key);
            k2 = med3_i32(k1, k2, key);
            k1 = min(k1, key);
        }
    };
    auto scan4w = [&](const uint4_a4 W, int rem, float qyr, float qzr) {
        int keys[4];
        keys4(W, rem, qx, xsub, qyr, qzr, ord, keys);
        fold4(keys);
        ord += 4;
    };
    auto scan4 = [&](int j0, int e, float qyr, float qzr) { scan4w(*(const uint4_a4 *)(g.q10 + j0), e - j0, qyr, qzr); };
    // centre row first: it gives the pruning threshold for the other eight
#ifdef R3D_ICP_STATS
    {
        int lg = 0, lc = 0;
        for (int q = 1; q < 9; q++) { lg += (re[q] - rb[q] + 3) >> 2; lc += re[q] - rb[q]; }
        icp_stat(0, (re[0] - rb[0] + 3) >> 2);      // centre-row groups
        icp_stat(2, re[0] - rb[0]);                 // centre-row candidates
        icp_stat(4, lg);                            // groups of the other eight rows before pruning
        icp_stat(6, lc);
    }
#endif
    if (G == 1) {
        for (int j0 = rb[0]; j0 < re[0]; j0 += 4) scan4(j0, re[0], qyc - 1024.5f, qzc - 1024.5f);
    } else {
        for (int j0 = rb[0]; j0 < re[0]; j0 += 4 * G) {
            uint4_a4 W[G];
#pragma unroll
            for (int q = 0; q < G; q++) W[q] = *(const uint4_a4 *)(g.q10 + min(j0 + 4 * q, re[0] - 1));   // clamped: a group past the run is not used
#pragma unroll
            for (int q = 0; q < G; q++)
                if (j0 + 4 * q < re[0]) scan4w(W[q], re[0] - (j0 + 4 * q), qyc - 1024.5f, qzc - 1024.5f);
        }
    }
    const float thr = fminf(thr_r, thr_of(__int_as_float(k1 & ~1023)) * 1.0003f);   // (key truncation: the key is below f by < 2^-13)
    const bool trim = de == 3 && xa == x0;         // the run is exactly the cells cx-1, cx, cx+1 (no clamping at the grid border)
    const float sxl = (qxc - 1024.f) * (qxc - 1024.f), sxh = (2048.f - qxc) * (2048.f - qxc);
    // a record is two words: first slot, and length | row << 20 (lengths are below 1024); ordinal bases are the running sum of
    // the padded lengths, recomputed by the one decode at the end.  Record 0: the centre row (ordinal 0)
    sRun[tid] = rb[0];
    sRun[B + tid] = re[0] - rb[0];
    int nr = 1, myg = 0;                            // records, and groups of four candidates in the records after the centre row
#pragma unroll
    for (int q = 1; q < 9; q++) {
        const int r = ORD[q], ry = r % 3, rz = r / 3;
        const float sy = ry == 0 ? qyc - 1024.f : ry == 2 ? 2048.f - qyc : 0.f, sz = rz == 0 ? qzc - 1024.f : rz == 2 ? 2048.f - qzc : 0.f;
        const float syz = sy * sy + sz * sz;
        // a full run (cells cx-1, cx, cx+1) also drops an end cell that is too far in x
        const int b_ = trim && sxl + syz > thr ? rm1[q] : rb[q], e_ = trim && sxh + syz > thr ? rm2[q] : re[q];
        if (re[q] > rb[q] && e_ > b_ && syz <= thr) {   // (a row outside the grid has re == rb and meaningless cell boundaries)
            sRun[(nr * 2) * B + tid] = b_;
            sRun[(nr * 2 + 1) * B + tid] = (e_ - b_) | (r << 20);
            myg += (e_ - b_ + 3) >> 2;
            nr++;
        }
    }
#ifdef R3D_ICP_STATS
    {
        int lg = 0, lc = 0;
        for (int k = 1; k < nr; k++) { const int len = sRun[(k * 2 + 1) * B + tid] & 0xfffff; lg += (len + 3) >> 2; lc += len; }
        icp_stat(8, lg);                            // groups of the surviving rows (after the slab test and the end-cell trim)
        icp_stat(10, lc);
        icp_stat(12, nr - 1);                       // surviving rows
        icp_stat(14, 1);                            // [14] waves, [15] queries
    }
#endif
    // POOLED WALK of the other rows (G == 1, large clouds).  Left to itself a wave walks its lists for as long as its longest lane:
    // 9.3 groups on average where its average lane has 2.3 (tools/gpu_icp_stats.py) -- three quarters of the wave's lane-steps
    // are idle.  Here the lanes put their groups into ONE queue of the wave (LDS), every lane computes the keys of every
    // (active-lane-count)-th group, whoever's query it belongs to -- the owner's query parameters travel through LDS, the
    // arithmetic is keys4() with the same operands, so the keys are the ones the owner would have computed -- and the owners
    // fold their groups' keys into their best four (4 instructions per candidate instead of 20).  A wave whose queue would
    // overflow takes the per-lane walk below.  R3D_ICP_POOL=0 (host side: no pool buffer is passed) keeps the per-lane walk.
    bool pooled = false;
    if (G == 1 && sPool) {
        int *pw = sPool + (tid >> 6) * pool_words(PQ);
        int *p_keys = pw, *p_item = pw + 4 * PQ, *p_q = pw + 6 * PQ, *p_tot = pw + 6 * PQ + 256;
        const int lane = tid & 63;
        *(volatile int *)p_tot = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int off = atomicAdd(p_tot, myg);                    // ds_add_rtn: a disjoint range of the queue per lane
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int tot = *(volatile int *)p_tot;                   // the same for every lane of the wave
        if (tot <= PQ) {
            pooled = true;
            p_q[lane] = __float_as_int(qx);
            p_q[64 + lane] = __float_as_int(qyc);
            p_q[128 + lane] = __float_as_int(qzc);
            p_q[192 + lane] = (int)xsub;
            int it = off, ob = ord;                               // ord: the centre row's padded length
            for (int k = 1; k < nr; k++) {
                const int jb = sRun[(k * 2) * B + tid], lr = sRun[(k * 2 + 1) * B + tid], len = lr & 0xfffff, r = lr >> 20;
                for (int o = 0; o < len; o += 4, it++) {
                    p_item[2 * it] = jb + o;
                    p_item[2 * it + 1] = lane | min(len - o, 4) << 6 | r << 9 | (ob + o) << 13;
                }
                ob += (len + 3) & ~3;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long act = __ballot(1);
            const int nact = __popcll(act), rank = __popcll(act & ((1ull << lane) - 1ull));
            for (int i = rank; i < tot; i += nact) {
                const int j0 = p_item[2 * i], meta = p_item[2 * i + 1];
                const int o = meta & 63, rem = (meta >> 6) & 7, r = (meta >> 9) & 15, ordb = (int)((unsigned)meta >> 13);
                const float oqx = __int_as_float(p_q[o]), oqy = __int_as_float(p_q[64 + o]), oqz = __int_as_float(p_q[128 + o]);
                const unsigned oxs = (unsigned)p_q[192 + o];
                const int rz = (r * 11) >> 5, ry = r - 3 * rz;
                int keys[4];
                keys4(*(const uint4_a4 *)(g.q10 + j0), rem, oqx, oxs, oqy - 0.5f - 1024.f * (float)ry, oqz - 0.5f - 1024.f * (float)rz, ordb, keys);
                *(int4 *)(p_keys + 4 * i) = make_int4(keys[0], keys[1], keys[2], keys[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int gq = 0; gq < myg; gq++) {
                const int4 kk = *(const int4 *)(p_keys + 4 * (off + gq));
                const int keys[4] = {kk.x, kk.y, kk.z, kk.w};
                fold4(keys);
            }
        }
    }
    if (!pooled) {
        int k = 1, j0 = 0, e = 0;
        float qyr = 0.f, qzr = 0.f;
        auto next_run = [&]() {     // false: the list is exhausted
            if (k >= nr) return false;
            j0 = sRun[(k * 2) * B + tid];
            const int lr = sRun[(k * 2 + 1) * B + tid];
            e = j0 + (lr & 0xfffff);
            const int r = lr >> 20;
            const int rz = (r * 11) >> 5, ry = r - 3 * rz;
            qyr = qyc - 0.5f - 1024.f * (float)ry;
            qzr = qzc - 0.5f - 1024.f * (float)rz;
            k++;
            return true;
        };
        if (G == 1) {
            for (;;) {
                if (j0 >= e && !next_run()) break;
                scan4(j0, e, qyr, qzr);
                j0 += 4;
            }
        } else {
            // walk the list G groups ahead (addresses and row constants only), request them all, then use them in the same order
            for (bool more = true; more;) {
                int aj[G], ar[G];
                float ay[G], az[G];
                int ng = 0;
#pragma unroll
                for (int q = 0; q < G; q++) {
                    aj[q] = 0; ar[q] = 0; ay[q] = 0.f; az[q] = 0.f;
                    if (more && j0 >= e) more = next_run();
                    if (more) { aj[q] = j0; ar[q] = e - j0; ay[q] = qyr; az[q] = qzr; j0 += 4; ng = q + 1; }
                }
                uint4_a4 W[G];
#pragma unroll
                for (int q = 0; q < G; q++) W[q] = *(const uint4_a4 *)(g.q10 + aj[q]);     // unused slots read entry 0
#pragma unroll
                for (int q = 0; q < G; q++)
                    if (q < ng) scan4w(W[q], ar[q], ay[q], az[q]);
            }
        }
    }
    if (k1 == KINF) return;                        // nothing scanned: the caller goes on to the outer shells
    const float t1 = __int_as_float(k1 | 1023);    // upper end of the key's truncation interval
    const float thr2 = thr_of(t1) * 1.0003f;
    if (__int_as_float(k4 & ~1023) <= thr2) {      // a fourth contender: the triple may be incomplete
        nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
        return;
    }
    // ordinal -> slot: the record with the largest ordinal base not above it.  The best candidate is always evaluated; the
    // second and third only when they lie inside the margin (about one query in a hundred), in a branch of their own
    auto slot_of = [&](int o) {
        int jb = sRun[tid], ob = 0, next = ((sRun[B + tid] & 0xfffff) + 3) & ~3;
        for (int k = 1; k < nr; k++) {
            const int b_ = sRun[(k * 2) * B + tid], o_ = next;
            next += ((sRun[(k * 2 + 1) * B + tid] & 0xfffff) + 3) & ~3;
            if (o >= o_) { jb = b_; ob = o_; }
        }
        return jb + (o - ob);
    };
    auto exact = [&](int j) {
        const double dx = g.pts[(int64_t)j * 3] - px, dy = g.pts[(int64_t)j * 3 + 1] - py, dz = g.pts[(int64_t)j * 3 + 2] - pz;
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 <= best) {
            if (d2 < best || (bi >= 0 && g.idx[j] < g.idx[bi])) { best = d2; bi = j; }
        }
    };
    exact(slot_of(k1 & 1023));
    if (k2 != KINF && __int_as_float(k2 & ~1023) <= thr2) {
        exact(slot_of(k2 & 1023));
        if (k3 != KINF && __int_as_float(k3 & ~1023) <= thr2) exact(slot_of(k3 & 1023));
    }
}

// shells 2.. of the 1-NN search (only queries whose nearest point is further than one cell away get here: most of them in the
// first evaluation of an unaligned pair, a handful per million afterwards), by pruned x-rows (for_shell_rows); the result does
// not depend on the visiting order because of the total order (d2, index).
__device__ __forceinline__ void nn_outer_shells(const GridView &g, double px, double py, double pz, int cx, int cy, int cz, int smax,
                                                double &best, int &bi) {
    auto visit = [&](int b, int e) {
        for (int j0 = b; j0 < e; j0 += 4) {       // four candidates per round trip, clamped slots (no branch around a load)
            double X[4], Y[4], Z[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t jj = min(j0 + u, e - 1);
                X[u] = g.pts[jj * 3]; Y[u] = g.pts[jj * 3 + 1]; Z[u] = g.pts[jj * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                const double dx = X[u] - px, dy = Y[u] - py, dz = Z[u] - pz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (j < e && d2 <= best) {
                    if (d2 < best || (bi >= 0 && g.idx[j] < g.idx[bi])) { best = d2; bi = j; }
                }
            }
        }
    };
    for (int s = 1; s <= max(smax, 1); s++) {
        if (s > 1) for_shell_rows(g, px, py, pz, cx, cy, cz, s, [&]() { return best; }, visit);
        const double reach = s * g.cell;
        if (bi >= 0 && best <= reach * reach) break;
    }
}

// true when the reach bitmap proves that no target point lies within rs fine cells of the query's cell in any direction (so none
// within the correspondence radius).  Queries outside the grid are measured from the nearest grid cell: conservative.
__device__ __forceinline__ bool icp_out_of_reach(const GridView &g, int cx, int cy, int cz) {
    if (!g.reach) return false;
    if (cx < -g.rs || cy < -g.rs || cz < -g.rs || cx >= g.nx + g.rs || cy >= g.ny + g.rs || cz >= g.nz + g.rs) return true;   // beyond the grid by more than the radius
    const int qx = min(max(cx, 0), g.nx - 1) / g.rs, qy = min(max(cy, 0), g.ny - 1) / g.rs, qz = min(max(cz, 0), g.nz - 1) / g.rs;
    return g.reach[((int64_t)qz * g.rny + qy) * g.rnx + qx] == 0;
}

// adds one correspondence (source point i at p, target slot bi at squared distance `best`) to the accumulators
template <int MODE>
__device__ __forceinline__ void icp_accumulate(const GridView &g, const double *__restrict__ src_n, const double *__restrict__ tgt_n,
                                               int64_t i, const Rigid &T, double eps, double px, double py, double pz, double best, int bi,
                                               double (&acc)[29]) {
        const double tx = g.pts[(int64_t)bi * 3], ty = g.pts[(int64_t)bi * 3 + 1], tz = g.pts[(int64_t)bi * 3 + 2];
        acc[0] += 1.0;
        acc[1] += best;
        if (MODE == MODE_P2P) {
            acc[2] += px; acc[3] += py; acc[4] += pz;
            acc[5] += tx; acc[6] += ty; acc[7] += tz;
            acc[8] += px * tx; acc[9] += px * ty; acc[10] += px * tz;
            acc[11] += py * tx; acc[12] += py * ty; acc[13] += py * tz;
            acc[14] += pz * tx; acc[15] += pz * ty; acc[16] += pz * tz;
        } else {
            double J[3][6], rr[3];
            int rows;
            const double nx = tgt_n[(int64_t)bi * 3], ny = tgt_n[(int64_t)bi * 3 + 1], nz = tgt_n[(int64_t)bi * 3 + 2];
            const double dx = px - tx, dy = py - ty, dz = pz - tz;
            if (MODE == MODE_P2PLANE) {
                rows = 1;
                J[0][0] = py * nz - pz * ny; J[0][1] = pz * nx - px * nz; J[0][2] = px * ny - py * nx;
                J[0][3] = nx; J[0][4] = ny; J[0][5] = nz;
                rr[0] = dx * nx + dy * ny + dz * nz;
#pragma unroll
                for (int a = 0, q = 2; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++, q++) acc[q] += J[0][a] * J[0][b];
#pragma unroll
                for (int a = 0; a < 6; a++) acc[23 + a] += J[0][a] * rr[0];
            } else {
                rows = 3;
                (void)rows;
                // effective normals (QUIRK of GetRotationFromE1ToX: n.e1 < -0.99 -> e1)
                double ax = nx, ay = ny, az = nz;
                if (ax < -0.99) { ax = 1; ay = 0; az = 0; }
                double ux = src_n[i * 3], uy = src_n[i * 3 + 1], uz = src_n[i * 3 + 2];
                if (ux < -0.99) { ux = 1; uy = 0; uz = 0; }
                const double bx = T.r[0] * ux + T.r[1] * uy + T.r[2] * uz, by = T.r[3] * ux + T.r[4] * uy + T.r[5] * uz,
                             bz = T.r[6] * ux + T.r[7] * uy + T.r[8] * uz;
                const double w = 1.0 - eps;
                double M[6] = {2.0 - w * (ax * ax + bx * bx), -w * (ax * ay + bx * by), -w * (ax * az + bx * bz),
                               2.0 - w * (ay * ay + by * by), -w * (ay * az + by * bz), 2.0 - w * (az * az + bz * bz)};
                double A[6];
                if (!inv3_sym(M, A)) return;
                // Jb = [K | I], K = -[p]x ; G = A * Jb (3x6) ; JtJ = Jb^T G ; Jtr = Jb^T (A d).
                // Written out by blocks: K has one zero per row and column and the right half of Jb is the identity; spelled as
                // dense 3-term products those zeros and ones are real float64 multiplies (0 * x is not foldable under IEEE rules),
                // a third of this function's arithmetic.  Dropping an exact-zero term of a sum leaves the sum unchanged bit for bit:
                //   K = [0 pz -py; -pz 0 px; py -px 0],  AK[a][b] = sum_c A[a][c] K[c][b],  JtJ = [K^T A K, K^T A; A K, A]
                const double As[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
                double AK[3][3];
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    AK[a][0] = As[a][1] * -pz + As[a][2] * py;
                    AK[a][1] = As[a][0] * pz + As[a][2] * -px;
                    AK[a][2] = As[a][0] * -py + As[a][1] * px;
                }
                const double Ad[3] = {As[0][0] * dx + As[0][1] * dy + As[0][2] * dz, As[1][0] * dx + As[1][1] * dy + As[1][2] * dz,
                                      As[2][0] * dx + As[2][1] * dy + As[2][2] * dz};
                // rows of K^T X for a 3-vector of rows X[0..2]: (K^T X)[0] = -pz X[1] + py X[2], [1] = pz X[0] - px X[2], [2] = -py X[0] + px X[1]
                // slots: q = 2 + index of (a, b), a <= b, row-major over the upper triangle of the 6x6
                // a = 0: b = 0..5
                acc[2] += -pz * AK[1][0] + py * AK[2][0];
                acc[3] += -pz * AK[1][1] + py * AK[2][1];
                acc[4] += -pz * AK[1][2] + py * AK[2][2];
                acc[5] += -pz * As[1][0] + py * As[2][0];
                acc[6] += -pz * As[1][1] + py * As[2][1];
                acc[7] += -pz * As[1][2] + py * As[2][2];
                // a = 1: b = 1..5
                acc[8] += pz * AK[0][1] + -px * AK[2][1];
                acc[9] += pz * AK[0][2] + -px * AK[2][2];
                acc[10] += pz * As[0][0] + -px * As[2][0];
                acc[11] += pz * As[0][1] + -px * As[2][1];
                acc[12] += pz * As[0][2] + -px * As[2][2];
                // a = 2: b = 2..5
                acc[13] += -py * AK[0][2] + px * AK[1][2];
                acc[14] += -py * As[0][0] + px * As[1][0];
                acc[15] += -py * As[0][1] + px * As[1][1];
                acc[16] += -py * As[0][2] + px * As[1][2];
                // a = 3..5: the rows of A itself
                acc[17] += As[0][0]; acc[18] += As[0][1]; acc[19] += As[0][2];
                acc[20] += As[1][1]; acc[21] += As[1][2];
                acc[22] += As[2][2];
                acc[23] += -pz * Ad[1] + py * Ad[2];
                acc[24] += pz * Ad[0] + -px * Ad[2];
                acc[25] += -py * Ad[0] + px * Ad[1];
                acc[26] += Ad[0]; acc[27] += Ad[1]; acc[28] += Ad[2];
            }
        }
    
}

enum { SEARCH_EXACT = 0, SEARCH_F32 = 1, SEARCH_Q10 = 2, SEARCH_Q10_DEEP = 3 /* nn_block_q10<3>: small clouds */ };
template <int MODE, int SEARCH>
__global__ void __launch_bounds__(ICP_BLOCK, SEARCH == SEARCH_Q10_DEEP ? 2 : 3) k_icp_eval(GridView g, const double *__restrict__ src, const double *__restrict__ src_n,
                                                        const double *__restrict__ tgt_n /* cell-sorted order */, int64_t ns,
                                                        const IcpState *__restrict__ st, double max_dist, double eps, double *__restrict__ partial,
                                                        int *__restrict__ corr /* optional [ns] target original index or -1 */) {
    static_assert(ICP_BLOCK == 256, "nn_block_q10 assumes 256 threads");
    __shared__ int sRun[SEARCH >= SEARCH_Q10 ? 18 * ICP_BLOCK : 1];
    __shared__ __attribute__((aligned(16))) int sPoolBuf[SEARCH == SEARCH_Q10 ? (ICP_BLOCK / 64) * pool_words(POOL_Q) : 4];
    int *const sPool = SEARCH == SEARCH_Q10 && g.pool ? sPoolBuf : nullptr;
    if (st->done) return;                        // the loop ended in an earlier launch of this batch (uniform)
    const Rigid T = load_rigid(st);
    double acc[ICP_SLOTS];
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) acc[q] = 0;
    const double r2 = max_dist * max_dist;
    const int smax = (int)ceil(max_dist * g.inv_cell);
    // XCD-aware share of the queries: workgroups go round-robin to the eight XCDs, each with an L2 of its own; the source is in
    // Morton order of the target cells, so giving XCD x the x-th eighth of the chunks (256 consecutive queries) confines the
    // target data an L2 sees (cell table, packed candidates, points, normals) to an eighth of the cloud
    const int64_t nchunk = (ns + ICP_BLOCK - 1) / ICP_BLOCK;
    int64_t c_begin = blockIdx.x, c_end = nchunk, c_step = gridDim.x;
    if ((gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        c_begin = x * nchunk / 8 + (blockIdx.x >> 3);
        c_end = (x + 1) * nchunk / 8;
        c_step = gridDim.x >> 3;
    }
    for (int64_t c = c_begin; c < c_end; c += c_step) {
        const int64_t i = c * ICP_BLOCK + threadIdx.x;
        if (i >= ns) continue;
        const double sx = src[i * 3], sy = src[i * 3 + 1], sz = src[i * 3 + 2];
        const double px = T.r[0] * sx + T.r[1] * sy + T.r[2] * sz + T.t[0];
        const double py = T.r[3] * sx + T.r[4] * sy + T.r[5] * sz + T.t[1];
        const double pz = T.r[6] * sx + T.r[7] * sy + T.r[8] * sz + T.t[2];
        const int cx = cell_coord(px, g.ox, g.inv_cell), cy = cell_coord(py, g.oy, g.inv_cell), cz = cell_coord(pz, g.oz, g.inv_cell);
        double best = r2;
        int bi = -1;
        if (!icp_out_of_reach(g, cx, cy, cz)) {
            if (SEARCH == SEARCH_Q10_DEEP) nn_block_q10<3>(g, px, py, pz, cx, cy, cz, r2, best, bi, sRun);
            else if (SEARCH == SEARCH_Q10) nn_block_q10<1>(g, px, py, pz, cx, cy, cz, r2, best, bi, sRun, sPool);
            else if (SEARCH == SEARCH_F32) nn_block_top4(g, px, py, pz, cx, cy, cz, best, bi);
            else nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
            nn_outer_shells(g, px, py, pz, cx, cy, cz, smax, best, bi);
        }
        if (corr) corr[i] = bi >= 0 ? g.idx[bi] : -1;
        if (bi >= 0) icp_accumulate<MODE>(g, src_n, tgt_n, i, T, eps, px, py, pz, best, bi, acc);
    }
    __shared__ double sm[ICP_BLOCK / 64][ICP_SLOTS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) {
        double v = wave_sum(acc[q]);
        if (lane == 0) sm[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICP_SLOTS) {
        double v = 0;
        for (int w2 = 0; w2 < ICP_BLOCK / 64; w2++) v += sm[w2][threadIdx.x];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = v;   // [slot][workgroup]: the step kernel reads a slot's sums as one contiguous run
    }
}

// R3D_ICP_SPLIT=1: the evaluation as TWO kernels.  k_icp_search is k_icp_eval's correspondence search alone -- without the 29
// float64 accumulators and the Jacobian temporaries it is compiled for four waves per SIMD (k_icp_eval: three) -- and writes the
// winning target slot per query (4 B; the squared distance is recomputed from it, same expression, so nothing else need travel);
// k_icp_accum streams queries + slots and runs k_icp_eval's accumulation and reduction with the SAME query -> thread -> workgroup
// assignment, so the 768 x 29 partial sums are bit for bit those of the fused kernel.
template <int SEARCH>
__global__ void __launch_bounds__(ICP_BLOCK, 4) k_icp_search(GridView g, const double *__restrict__ src, int64_t ns, const IcpState *__restrict__ st,
                                                             double max_dist, int *__restrict__ nn) {
    __shared__ int sRun[SEARCH >= SEARCH_Q10 ? 18 * ICP_BLOCK : 1];
    // a smaller queue than k_icp_eval's (176 groups per wave: 21 KB): with the 18 KB of row lists four workgroups still fit a CU
    __shared__ __attribute__((aligned(16))) int sPoolBuf[SEARCH == SEARCH_Q10 ? (ICP_BLOCK / 64) * pool_words(POOL_Q_SEARCH) : 4];
    int *const sPool = SEARCH == SEARCH_Q10 && g.pool ? sPoolBuf : nullptr;
    if (st->done) return;
    const Rigid T = load_rigid(st);
    const double r2 = max_dist * max_dist;
    const int smax = (int)ceil(max_dist * g.inv_cell);
    const int64_t nchunk = (ns + ICP_BLOCK - 1) / ICP_BLOCK;
    int64_t c_begin = blockIdx.x, c_end = nchunk, c_step = gridDim.x;
    if ((gridDim.x & 7) == 0) {                    // XCD-aware shares: see k_icp_eval
        const int x = blockIdx.x & 7;
        c_begin = x * nchunk / 8 + (blockIdx.x >> 3);
        c_end = (x + 1) * nchunk / 8;
        c_step = gridDim.x >> 3;
    }
    for (int64_t c = c_begin; c < c_end; c += c_step) {
        const int64_t i = c * ICP_BLOCK + threadIdx.x;
        if (i >= ns) continue;
        const double sx = src[i * 3], sy = src[i * 3 + 1], sz = src[i * 3 + 2];
        const double px = T.r[0] * sx + T.r[1] * sy + T.r[2] * sz + T.t[0];
        const double py = T.r[3] * sx + T.r[4] * sy + T.r[5] * sz + T.t[1];
        const double pz = T.r[6] * sx + T.r[7] * sy + T.r[8] * sz + T.t[2];
        const int cx = cell_coord(px, g.ox, g.inv_cell), cy = cell_coord(py, g.oy, g.inv_cell), cz = cell_coord(pz, g.oz, g.inv_cell);
        double best = r2;
        int bi = -1;
        if (!icp_out_of_reach(g, cx, cy, cz)) {
            if (SEARCH == SEARCH_Q10_DEEP) nn_block_q10<3>(g, px, py, pz, cx, cy, cz, r2, best, bi, sRun);
            else if (SEARCH == SEARCH_Q10) nn_block_q10<1, POOL_Q_SEARCH>(g, px, py, pz, cx, cy, cz, r2, best, bi, sRun, sPool);
            else if (SEARCH == SEARCH_F32) nn_block_top4(g, px, py, pz, cx, cy, cz, best, bi);
            else nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
            nn_outer_shells(g, px, py, pz, cx, cy, cz, smax, best, bi);
        }
        nn[i] = bi;
    }
}
template <int MODE>
__global__ void __launch_bounds__(ICP_BLOCK, 3) k_icp_accum(GridView g, const double *__restrict__ src, const double *__restrict__ src_n,
                                                            const double *__restrict__ tgt_n, int64_t ns, const IcpState *__restrict__ st, double eps,
                                                            const int *__restrict__ nn, double *__restrict__ partial) {
    if (st->done) return;
    const Rigid T = load_rigid(st);
    double acc[ICP_SLOTS];
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) acc[q] = 0;
    const int64_t nchunk = (ns + ICP_BLOCK - 1) / ICP_BLOCK;
    int64_t c_begin = blockIdx.x, c_end = nchunk, c_step = gridDim.x;
    if ((gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        c_begin = x * nchunk / 8 + (blockIdx.x >> 3);
        c_end = (x + 1) * nchunk / 8;
        c_step = gridDim.x >> 3;
    }
    for (int64_t c = c_begin; c < c_end; c += c_step) {
        const int64_t i = c * ICP_BLOCK + threadIdx.x;
        if (i >= ns) continue;
        const int bi = nn[i];
        if (bi < 0) continue;
        const double sx = src[i * 3], sy = src[i * 3 + 1], sz = src[i * 3 + 2];
        const double px = T.r[0] * sx + T.r[1] * sy + T.r[2] * sz + T.t[0];
        const double py = T.r[3] * sx + T.r[4] * sy + T.r[5] * sz + T.t[1];
        const double pz = T.r[6] * sx + T.r[7] * sy + T.r[8] * sz + T.t[2];
        const double dx = g.pts[(int64_t)bi * 3] - px, dy = g.pts[(int64_t)bi * 3 + 1] - py, dz = g.pts[(int64_t)bi * 3 + 2] - pz;
        const double best = dx * dx + dy * dy + dz * dz;   // the search's own expression for the winner
        icp_accumulate<MODE>(g, src_n, tgt_n, i, T, eps, px, py, pz, best, bi, acc);
    }
    __shared__ double sm[ICP_BLOCK / 64][ICP_SLOTS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) {
        double v = wave_sum(acc[q]);
        if (lane == 0) sm[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICP_SLOTS) {
        double v = 0;
        for (int w2 = 0; w2 < ICP_BLOCK / 64; w2++) v += sm[w2][threadIdx.x];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = v;   // [slot][workgroup]: the step kernel reads a slot's sums as one contiguous run
    }
}

// LDS-tiled variant (R3D_ICP_IMPL=tiled; slower than the global path, see r3d_icp): one wave per 64 consecutive source points.  The source is Morton-ordered, so the cells of a
// wave's queries form a small box; the wave stages the box's cell-start entries and all target points of the box (+-1 cell)
// into LDS with coalesced loads, and every lane then walks its own nine runs out of LDS in the same order as
// nn_block_global (identical results).  Boxes that do not fit (spread-out chunks) use the global path for that chunk.
constexpr int TL_MAXROWS = 64, TL_MAXW = 15, TL_MAXPTS = 512, TL_CSP = TL_MAXW + 2;

template <int MODE>
__global__ void __launch_bounds__(64) k_icp_eval_t(GridView g, const double *__restrict__ src, const double *__restrict__ src_n,
                                                   const double *__restrict__ tgt_n, int64_t ns, const IcpState *__restrict__ st, double max_dist, double eps,
                                                   double *__restrict__ partial, int *__restrict__ corr) {
    __shared__ double lx[TL_MAXPTS], ly[TL_MAXPTS], lz[TL_MAXPTS];
    __shared__ int lcs[TL_MAXROWS * TL_CSP];  // cell starts of each box row: bw + 1 entries
    __shared__ int lbase[TL_MAXROWS + 1];     // LDS offset of each box row's points
    if (st->done) return;
    const Rigid T = load_rigid(st);
    const int lane = threadIdx.x;
    double acc[ICP_SLOTS];
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) acc[q] = 0;
    const double r2 = max_dist * max_dist;
    const int smax = (int)ceil(max_dist * g.inv_cell);
    for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < ns; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + lane;
        const bool live = i < ns;
        const int64_t ic = live ? i : ns - 1;
        const double sx = src[ic * 3], sy = src[ic * 3 + 1], sz = src[ic * 3 + 2];
        const double px = T.r[0] * sx + T.r[1] * sy + T.r[2] * sz + T.t[0];
        const double py = T.r[3] * sx + T.r[4] * sy + T.r[5] * sz + T.t[1];
        const double pz = T.r[6] * sx + T.r[7] * sy + T.r[8] * sz + T.t[2];
        const int cx = cell_coord(px, g.ox, g.inv_cell), cy = cell_coord(py, g.oy, g.inv_cell), cz = cell_coord(pz, g.oz, g.inv_cell);
        // box of the (clamped) query cells, +-1
        int mnx = min(max(cx, 0), g.nx - 1), mny = min(max(cy, 0), g.ny - 1), mnz = min(max(cz, 0), g.nz - 1);
        int mxx = mnx, mxy = mny, mxz = mnz;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o)); mnz = min(mnz, __shfl_xor(mnz, o));
            mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o)); mxz = max(mxz, __shfl_xor(mxz, o));
        }
        const int bx0 = max(mnx - 1, 0), bx1 = min(mxx + 1, g.nx - 1), by0 = max(mny - 1, 0), by1 = min(mxy + 1, g.ny - 1);
        const int bz0 = max(mnz - 1, 0), bz1 = min(mxz + 1, g.nz - 1);
        const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1, nrows = bh * (bz1 - bz0 + 1);
        bool tiled = nrows <= TL_MAXROWS && bw <= TL_MAXW;  // wave-uniform
        int total = 0;
        if (tiled) {
            for (int e = lane; e < nrows * (bw + 1); e += 64) {
                const int r = e / (bw + 1), c = e - r * (bw + 1);
                const int z = bz0 + r / bh, y = by0 + r % bh;
                lcs[r * TL_CSP + c] = cs1(g, ((int64_t)z * g.ny + y) * g.nx + bx0 + c);
            }
            __syncthreads();
            const int len = lane < nrows ? lcs[lane * TL_CSP + bw] - lcs[lane * TL_CSP] : 0;
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            total = __shfl(incl, 63);
            if (lane < nrows) lbase[lane] = incl - len;
            if (lane == 0) lbase[nrows] = total;
            tiled = total <= TL_MAXPTS;
        }
        if (tiled) {
            __syncthreads();
            for (int t = lane; t < total; t += 64) {
                int lo = 0, hi = nrows;  // row with lbase[lo] <= t < lbase[lo + 1]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (lbase[mid] <= t) lo = mid;
                    else hi = mid;
                }
                const int64_t j = lcs[lo * TL_CSP] + (t - lbase[lo]);
                lx[t] = g.pts[j * 3]; ly[t] = g.pts[j * 3 + 1]; lz[t] = g.pts[j * 3 + 2];
            }
            __syncthreads();
        }
        double best = r2;
        int bi = -1;
        if (live) {
            if (tiled) {
                const int xa = cx - 1, xb = cx + 1;
                const bool xok = xb >= 0 && xa <= g.nx - 1;
                const int x0 = min(max(xa, 0), g.nx - 1) - bx0, x1 = min(max(xb, 0), g.nx - 1) - bx0;
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
                    if (!(xok && z >= 0 && z < g.nz && y >= 0 && y < g.ny)) continue;
                    const int row = (z - bz0) * bh + (y - by0);
                    const int *cs = lcs + row * TL_CSP;
                    const int gb = cs[0], off = lbase[row] - gb;  // LDS slot = global slot + off
                    const int te = cs[x1 + 1] + off;
                    for (int t = cs[x0] + off; t < te; t++) {
                        const double dx = lx[t] - px, dy = ly[t] - py, dz = lz[t] - pz;
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 <= best) {
                            const int j = t - off;
                            if (d2 < best || (bi >= 0 && g.idx[j] < g.idx[bi])) { best = d2; bi = j; }
                        }
                    }
                }
            } else {
                nn_block_global(g, px, py, pz, cx, cy, cz, best, bi);
            }
            nn_outer_shells(g, px, py, pz, cx, cy, cz, smax, best, bi);
            if (corr) corr[i] = bi >= 0 ? g.idx[bi] : -1;
            if (bi >= 0) icp_accumulate<MODE>(g, src_n, tgt_n, i, T, eps, px, py, pz, best, bi, acc);
        }
        __syncthreads();  // the next chunk overwrites the tile
    }
#pragma unroll
    for (int q = 0; q < ICP_SLOTS; q++) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) partial[(size_t)q * gridDim.x + blockIdx.x] = v;
    }
}

__global__ void __launch_bounds__(256) k_transform(const double *__restrict__ in, int64_t n, Rigid T, int rotate_only, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
    const double tx = rotate_only ? 0 : T.t[0], ty = rotate_only ? 0 : T.t[1], tz = rotate_only ? 0 : T.t[2];
    out[i * 3] = T.r[0] * x + T.r[1] * y + T.r[2] * z + tx;
    out[i * 3 + 1] = T.r[3] * x + T.r[4] * y + T.r[5] * z + ty;
    out[i * 3 + 2] = T.r[6] * x + T.r[7] * y + T.r[8] * z + tz;
}

// k_transform over a count that is still on the device (see k_bbox_partial_dn); the grid covers the upper bound
__global__ void __launch_bounds__(256) k_transform_dn(const double *__restrict__ in, const int *__restrict__ c0, const int *__restrict__ c1, Rigid T,
                                                      double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)*c0 + *c1) return;
    const double x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
    out[i * 3] = T.r[0] * x + T.r[1] * y + T.r[2] * z + T.t[0];
    out[i * 3 + 1] = T.r[3] * x + T.r[4] * y + T.r[5] * z + T.t[1];
    out[i * 3 + 2] = T.r[6] * x + T.r[7] * y + T.r[8] * z + T.t[2];
}

// several blocks of triplets, each with its own rigid transform, in ONE launch (the fuse step of the multi-view exchange: eight
// views x (points, normals) were sixteen launches of ~2 us of work each behind ~18 us of host call overhead each)
constexpr int TB_MAX = 16;
struct TransformBlocks {
    const double *in[TB_MAX];
    double *out[TB_MAX];
    long long end[TB_MAX];      // running end offset of block b in the concatenated index space
    double r[TB_MAX][12];       // row-major 3x3 + translation (zero for rotate-only blocks)
    int n;
};
__global__ void __launch_bounds__(256) k_transform_blocks(TransformBlocks B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B.end[B.n - 1]) return;
    int b = 0;
#pragma unroll
    for (int q = 0; q < TB_MAX - 1; q++) b += (q < B.n - 1 && i >= B.end[q]) ? 1 : 0;
    const long long j = i - (b ? B.end[b - 1] : 0);
    const double *in = B.in[b];
    double *out = B.out[b];
    const double *r = B.r[b];
    const double x = in[j * 3], y = in[j * 3 + 1], z = in[j * 3 + 2];
    out[j * 3] = r[0] * x + r[1] * y + r[2] * z + r[9];          // the evaluation order of k_transform
    out[j * 3 + 1] = r[3] * x + r[4] * y + r[5] * z + r[10];
    out[j * 3 + 2] = r[6] * x + r[7] * y + r[8] * z + r[11];
}

// ------------------------------------------------------------------------------------------------ disparity -> cloud
// cv2.reprojectImageTo3D semantics: [X Y Z W]^T = Q [x y d 1]^T, point = (X, Y, Z) / W with d = disp / 16;
// pixels with disp < min_valid (the matcher's invalid marker and anything below minDisparity) are dropped.
__global__ void __launch_bounds__(256) k_disp_flags(const int16_t *__restrict__ disp, int64_t n, int min_valid, int *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = disp[i] >= min_valid ? 1 : 0;
}
struct Mat4 { double m[16]; };
// same, and additionally |Z/W| <= max_depth (the value k_reproject will write, computed by the same expressions)
__global__ void __launch_bounds__(256) k_disp_flags_depth(const int16_t *__restrict__ disp, int w, int64_t n, int min_valid, Mat4 Q,
                                                         double max_depth, int *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = (double)(i % w), y = (double)(i / w), d = (double)disp[i] / 16.0;
    const double Z = Q.m[8] * x + Q.m[9] * y + Q.m[10] * d + Q.m[11];
    const double W = Q.m[12] * x + Q.m[13] * y + Q.m[14] * d + Q.m[15];
    flags[i] = (disp[i] >= min_valid && fabs(Z / W) <= max_depth) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_reproject(const int16_t *__restrict__ disp, const int *__restrict__ flags, const int *__restrict__ scan,
                                                   int w, int64_t n, Mat4 Q, double *__restrict__ xyz, int *__restrict__ pix) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const double x = (double)(i % w), y = (double)(i / w), d = (double)disp[i] / 16.0;
    const double X = Q.m[0] * x + Q.m[1] * y + Q.m[2] * d + Q.m[3];
    const double Y = Q.m[4] * x + Q.m[5] * y + Q.m[6] * d + Q.m[7];
    const double Z = Q.m[8] * x + Q.m[9] * y + Q.m[10] * d + Q.m[11];
    const double W = Q.m[12] * x + Q.m[13] * y + Q.m[14] * d + Q.m[15];
    const int64_t o = scan[i];
    xyz[o * 3] = X / W; xyz[o * 3 + 1] = Y / W; xyz[o * 3 + 2] = Z / W;
    if (pix) pix[o] = (int)i;
}

// ------------------------------------------------------------------------------------------------ depth image -> cloud
// open3d.geometry.PointCloud.create_from_rgbd_image(RGBDImage.create_from_color_and_depth(color, depth, depth_scale, depth_trunc,
// convert_rgb_to_intensity=False), intrinsic) followed by the flip of test/check84.py:155-159,172-178 -- the rule SURVEY.md
// Appendix C verified on all 163 recorded frames: z = float(raw) / float(depth_scale) in FLOAT32, (double)z >= depth_trunc
// ([recalled] Image::ConvertDepthToFloatImage clips `*p >= depth_trunc`, the float promoted to the double parameter; the recorded
// frames cannot tell >= from > because their scale 1/float(0.001) puts raw 3000 at 3.0000002) or z == 0 dropped, x = (u - ppx) z / fx, y = (v - ppy) z / fy in float64 (u = column, v = row), then (x, -y, -z); row-major pixel order;
// colours = channel / 255.
struct DepthCam { double fx, fy, ppx, ppy, trunc; float scale; int flip; };
__device__ __forceinline__ float depth_z(unsigned short raw, const DepthCam &c) {
    const float z = (float)raw / c.scale;
    return (double)z >= c.trunc ? 0.0f : z;
}
__global__ void __launch_bounds__(256) k_depth_flags(const unsigned short *__restrict__ depth, int w, int stride, int64_t n, DepthCam c,
                                                     int *__restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flags[i] = depth_z(depth[(i / w) * stride + (i % w)], c) > 0.0f ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_backproject(const unsigned short *__restrict__ depth, const unsigned char *__restrict__ color,
                                                     const int *__restrict__ flags, const int *__restrict__ scan, int w, int stride,
                                                     int cstride, int64_t n, DepthCam c, double *__restrict__ xyz, double *__restrict__ rgb,
                                                     int *__restrict__ pix) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const int u = (int)(i % w), v = (int)(i / w);
    const double z = (double)depth_z(depth[(int64_t)v * stride + u], c);
    const double x = ((double)u - c.ppx) * z / c.fx, y = ((double)v - c.ppy) * z / c.fy;
    const int64_t o = scan[i];
    xyz[o * 3] = x; xyz[o * 3 + 1] = c.flip ? -y : y; xyz[o * 3 + 2] = c.flip ? -z : z;
    if (rgb) {
        const unsigned char *p = color + (int64_t)v * cstride + 3 * u;
        rgb[o * 3] = (double)p[0] / 255.0; rgb[o * 3 + 1] = (double)p[1] / 255.0; rgb[o * 3 + 2] = (double)p[2] / 255.0;
    }
    if (pix) pix[o] = (int)i;
}

// ------------------------------------------------------------------------------------------------ host side
// ------------------------------------------------------------------------------------------------ incremental model voxel table
// The scanning loop re-down-samples the WHOLE model every frame (pointcloud_alignment.py:23; main.py:48).  The legacy grid's voxel
// of a point depends only on the model's minimum corner (origin = min_bound - voxel / 2) and a voxel's mean is its members summed
// in index order, so while the minimum corner stays put, a frame appended at the END of the model only continues running sums:
// the table keeps, per occupied voxel in lexicographic (kx, ky, kz) order, the packed key, the running float64 sums and the
// count; a new frame's points are sorted by (voxel, index), every run is added member by member onto its voxel's running sum
// (or starts a new voxel from zero), and the new voxels are merged into the sorted table.  means = sums / count are then bit for
// bit those of a full re-voxelisation (tests/test_pipeline_gpu.py asserts it; R3D_MODEL_IMPL=rebuild keeps the full pass).
__device__ __forceinline__ unsigned long long vt_key_of(const double *__restrict__ p, double ox, double oy, double oz, double voxel) {
    const long long kx = (long long)floor((p[0] - ox) / voxel), ky = (long long)floor((p[1] - oy) / voxel), kz = (long long)floor((p[2] - oz) / voxel);
    return (unsigned long long)kx << 42 | (unsigned long long)ky << 21 | (unsigned long long)kz;   // each index < 2^21 (checked on the host)
}
// table from scratch: one thread per voxel segment of the (voxel, index)-sorted model (the loop of k_voxel_mean_sorted)
template <int R>
__global__ void __launch_bounds__(256) k_vt_build(const double *__restrict__ sp, const int *__restrict__ starts, int64_t nseg, int64_t n, double ox,
                                                  double oy, double oz, double voxel, unsigned long long *__restrict__ keys,
                                                  double *__restrict__ sums, int *__restrict__ cnt) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const int b = starts[s], e = s + 1 < nseg ? starts[s + 1] : (int)n, m = e - b;
    const double *__restrict__ p = sp + (int64_t)b * 3;
    double x = 0, y = 0, z = 0;
    double cur[3 * R], nxt[3 * R];
#pragma unroll
    for (int u = 0; u < R; u++) {
        const int64_t q = (int64_t)min(u, m - 1) * 3;
        cur[u * 3] = p[q]; cur[u * 3 + 1] = p[q + 1]; cur[u * 3 + 2] = p[q + 2];
    }
    keys[s] = vt_key_of(cur, ox, oy, oz, voxel);
    for (int i0 = 0; i0 < m; i0 += R) {
#pragma unroll
        for (int u = 0; u < R; u++) {
            const int64_t q = (int64_t)min(i0 + R + u, m - 1) * 3;
            nxt[u * 3] = p[q]; nxt[u * 3 + 1] = p[q + 1]; nxt[u * 3 + 2] = p[q + 2];
        }
#pragma unroll
        for (int u = 0; u < R; u++)
            if (i0 + u < m) { x += cur[u * 3]; y += cur[u * 3 + 1]; z += cur[u * 3 + 2]; }
#pragma unroll
        for (int u = 0; u < 3 * R; u++) cur[u] = nxt[u];
    }
    sums[s * 3] = x; sums[s * 3 + 1] = y; sums[s * 3 + 2] = z;
    cnt[s] = m;
}
__device__ __forceinline__ int64_t vt_lower_bound(const unsigned long long *__restrict__ keys, int64_t n, unsigned long long k) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// one thread per run of the sorted slice: continue the voxel's running sum in place, or start a new voxel
__global__ void __launch_bounds__(256) k_vt_update(const double *__restrict__ sp, const int *__restrict__ starts, int64_t nseg, int64_t n, double ox,
                                                   double oy, double oz, double voxel, const unsigned long long *__restrict__ tkeys, int64_t tn,
                                                   double *__restrict__ tsums, int *__restrict__ tcnt, unsigned long long *__restrict__ segkey,
                                                   int *__restrict__ is_new, int64_t *__restrict__ ins, double *__restrict__ nsum, int *__restrict__ ncnt) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const int b = starts[s], e = s + 1 < nseg ? starts[s + 1] : (int)n;
    const double *__restrict__ p = sp + (int64_t)b * 3;
    const unsigned long long k = vt_key_of(p, ox, oy, oz, voxel);
    const int64_t pos = vt_lower_bound(tkeys, tn, k);
    const bool found = pos < tn && tkeys[pos] == k;
    double x = 0, y = 0, z = 0;
    if (found) { x = tsums[pos * 3]; y = tsums[pos * 3 + 1]; z = tsums[pos * 3 + 2]; }
    for (int i = 0; i < e - b; i++) { x += p[(int64_t)i * 3]; y += p[(int64_t)i * 3 + 1]; z += p[(int64_t)i * 3 + 2]; }
    segkey[s] = k;
    is_new[s] = found ? 0 : 1;
    ins[s] = pos;
    if (found) {
        tsums[pos * 3] = x; tsums[pos * 3 + 1] = y; tsums[pos * 3 + 2] = z;
        tcnt[pos] += e - b;
    } else {
        nsum[s * 3] = x; nsum[s * 3 + 1] = y; nsum[s * 3 + 2] = z;
        ncnt[s] = e - b;
    }
}
// merge: old voxel i moves up by the number of NEW keys below it; new voxel s goes to (old keys below it) + (new keys below it)
__global__ void __launch_bounds__(256) k_vt_merge(const unsigned long long *__restrict__ tkeys, const double *__restrict__ tsums, const int *__restrict__ tcnt,
                                                  int64_t tn, const unsigned long long *__restrict__ segkey, const int *__restrict__ is_new,
                                                  const int *__restrict__ rank /* exclusive scan of is_new */, const int64_t *__restrict__ ins,
                                                  const double *__restrict__ nsum, const int *__restrict__ ncnt, int64_t nseg, int total_new,
                                                  unsigned long long *__restrict__ okeys, double *__restrict__ osums, int *__restrict__ ocnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < tn) {
        const unsigned long long k = tkeys[i];
        const int64_t lb = vt_lower_bound(segkey, nseg, k);          // runs with a smaller key
        const int64_t pos = i + (lb < nseg ? rank[lb] : total_new);
        okeys[pos] = k;
        osums[pos * 3] = tsums[i * 3]; osums[pos * 3 + 1] = tsums[i * 3 + 1]; osums[pos * 3 + 2] = tsums[i * 3 + 2];
        ocnt[pos] = tcnt[i];
    } else if (i < tn + nseg) {
        const int64_t s = i - tn;
        if (!is_new[s]) return;
        const int64_t pos = ins[s] + rank[s];
        okeys[pos] = segkey[s];
        osums[pos * 3] = nsum[s * 3]; osums[pos * 3 + 1] = nsum[s * 3 + 1]; osums[pos * 3 + 2] = nsum[s * 3 + 2];
        ocnt[pos] = ncnt[s];
    }
}
__global__ void __launch_bounds__(256) k_vt_means(const double *__restrict__ sums, const int *__restrict__ cnt, int64_t n, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double c = (double)cnt[i];
    out[i * 3] = sums[i * 3] / c; out[i * 3 + 1] = sums[i * 3 + 1] / c; out[i * 3 + 2] = sums[i * 3 + 2] / c;
}

struct DevArena {  // simple bump allocator over ctx->cloud_bufs (grow-only, reused across calls)
    r3d_ctx *ctx;
    size_t next = 0;
    int rc = R3D_OK;
    explicit DevArena(r3d_ctx *c) : ctx(c) {}
    void *get(size_t bytes) {
        if (rc) return nullptr;
        if (ctx->poisoned) {
            rc = r3d_fail(ctx, R3D_E_HIP, "context poisoned by an earlier timed-out call (its kernels may still use the arena): destroy it");
            return nullptr;
        }
        if (next >= ctx->cloud_bufs.size()) ctx->cloud_bufs.emplace_back();
        r3d_buf &b = ctx->cloud_bufs[next++];
        rc = r3d_reserve(ctx, b, bytes ? bytes : 16);
        return rc ? nullptr : b.p;
    }
};

// Small device -> host reads between kernels (counts, bounding boxes).  k_publish copies every piece into the context's pinned,
// device-mapped buffer and then stores a sequence number there (system-scope fence in between); the host polls that number in
// memory.  No copy command, no marker, no runtime call in the wait: two staged 4-byte hipMemcpyAsync + a blocking synchronise
// cost ~50 us of idle GPU per round trip, two pinned copies + an event poll ~25 us (and were, in the registration loop, the part
// whose completion sometimes reached the host tens of milliseconds late), this form ~15 us.
struct PubDesc { const void *src[8]; unsigned off[8], bytes[8]; int n; unsigned seq; };
__global__ void __launch_bounds__(256) k_publish(PubDesc d, char *__restrict__ host) {
    for (int q = 0; q < d.n; q++) {
        const char *s = (const char *)d.src[q];
        for (unsigned i = threadIdx.x * 4; i < d.bytes[q]; i += 256 * 4) *(int *)(host + d.off[q] + i) = *(const int *)(s + i);
    }
    // every writer makes its stores visible at system scope, the barrier orders them before thread 0, which releases once more before
    // the sequence number goes out (PCIe keeps posted writes of one device in order)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        *(volatile unsigned *)host = d.seq;
    }
}
struct PinRead {
    r3d_ctx *ctx;
    size_t off = 64;   // the first 64 bytes of the buffer hold the sequence number
    PubDesc d;
    void *host[8];
    explicit PinRead(r3d_ctx *c) : ctx(c) { d.n = 0; }
    int add(void *host_dst, const void *dev, size_t bytes) {   // bytes: a multiple of 4
        if (!ctx->pin) {
            hipError_t e = hipHostMalloc(&ctx->pin, R3D_PIN_BYTES, hipHostMallocMapped);
            if (e != hipSuccess) { ctx->pin = nullptr; return r3d_fail(ctx, R3D_E_OOM, "hipHostMalloc(%d) failed: %s", R3D_PIN_BYTES, hipGetErrorString(e)); }
            memset(ctx->pin, 0, 64);
        }
        if (d.n >= 8 || (bytes & 3) || off + bytes > R3D_PIN_BYTES) return r3d_fail(ctx, R3D_E_HIP, "PinRead: too much for one read-back");
        d.src[d.n] = dev; d.off[d.n] = (unsigned)off; d.bytes[d.n] = (unsigned)bytes;
        host[d.n++] = host_dst;
        off += (bytes + 15) & ~(size_t)15;
        return R3D_OK;
    }
    int wait() {
        char *dpin = nullptr;
        R3D_HIP(ctx, hipHostGetDevicePointer((void **)&dpin, ctx->pin, 0));
        d.seq = ++ctx->pin_seq;
        k_publish<<<1, 256, 0, ctx->stream>>>(d, dpin);
        R3D_HIP(ctx, hipGetLastError());
        // the read is a few microseconds away: poll the sequence number briefly, then wait properly (a blocking synchronise may put
        // the thread to sleep for a scheduler tick)
        const auto t0 = std::chrono::steady_clock::now();
        bool done = false;
        for (unsigned spins = 0; !done; spins++) {
            if (*(volatile unsigned *)ctx->pin == d.seq) { done = true; break; }
            if ((spins & 255) == 255 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1e-3) break;
        }
        if (!done) {
            R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (*(volatile unsigned *)ctx->pin != d.seq) return r3d_fail(ctx, R3D_E_HIP, "PinRead: the published sequence number did not arrive");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int i = 0; i < d.n; i++) memcpy(host[i], (const char *)ctx->pin + d.off[i], d.bytes[i]);
        d.n = 0;
        off = 64;
        return R3D_OK;
    }
};

// big_ws: >= (nseg + 16) ints of scratch (the counter, then the list)
static inline int launch_voxel_mean(r3d_ctx *ctx, DevArena &ar, bool f32, const double *a, const int *idx, const int *starts, int64_t nseg, int64_t n,
                                    double *out, const double *sorted = nullptr) {
    if (sorted) {   // `a` in sorted order is available (the sort wrote it): contiguous streams, no index list
        const unsigned nbs = (unsigned)((nseg + 255) / 256);
        // R = 8 members per round trip: 4 / 8 / 16 measured 4.71 / 4.54-4.59 / 4.59-4.64 ms per C5 view, interleaved on one box
        if (f32) k_voxel_mean_sorted<float, 8><<<nbs, 256, 0, ctx->stream>>>(sorted, starts, nseg, n, out);
        else k_voxel_mean_sorted<double, 8><<<nbs, 256, 0, ctx->stream>>>(sorted, starts, nseg, n, out);
        R3D_HIP(ctx, hipGetLastError());
        return R3D_OK;
    }
    int *ws = (int *)ar.get((size_t)(nseg + 16) * 4);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemsetAsync(ws, 0, 64, ctx->stream));
    const unsigned nb = (unsigned)((nseg + 255) / 256);
    const unsigned nbig = (unsigned)std::min<int64_t>(nseg, 4096);
    if (f32) {
        k_voxel_mean_t<float><<<nb, 256, 0, ctx->stream>>>(a, idx, starts, nseg, n, out, ws + 16, ws);
        k_voxel_mean_big<float><<<nbig, 64, 0, ctx->stream>>>(a, idx, starts, nseg, n, out, ws + 16, ws);
    } else {
        k_voxel_mean_t<double><<<nb, 256, 0, ctx->stream>>>(a, idx, starts, nseg, n, out, ws + 16, ws);
        k_voxel_mean_big<double><<<nbig, 64, 0, ctx->stream>>>(a, idx, starts, nseg, n, out, ws + 16, ws);
    }
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

struct Grid {
    GridView v;
    int64_t n = 0;
    int64_t ncells = 0;
    void *keys = nullptr;   // sorted keys: unsigned if keys32, else unsigned long long
    bool keys32 = false;
    double mn[3], mx[3];
};

// a point at infinity (e.g. a zero disparity reprojected through Q) or a NaN has no cell: refuse instead of building a grid
// around it (fmin/fmax drop NaNs, so those show up as an untouched +-1e300 bound only if every value is NaN)
int bbox_check(r3d_ctx *ctx, const double mn[3], const double mx[3]) {
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]) || mn[a] > mx[a] || std::fabs(mn[a]) > 1e290 || std::fabs(mx[a]) > 1e290)
            return r3d_fail(ctx, R3D_E_BADARG, "cloud has non-finite coordinates (axis %d spans [%g, %g])", a, mn[a], mx[a]);
    return R3D_OK;
}
// enqueues the bounding box of the first (*c0 + *c1) points (at most `bound`) and returns where its 6 doubles will be: the caller
// reads them with its next read-back, together with the count
int cloud_bbox_enqueue_dn(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t bound, const int *c0, const int *c1, const double **d_box6,
                          double *d_dst6 = nullptr /* where to leave the box (a buffer that outlives the arena); default: arena */) {
    const int nb = (int)std::min<int64_t>((bound + 255) / 256, 1024);
    double *part = (double *)ar.get((size_t)(nb + 1) * 6 * 8);
    if (ar.rc) return ar.rc;
    if (c0) k_bbox_partial_dn<<<nb, 256, 0, ctx->stream>>>(d_pts, c0, c1, part);
    else k_bbox_partial<<<nb, 256, 0, ctx->stream>>>(d_pts, bound, part);       // the count is known: `bound` IS the count
    double *dst = d_dst6 ? d_dst6 : part + (size_t)nb * 6;
    k_bbox_final<<<1, 256, 0, ctx->stream>>>(part, nb, dst);
    R3D_HIP(ctx, hipGetLastError());
    if (d_box6) *d_box6 = dst;
    return R3D_OK;
}
// an extra item for somebody else's read-back (PinRead): a value the caller wants on the host, produced earlier on the device
struct PinExtra { void *host = nullptr; const void *dev = nullptr; size_t bytes = 0; };
int cloud_bbox(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t n, double mn[3], double mx[3]) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 1024);
    double *part = (double *)ar.get((size_t)(nb + 1) * 6 * 8);
    if (ar.rc) return ar.rc;
    k_bbox_partial<<<nb, 256, 0, ctx->stream>>>(d_pts, n, part);
    k_bbox_final<<<1, 256, 0, ctx->stream>>>(part, nb, part + (size_t)nb * 6);
    R3D_HIP(ctx, hipGetLastError());
    double h[6];
    {
        PinRead rd(ctx);
        int prc;
        if ((prc = rd.add(h, part + (size_t)nb * 6, sizeof h)) || (prc = rd.wait())) return prc;
    }
    for (int a = 0; a < 3; a++) { mn[a] = h[a]; mx[a] = h[3 + a]; }
    return bbox_check(ctx, mn, mx);
}

// exclusive scan of n ints / packed 64-bit counters on the ctx stream (k_scan_sums + k_scan_apply)
template <class T, bool MAXOP = false>
int dev_exclusive_scan(r3d_ctx *ctx, DevArena &ar, const T *in, T *out, int64_t n, int *lo_out = nullptr, int *hi_max = nullptr) {
    if (n <= 0) return R3D_OK;
    const int tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    T *part = (T *)ar.get((size_t)tiles * sizeof(T));
    if (ar.rc) return ar.rc;
    k_scan_sums<T, MAXOP><<<tiles, SCAN_T, 0, ctx->stream>>>(in, n, part);
    k_scan_apply<T, MAXOP><<<tiles, SCAN_T, 0, ctx->stream>>>(in, n, part, out, lo_out, hi_max);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

// ---- stable LSD radix sort of (key, index) pairs, hand-written (rounds 1-3 called hipCUB here).  8 bits per pass; a pass is
// k_rx_hist (per-tile digit histogram) -> exclusive scan of the [256][tiles] table (digit-major: all tiles of digit 0, then digit
// 1, ...) -> k_rx_scatter (every item to scanned base of its (digit, tile) + its rank among the tile's items of that digit in
// original order).  A tile is ONE wave x 16 items, walked 64 items at a time: an item's rank inside such a group is the number of
// lower lanes holding the same digit -- eight ballots give every lane the mask of its peers -- plus a running per-digit count in
// LDS that the group's first peer advances; no cross-wave ordering to arrange, and the order is the stable one by construction.
// No host round trip, no fallback: used for cell sorts of large clouds without run structure (search-grid / Morton keys of an
// arbitrary cloud, where every point is its own run and the counting sort above degenerates to sorting runs one by one) and
// wherever the counting sort declines.
constexpr int RX_ITEMS = 16, RX_TILE = 64 * RX_ITEMS;
template <class KEY>
__global__ void __launch_bounds__(64) k_rx_hist(const KEY *__restrict__ keys, int64_t n, int shift, int *__restrict__ hist, int ntiles) {
    __shared__ int cnt[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) cnt[d] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RX_TILE;
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++) {
        const int64_t i = base + j * 64 + lane;
        if (i < n) atomicAdd(&cnt[(int)((keys[i] >> shift) & 255)], 1);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(size_t)d * ntiles + blockIdx.x] = cnt[d];
}
template <class KEY, bool FIRST>
__global__ void __launch_bounds__(64) k_rx_scatter(const KEY *__restrict__ kin, const int *__restrict__ vin, int64_t n, int shift,
                                                   const int *__restrict__ offs, int ntiles, KEY *__restrict__ kout, int *__restrict__ vout) {
    __shared__ int base[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) base[d] = offs[(size_t)d * ntiles + blockIdx.x];
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * RX_TILE;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));      // lanes below this one
    for (int j = 0; j < RX_ITEMS; j++) {
        const int64_t i = t0 + j * 64 + lane;
        const bool live = i < n;
        const KEY key = live ? kin[i] : (KEY)0;
        const int val = FIRST ? (int)i : (live ? vin[i] : 0);
        const int d = (int)((key >> shift) & 255);
        unsigned long long peers = __ballot(live);                                  // live lanes with my digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot(live && ((d >> b) & 1));
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int rank = __popcll(peers & lt), group = __popcll(peers);
        int pos = 0;
        if (live) pos = base[d] + rank;
        __syncthreads();                          // every lane has read its digit's count before the first peer advances it
        if (live && rank == 0) base[d] += group;
        __syncthreads();
        if (live) { kout[pos] = key; vout[pos] = val; }
    }
}

// sorts (cell key, index) pairs with the hand-written radix sort; keys come from k_cell_keys
template <class KEY>
int radix_sort_by_cell_t(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t n, const double org[3], double cell, const int dims[3],
                         int key_order, int bits, void **keys_sorted, int **idx_sorted) {
    KEY *k0 = (KEY *)ar.get((size_t)n * sizeof(KEY)), *k1 = (KEY *)ar.get((size_t)n * sizeof(KEY));
    int *v0 = (int *)ar.get((size_t)n * 4), *v1 = (int *)ar.get((size_t)n * 4);
    const int ntiles = (int)((n + RX_TILE - 1) / RX_TILE);
    int *hist = (int *)ar.get((size_t)256 * ntiles * 4), *offs = (int *)ar.get((size_t)256 * ntiles * 4);
    if (ar.rc) return ar.rc;
    const int nb = (int)((n + 255) / 256);
    k_cell_keys<KEY><<<nb, 256, 0, ctx->stream>>>(d_pts, n, org[0], org[1], org[2], cell, dims[0], dims[1], dims[2], key_order, k0, v0);
    R3D_HIP(ctx, hipGetLastError());
    const int passes = std::max(1, (bits + 7) / 8);
    KEY *kin = k0, *kout = k1;
    int *vin = v0, *vout = v1;
    for (int ps = 0; ps < passes; ps++) {
        k_rx_hist<KEY><<<ntiles, 64, 0, ctx->stream>>>(kin, n, 8 * ps, hist, ntiles);
        if (int rc = dev_exclusive_scan<int>(ctx, ar, hist, offs, (int64_t)256 * ntiles)) return rc;
        k_rx_scatter<KEY, false><<<ntiles, 64, 0, ctx->stream>>>(kin, vin, n, 8 * ps, offs, ntiles, kout, vout);   // (k_cell_keys wrote vals = index)
        R3D_HIP(ctx, hipGetLastError());
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    *keys_sorted = kin;
    *idx_sorted = vin;
    return R3D_OK;
}

constexpr int CS_MAX_RUNS = 4096;          // runs per bucket above which k_cs_rank's quadratic step is declined (radix fallback)
constexpr int64_t CS_MAX_BUCKETS = 1ll << 26;
// A sort whose host round trip (run count + fallback test) is DEFERRED: the caller adds d_total / d_maxruns to its own next
// read-back (PinRead) and calls ok(); if that says no, the sorted arrays are garbage and the caller sorts again with impl = 1.
struct SortDefer {
    const unsigned long long *d_total = nullptr;   // runs << 32 | points
    const int *d_maxruns = nullptr;
    int64_t n = 0;
    bool used = false;                             // false: the sort that ran was complete on its own (radix path)
    unsigned long long h_total = 0;
    int h_maxruns = 0;
    int add_to(PinRead &rd) { if (!used) return R3D_OK; if (int rc = rd.add(&h_total, d_total, 8)) return rc; return rd.add(&h_maxruns, d_maxruns, 4); }
    bool ok() const { return !used || (h_maxruns <= CS_MAX_RUNS && (int64_t)(unsigned)h_total == n); }
};
// returns R3D_OK with *done = false when it declines (more than CS_MAX_RUNS runs in one bucket)
template <class KEY>
int counting_sort_by_cell_t(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t n, const CsGeom &g, int64_t nbuckets, void **keys_sorted,
                            int **idx_sorted, double *sorted_pts, bool *done, SortDefer *defer = nullptr) {
    *done = false;
    KEY *keys = (KEY *)ar.get((size_t)n * sizeof(KEY)), *keys_out = (KEY *)ar.get((size_t)n * sizeof(KEY));
    int *gid_of = (int *)ar.get((size_t)n * 4), *idx_out = (int *)ar.get((size_t)n * 4), *final_base = (int *)ar.get((size_t)n * 4);
    CsRun<KEY> *runs = (CsRun<KEY> *)ar.get((size_t)n * sizeof(CsRun<KEY>)), *placed = (CsRun<KEY> *)ar.get((size_t)n * sizeof(CsRun<KEY>));
    // bucket table (nbuckets + 1 entries, the last one stays 0) followed by the counter cell: ONE fill zeroes both
    unsigned long long *cnt = (unsigned long long *)ar.get((size_t)(nbuckets + 1 + 8) * 8), *start = (unsigned long long *)ar.get((size_t)(nbuckets + 1) * 8);
    if (ar.rc) return ar.rc;
    int *ctr = (int *)(cnt + nbuckets + 1);   // [0] most runs in one bucket
    R3D_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)(nbuckets + 1 + 8) * 8, ctx->stream));
    const int nb = (int)((n + 255) / 256);
    k_cs_runs<KEY><<<nb, 256, 0, ctx->stream>>>(d_pts, n, g, keys, gid_of, cnt, runs);
    int rc = dev_exclusive_scan<unsigned long long>(ctx, ar, cnt, start, nbuckets + 1, nullptr, ctr);
    if (rc) return rc;
    if (defer) {   // no round trip here: launch over the upper bounds, counts and the fallback test stay on the device for now
        k_cs_place<KEY><<<nb, 256, 0, ctx->stream>>>(runs, gid_of, n, start, placed);
        k_cs_rank<KEY><<<nb, 256, 0, ctx->stream>>>(placed, 0, start, final_base, start + nbuckets, ctr, CS_MAX_RUNS);
        k_cs_emit<KEY><<<nb, 256, 0, ctx->stream>>>(keys, gid_of, final_base, n, keys_out, idx_out, d_pts, sorted_pts, ctr, CS_MAX_RUNS);
        R3D_HIP(ctx, hipGetLastError());
        defer->d_total = start + nbuckets;
        defer->d_maxruns = ctr;
        defer->n = n;
        defer->used = true;
        *keys_sorted = keys_out;
        *idx_sorted = idx_out;
        *done = true;
        return R3D_OK;
    }
    // ONE host round trip: the scan total (runs << 32 | points: cnt[nbuckets] is 0, so start[nbuckets] is the total) and the
    // largest run count of a bucket, which decides whether the quadratic ranking step is affordable
    struct { unsigned long long total; int max_runs; } h = {0, 0};
    {
        PinRead rd(ctx);
        int prc;
        if ((prc = rd.add(&h.total, start + nbuckets, 8)) || (prc = rd.add(&h.max_runs, ctr, 4)) || (prc = rd.wait())) return prc;
    }
    if ((int64_t)(unsigned)h.total != n) return r3d_fail(ctx, R3D_E_HIP, "counting sort: bucket populations sum to %u, expected %lld", (unsigned)h.total, (long long)n);
    static const bool dbg = [] { const char *e = getenv("R3D_SORT_DEBUG"); return e && *e == '1'; }();
    if (dbg) fprintf(stderr, "[r3d sort] n=%lld key_order=%d buckets=%lld runs=%d max_runs_per_bucket=%d keys%d\n", (long long)n, g.key_order,
                     (long long)nbuckets, (int)(h.total >> 32), h.max_runs, (int)sizeof(KEY) * 8);
    if (h.max_runs > CS_MAX_RUNS) return R3D_OK;
    const int nruns = (int)(h.total >> 32), nrb = (nruns + 255) / 256;
    k_cs_place<KEY><<<nb, 256, 0, ctx->stream>>>(runs, gid_of, n, start, placed);
    k_cs_rank<KEY><<<nrb, 256, 0, ctx->stream>>>(placed, nruns, start, final_base, nullptr, nullptr, 0);
    k_cs_emit<KEY><<<nb, 256, 0, ctx->stream>>>(keys, gid_of, final_base, n, keys_out, idx_out, d_pts, sorted_pts, nullptr, 0);
    R3D_HIP(ctx, hipGetLastError());
    *keys_sorted = keys_out;
    *idx_sorted = idx_out;
    *done = true;
    return R3D_OK;
}

// sorts point indices by (cell key, index); fills keys_sorted / idx_sorted (device) and, if sorted_pts != NULL, the points in that
// order.  The keys are 32-bit whenever the key space allows; *keys32 tells the caller which type keys_sorted points to.
// Default: the run-based counting sort above; the library radix sort when that declines.
int sort_by_cell(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t n, const double org[3], double cell, const int dims[3],
                 int key_order, void **keys_sorted, bool *keys32, int **idx_sorted, double *sorted_pts = nullptr, int impl = -1,
                 SortDefer *defer = nullptr) {
    if (defer) *defer = SortDefer();
    int bits = 1;
    if (key_order == 2) {
        int m = std::max(dims[0], std::max(dims[1], dims[2])), b1 = 1;
        while ((1 << b1) < m) b1++;
        bits = std::min(63, 3 * b1);
    } else {
        const unsigned long long maxkey = (unsigned long long)dims[0] * dims[1] * dims[2];
        while (bits < 64 && (maxkey >> bits)) bits++;
    }
    *keys32 = bits <= 32;
    static const int env_impl = [] { const char *e = getenv("R3D_SORT_IMPL"); return !e ? -1 : !strcmp(e, "radix") ? 1 : !strcmp(e, "counting") ? 0 : -1; }();
    // default choice: the counting sort where keys arrive in runs (voxel keys of image- or voxel-ordered clouds) and for small
    // clouds (six launches + one round trip); the radix sort for search-grid / Morton keys of large clouds, which have no run
    // structure to exploit (every point its own run: 1 M points 0.19 ms against 0.09 ms, and no host round trip)
    static const int64_t rx_min = [] { const char *e = getenv("R3D_SORT_RADIX_MIN"); return e ? (int64_t)atoll(e) : (int64_t)262144; }();
    const bool prefer_radix = (key_order == 0 || key_order == 2) && n >= rx_min;
    const bool force_radix = impl >= 0 ? impl == 1 : (env_impl >= 0 ? env_impl == 1 : prefer_radix);
    // bucket table of the counting sort: a monotone coarsening of the key (consecutive keys share a bucket) with about half as
    // many entries as there are points (2^14 .. 2^26): small enough to fill and scan in a few microseconds, fine enough that a
    // bucket holds tens of runs (the ranking step is quadratic in the runs of a bucket: whole (kx, ky) columns of an 8 MP view,
    // pierced lengthwise by a slanted surface, held up to 1 500)
    CsGeom g{org[0], org[1], org[2], cell, dims[0], dims[1], dims[2], key_order, 0, 1};
    int64_t nbuckets;
    int tb = 14;
    // voxel keys of image-ordered clouds arrive in runs of several points (n / 2 buckets); search-grid and Morton keys of an
    // unordered cloud are one run per point, so the ranking step wants few points per bucket (2 n buckets)
    const int64_t want = (key_order == 1 || key_order == 3) ? n / 2 : 2 * n;
    while (tb < 26 && (1ll << tb) < want) tb++;
    if (key_order == 1 || key_order == 3) {
        const long double total = (long double)dims[0] * dims[1] * dims[2];
        unsigned long long d = (unsigned long long)(total / (long double)(1ll << tb)) + 1;
        g.shift = -1;
        g.div = d;
        nbuckets = (int64_t)(total / (long double)d) + 2;
    } else {
        g.shift = std::max(0, bits - tb);
        nbuckets = key_order == 2 ? (1ll << (bits - g.shift)) : (int64_t)(((unsigned long long)dims[0] * dims[1] * dims[2]) >> g.shift) + 1;
    }
    if (!force_radix && nbuckets <= CS_MAX_BUCKETS && n < 0x7fffffff) {
        bool done = false;
        const int rc = *keys32 ? counting_sort_by_cell_t<unsigned>(ctx, ar, d_pts, n, g, nbuckets, keys_sorted, idx_sorted, sorted_pts, &done, defer)
                               : counting_sort_by_cell_t<unsigned long long>(ctx, ar, d_pts, n, g, nbuckets, keys_sorted, idx_sorted, sorted_pts, &done, defer);
        if (rc || done) return rc;
    }
    const int rc = *keys32 ? radix_sort_by_cell_t<unsigned>(ctx, ar, d_pts, n, org, cell, dims, key_order, bits, keys_sorted, idx_sorted)
                           : radix_sort_by_cell_t<unsigned long long>(ctx, ar, d_pts, n, org, cell, dims, key_order, bits, keys_sorted, idx_sorted);
    if (rc) return rc;
    if (sorted_pts) {
        k_gather3<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_pts, *idx_sorted, n, sorted_pts);
        R3D_HIP(ctx, hipGetLastError());
    }
    return R3D_OK;
}

// Builds the search grid.  cell_hint: minimum useful cell (search radius, or <= 0 for pure kNN); the cell is
// refined so that occupied cells hold about `target_occ` points, and coarsened to keep the dense table <= 2^26 cells.
// enclosing[6] (optional): a box known to contain every point (e.g. the bounding box of the cloud these points are voxel means
// of): used, slightly widened, instead of a bounding-box pass and its host round trip.  A search grid's origin and extent decide
// only which cell a point is filed under; what the searches return does not depend on them.
// spacing_hint (optional): the cloud is a voxel-down-sampled surface with this voxel size, so a cell of side c holds about
// (c / spacing)^2 points -- used instead of the bounding-box estimate below, which is off by 3-6x on a room-sized scan (the box's
// faces are not the surface) and then picks cells of HALF the correspondence radius: every query whose nearest point is further
// than one such cell -- most of them while a scan frame is still misaligned by a centimetre -- walks the 98 cells of shell 2.
int grid_build(r3d_ctx *ctx, DevArena &ar, const double *d_pts, int64_t n, double cell_hint, double target_occ, Grid &G, bool dense_table = false,
               const double *enclosing = nullptr, double spacing_hint = 0) {
    if (n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "grid: empty cloud");
    if (n > 0x7fffffff) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "grid: more than 2^31-1 points");
    int rc;
    if (enclosing) {
        for (int a = 0; a < 3; a++) {   // a mean of values inside [lo, hi] can leave it by a rounding: widen by far more than that
            const double pad = 1e-9 * std::max(std::max(std::fabs(enclosing[a]), std::fabs(enclosing[3 + a])), enclosing[3 + a] - enclosing[a]) + 1e-300;
            G.mn[a] = enclosing[a] - pad;
            G.mx[a] = enclosing[3 + a] + pad;
        }
        if ((rc = bbox_check(ctx, G.mn, G.mx))) return rc;
    } else if ((rc = cloud_bbox(ctx, ar, d_pts, n, G.mn, G.mx))) return rc;
    double ext[3];
    for (int a = 0; a < 3; a++) ext[a] = std::max(G.mx[a] - G.mn[a], 1e-9);
    // surface-like data: occupied cells ~ (extent/cell)^2 * const.  Start from the volume / area heuristic and clamp.
    const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    double cell = cell_hint > 0 ? cell_hint : diag / std::max(1.0, std::sqrt((double)n / std::max(target_occ, 1.0)));
    if (cell_hint > 0 && target_occ > 0) {
        // estimate occupancy at cell_hint assuming a 2-manifold: area ~ n * spacing^2, spacing from bbox area
        const double area = 2.0 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2]) / 2.0;  // rough
        const double per_cell = spacing_hint > 0 ? (cell_hint / spacing_hint) * (cell_hint / spacing_hint)
                                                 : (double)n * cell_hint * cell_hint / std::max(area, 1e-30);
        int m = (int)std::floor(std::sqrt(std::max(per_cell / target_occ, 1.0)));
        m = std::max(1, std::min(m, 16));
        cell = cell_hint / m;
    }
    int dims[3];
    for (;;) {
        double nc = 1;
        for (int a = 0; a < 3; a++) { dims[a] = (int)std::floor(ext[a] / cell) + 1; nc *= dims[a]; }
        if (nc <= (double)(1 << 26)) break;
        cell *= 1.26;
    }
    G.n = n;
    G.ncells = (int64_t)dims[0] * dims[1] * dims[2];
    int *idx;
    double *sorted = (double *)ar.get((size_t)n * 24);
    if (ar.rc) return ar.rc;
    rc = sort_by_cell(ctx, ar, d_pts, n, G.mn, cell, dims, 0, &G.keys, &G.keys32, &idx, sorted);   // writes the cell-sorted copy as well
    if (rc) return rc;
    // R3D_CELL_TABLE = dense | lines overrides the caller's choice (A/B)
    static const int table_env = [] { const char *e = getenv("R3D_CELL_TABLE"); return !e ? -1 : !strcmp(e, "dense") ? 1 : !strcmp(e, "lines") ? 0 : -1; }();
    if (table_env >= 0) dense_table = table_env == 1;
    if (dense_table) {
        int *cnt = (int *)ar.get((size_t)(G.ncells + 1) * 4), *rs = (int *)ar.get((size_t)(G.ncells + 1) * 4);
        int *cs = (int *)ar.get((size_t)(G.ncells + 1 + 4) * 4);   // + 4: the search reads 16 bytes at a run's first cell
        if (ar.rc) return ar.rc;
        R3D_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)(G.ncells + 1) * 4, ctx->stream));
        const int nbd = (int)((n + 255) / 256);
        if (G.keys32) {
            k_cell_counts<unsigned><<<nbd, 256, 0, ctx->stream>>>((const unsigned *)G.keys, n, rs, cnt);
            k_cell_counts2<unsigned><<<nbd, 256, 0, ctx->stream>>>((const unsigned *)G.keys, n, rs, cnt);
        } else {
            k_cell_counts<unsigned long long><<<nbd, 256, 0, ctx->stream>>>((const unsigned long long *)G.keys, n, rs, cnt);
            k_cell_counts2<unsigned long long><<<nbd, 256, 0, ctx->stream>>>((const unsigned long long *)G.keys, n, rs, cnt);
        }
        if ((rc = dev_exclusive_scan<int>(ctx, ar, cnt, cs, G.ncells + 1))) return rc;
        G.v = GridView{G.mn[0], G.mn[1], G.mn[2], cell, 1.0 / cell, dims[0], dims[1], dims[2], nullptr, nullptr, cs, sorted, idx, nullptr, nullptr, nullptr, 0.f, nullptr};
        return R3D_OK;
    }
    // two-level cell table from the sorted keys (GridView::l1 / l2); sized for the worst case, only materialised lines are touched
    const int64_t nl = ((G.ncells + CL_STRIDE) >> CL_SHIFT) + 1;
    const int64_t max_lines = std::min<int64_t>(nl, 2 * n + 1);
    // one fill zeroes both; last1 starts at a multiple of four ints so that the scans' 16-byte accesses stay aligned
    const int64_t nl4 = (nl + 3) & ~(int64_t)3;
    int *mark = (int *)ar.get((size_t)(nl4 + nl) * 4), *last1 = mark + nl4;
    int *ids = (int *)ar.get((size_t)nl * 4), *before = (int *)ar.get((size_t)nl * 4), *l1 = (int *)ar.get((size_t)nl * 4);
    int *l2 = (int *)ar.get((size_t)(max_lines + 1) * CL_STRIDE * 4);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemsetAsync(mark, 0, (size_t)(nl4 + nl) * 4, ctx->stream));
    const int nb = (int)((n + 255) / 256), nlb = (int)((nl + 255) / 256);
    if (G.keys32) k_line_mark<unsigned><<<nb, 256, 0, ctx->stream>>>((const unsigned *)G.keys, n, mark, last1);
    else k_line_mark<unsigned long long><<<nb, 256, 0, ctx->stream>>>((const unsigned long long *)G.keys, n, mark, last1);
    if ((rc = dev_exclusive_scan<int>(ctx, ar, mark, ids, nl))) return rc;
    if ((rc = dev_exclusive_scan<int, true>(ctx, ar, last1, before, nl))) return rc;
    k_line_l1<<<nlb, 256, 0, ctx->stream>>>(mark, ids, before, nl, l1);
    if (G.keys32) k_line_fill<unsigned><<<(unsigned)((n + 256) / 256), 256, 0, ctx->stream>>>((const unsigned *)G.keys, n, l1, nl, l2);
    else k_line_fill<unsigned long long><<<(unsigned)((n + 256) / 256), 256, 0, ctx->stream>>>((const unsigned long long *)G.keys, n, l1, nl, l2);
    R3D_HIP(ctx, hipGetLastError());
    G.v = GridView{G.mn[0], G.mn[1], G.mn[2], cell, 1.0 / cell, dims[0], dims[1], dims[2], l1, l2, nullptr, sorted, idx, nullptr, nullptr, nullptr, 0.f, nullptr};
    return R3D_OK;
}

Rigid to_rigid(const double T[16]) {
    Rigid r;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.r[i * 3 + j] = T[i * 4 + j];
        r.t[i] = T[i * 4 + 3];
    }
    return r;
}
__host__ __device__ void mat4_mul(const double A[16], const double B[16], double C[16]) {
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
            t[i * 4 + j] = s;
        }
    for (int i = 0; i < 16; i++) C[i] = t[i];
}

// 3x3 SVD via Jacobi eigen-decomposition of S^T S (float64) -- only used for the 3x3 Kabsch/Umeyama step.  The update of a
// registration iteration (this, umeyama_from_sums, solve6_to_matrix) runs in one thread of k_icp_step.
__host__ __device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = i == j;
    // Cyclic Jacobi with the annihilated element set to exactly zero: the off-diagonal mass then falls quadratically to below
    // 1e-30 of the diagonal in 5-7 sweeps.  (Without the explicit zero the rounding residue of every rotation, ~1e-17 of the
    // diagonal, keeps all 60 sweeps busy: 72 us per step in one GPU thread, three times the evaluation kernel of a 40 k-point
    // cloud.)
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]), scale = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off < 1e-300 || off <= 1e-30 * scale) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (fabs(A[p][q]) <= 1e-33 * scale) { A[p][q] = A[q][p] = 0; continue; }
                double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                double t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) { double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
                A[p][q] = A[q][p] = 0;
            }
    }
}
__host__ __device__ double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}
// Eigen::umeyama(src, dst, false) from the accumulated sums: returns U (4x4) mapping src -> dst
__host__ __device__ void umeyama_from_sums(const double *s /*ICP slots*/, double U[16]) {
    const double n = s[0];
    const double ps[3] = {s[2] / n, s[3] / n, s[4] / n}, pt[3] = {s[5] / n, s[6] / n, s[7] / n};
    double sigma[3][3];  // (1/n) sum (t - mt)(p - mp)^T  = E[t p^T] - mt mp^T ; slots hold p_i t_j
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) sigma[i][j] = s[8 + j * 3 + i] / n - pt[i] * ps[j];
    // SVD sigma = Us D Vs^T through the symmetric eigenproblems
    double StS[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { StS[i][j] = 0; for (int k = 0; k < 3; k++) StS[i][j] += sigma[k][i] * sigma[k][j]; }
    jacobi_eig3(StS, V);
    // eigenvalues in descending order, eigenvectors with them: the stable insertion sort of three as its compare-and-swap network
    // (0,1) (1,2) (0,1) on VALUES -- an index permutation (V[r][ord[c]]) is a dynamically indexed local array, which the compiler
    // keeps in scratch memory: several dependent memory round trips in the one thread the whole loop waits for
    double ev[3] = {StS[0][0], StS[1][1], StS[2][2]};
    double Vs[3][3], Us[3][3], sv[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vs[r][c] = V[r][c];
    auto cswap = [&](int a, int b) {   // a < b: swap when the later one is strictly larger (ties keep their order)
        if (ev[b] > ev[a]) {
            const double t = ev[a]; ev[a] = ev[b]; ev[b] = t;
            for (int r = 0; r < 3; r++) { const double u = Vs[r][a]; Vs[r][a] = Vs[r][b]; Vs[r][b] = u; }
        }
    };
    cswap(0, 1); cswap(1, 2); cswap(0, 1);
    for (int c = 0; c < 3; c++) sv[c] = sqrt(fmax(ev[c], 0.0));
    int rank = 0;   // singular values are in descending order: the deficient columns come last
    for (int c = 0; c < 3; c++) {
        double u[3] = {0, 0, 0};
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) u[r] += sigma[r][k] * Vs[k][c];
        double l = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!(l > 1e-300 && sv[c] > 1e-14 * fmax(sv[0], 1e-300))) break;
        for (int r = 0; r < 3; r++) Us[r][c] = u[r] / l;
        rank = c + 1;
    }
    if (rank == 2) {          // planar correspondences: third left vector completes a right-handed basis
        Us[0][2] = Us[1][0] * Us[2][1] - Us[2][0] * Us[1][1];
        Us[1][2] = Us[2][0] * Us[0][1] - Us[0][0] * Us[2][1];
        Us[2][2] = Us[0][0] * Us[1][1] - Us[1][0] * Us[0][1];
    } else if (rank == 0) {   // one pair, or all pairs coincident: pure translation (an SVD of the zero matrix is U = V = I)
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Us[r][c] = Vs[r][c];
    } else if (rank == 1) {   // collinear pairs: the twist about the line is undetermined; take the smallest rotation v0 -> u0
        const double v0[3] = {Vs[0][0], Vs[1][0], Vs[2][0]}, u0[3] = {Us[0][0], Us[1][0], Us[2][0]};
        const double cs = v0[0] * u0[0] + v0[1] * u0[1] + v0[2] * u0[2];
        double Rm[3][3];
        if (cs > -1 + 1e-12) {
            const double w[3] = {v0[1] * u0[2] - v0[2] * u0[1], v0[2] * u0[0] - v0[0] * u0[2], v0[0] * u0[1] - v0[1] * u0[0]};
            const double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double k2 = 0;
                    for (int k = 0; k < 3; k++) k2 += K[i][k] * K[k][j];
                    Rm[i][j] = (i == j) + K[i][j] + k2 / (1 + cs);
                }
        } else {                // opposite directions: half turn about the second right vector (orthogonal to v0)
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm[i][j] = 2 * Vs[i][1] * Vs[j][1] - (i == j);
        }
        for (int c = 1; c < 3; c++)
            for (int r = 0; r < 3; r++) Us[r][c] = Rm[r][0] * Vs[0][c] + Rm[r][1] * Vs[1][c] + Rm[r][2] * Vs[2][c];
    }
    double S[3] = {1, 1, 1};
    if (det3(Us) * det3(Vs) < 0) S[2] = -1;
    double R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { R[i][j] = 0; for (int k = 0; k < 3; k++) R[i][j] += Us[i][k] * S[k] * Vs[j][k]; }
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) U[i * 4 + j] = R[i][j];
        U[i * 4 + 3] = pt[i] - (R[i][0] * ps[0] + R[i][1] * ps[1] + R[i][2] * ps[2]);
    }
}

// SolveJacobianSystemAndObtainExtrinsicMatrix: x = LDLT(JTJ) \ (-JTr); identity when |det| < 1e-6
__host__ __device__ bool solve6_to_matrix(const double *s, double U[16]) {
    double A[6][6], b[6];
    for (int a = 0, q = 2; a < 6; a++)
        for (int c = a; c < 6; c++, q++) A[a][c] = A[c][a] = s[q];
    for (int a = 0; a < 6; a++) b[a] = -s[23 + a];
    // LDL^T
    double L[6][6], Dg[6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) L[i][j] = 0;
    double det = 1;
    for (int j = 0; j < 6; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * Dg[k];
        Dg[j] = d;
        det *= d;
        L[j][j] = 1;
        for (int i = j + 1; i < 6; i++) {
            double v = A[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * Dg[k];
            L[i][j] = d != 0 ? v / d : 0;
        }
    }
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    if (!isfinite(det) || fabs(det) < 1e-6) return false;
    double y[6], x[6];
    for (int i = 0; i < 6; i++) { y[i] = b[i]; for (int k = 0; k < i; k++) y[i] -= L[i][k] * y[k]; }
    for (int i = 0; i < 6; i++) y[i] /= Dg[i];
    for (int i = 5; i >= 0; i--) { x[i] = y[i]; for (int k = i + 1; k < 6; k++) x[i] -= L[k][i] * x[k]; }
    // TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0)
    double ca, sa, cb, sb, cc, sc;   // one range reduction per angle (the update runs in a single GPU thread)
    sincos(x[0], &sa, &ca);
    sincos(x[1], &sb, &cb);
    sincos(x[2], &sc, &cc);
    const double R[9] = {cc * cb, cc * sb * sa - sc * ca, cc * sb * ca + sc * sa, sc * cb, sc * sb * sa + cc * ca, sc * sb * ca - cc * sa,
                         -sb, cb * sa, cb * ca};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) U[i * 4 + j] = R[i * 3 + j];
        U[i * 4 + 3] = x[3 + i];
    }
    return true;
}

// One step of the registration loop on the device.  16 waves: wave w reduces slots w and w + 16 (lane t adds the partials
// t, t+64, ... in order, then a fixed shuffle tree => deterministic); thread 0 then does what the host loop of
// RegistrationICP does between two evaluations: statistics, convergence test against the previous evaluation, update
// (Umeyama or the 6x6 Gauss-Newton system) and T <- U T.
// `mirror` (pinned host memory, device-visible): thread 0 publishes the state there after every step, fields first, then a
// system-scope fence, then `seq`; the host polls `seq` / `done` in that memory instead of waiting for a copy and an event -- the
// completion of a D2H copy + marker reached the host 16-57 ms late about once in 16 registrations of 1 M points (the device clock
// in the state showed the device done on time: R3D_ICP_DEBUG), a kernel's own store does not go that way.
__device__ __forceinline__ void icp_publish(const IcpState *st, IcpState *mirror) {
    if (!mirror) return;
    for (int i = 0; i < 16; i++) mirror->T[i] = st->T[i];
    mirror->fit = st->fit; mirror->rmse = st->rmse; mirror->corr = st->corr;
    mirror->evals = st->evals; mirror->converged = st->converged; mirror->iterations = st->iterations;
    mirror->t_first = st->t_first; mirror->t_last = st->t_last;
    mirror->done = st->done;
    __threadfence_system();
    // the termination travels in the released word itself: a plain `done` store could become visible before the statistics
    // above, and the host would copy a stale iteration count / fitness (the fields share one 64-byte line)
    *(volatile int *)&mirror->seq = st->evals | (st->done ? ICP_SEQ_DONE : 0);
}
__global__ void __launch_bounds__(1024) k_icp_step(const double *__restrict__ partial, int nblocks, IcpState *__restrict__ st, int64_t ns,
                                                   int mode, int max_it, double rel_fit, double rel_rmse, IcpState *__restrict__ mirror,
                                                   int publish /* this step is one the host waits for (the last step always is) */) {
    __shared__ double sums[32];
    if (st->done) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the state the one deciding thread needs, requested BEFORE the reduction so that its round trips run under the partial sums'
    // (they were a chain of dependent loads after the barrier)
    int k = 0;
    double fit0 = 0, rmse0 = 0, T[16];
#pragma unroll
    for (int i = 0; i < 16; i++) T[i] = 0;
    if (threadIdx.x == 0) {
        k = st->evals; fit0 = st->fit; rmse0 = st->rmse;
#pragma unroll
        for (int i = 0; i < 16; i++) T[i] = st->T[i];
    }
    {
        // wave w owns slots w and w + 16; the partial sums are stored [slot][workgroup], so lane t reads t, t + 64, ... of a
        // contiguous run, twelve requests per slot in flight (768 workgroups = one round), added in ascending order: the same
        // sum, bit for bit, as one lane walking them one dependent load at a time (that walk was 2 x 12 L2 round trips of the
        // 12 us this kernel took)
        constexpr int U = 12;
        const int s0 = w, s1 = w + 16;
        const bool two = s1 < ICP_SLOTS;
        const double *p0 = partial + (size_t)s0 * nblocks, *p1 = partial + (size_t)(two ? s1 : s0) * nblocks;
        double v0 = 0, v1 = 0;
        for (int b0 = 0; b0 < nblocks; b0 += 64 * U) {
            double a0[U], a1[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const int b = b0 + j * 64 + lane, bc = min(b, nblocks - 1);   // clamped address + select: no branch around a load
                const double x0 = p0[bc], x1 = p1[bc];
                a0[j] = b < nblocks ? x0 : 0.0;
                a1[j] = b < nblocks ? x1 : 0.0;
            }
#pragma unroll
            for (int j = 0; j < U; j++) { v0 += a0[j]; v1 += a1[j]; }
        }
        v0 = wave_sum(v0);
        v1 = wave_sum(v1);
        if (lane == 0) { sums[s0] = v0; if (two) sums[s1] = v1; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned long long now = wall_clock64();
    if (k == 0) st->t_first = now;
    st->t_last = now;
    const double fit = sums[0] / (double)ns, rmse = sums[0] > 0 ? sqrt(sums[1] / sums[0]) : 0.0;
    int stop = 0;
    if (k >= 1 && fabs(fit0 - fit) < rel_fit && fabs(rmse0 - rmse) < rel_rmse) { st->converged = 1; st->iterations = k; stop = 1; }
    else if (k >= max_it) { st->iterations = max_it; stop = 1; }
    st->fit = fit; st->rmse = rmse; st->corr = sums[0];
    st->evals = k + 1;
    if (stop) { st->done = 1; icp_publish(st, mirror); return; }
    double U[16];
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    if (sums[0] > 0) {
        if (mode == MODE_P2P) umeyama_from_sums(sums, U);
        else solve6_to_matrix(sums, U);
    }
    mat4_mul(U, T, T);
    for (int i = 0; i < 16; i++) st->T[i] = T[i];
    if (publish) icp_publish(st, mirror);   // (a publish is ~20 posted writes to host memory and a system-scope fence: not once per step)
}

int upload(r3d_ctx *ctx, DevArena &ar, const double *h, int64_t n3, double **d) {
    *d = (double *)ar.get((size_t)n3 * 8);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(*d, h, (size_t)n3 * 8, hipMemcpyHostToDevice, ctx->stream));
    return R3D_OK;
}

}  // namespace

// ================================================================================================ C ABI
// ---- device-pointer cores shared by the host-buffer entry points and the fused device-resident chain -------------

// disparity (device) -> compacted xyz (device).  max_depth > 0 additionally drops points with |z| > max_depth.
// pose4x4 (optional) moves the points; bounds_out[6] (optional) receives their bounding box -- enqueued BEFORE the host knows the
// point count (kernels over the upper bound w * h with the count read on the device), so count and box are ONE read-back.
int reproject_core(r3d_ctx *ctx, DevArena &ar, const int16_t *d_d, int w, int h, const double *Q4x4, int min_valid_x16, double max_depth,
                   bool want_pix, double **d_xyz_out, int **d_pix_out, int64_t *m_out, const double *pose4x4 = nullptr, double *bounds_out = nullptr) {
    const int64_t n = (int64_t)w * h;
    if (n > 0x7fffffff) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "reproject_disparity: image too large");
    int *flags = (int *)ar.get((size_t)n * 4), *scan = (int *)ar.get((size_t)n * 4);
    if (ar.rc) return ar.rc;
    Mat4 Q;
    memcpy(Q.m, Q4x4, sizeof Q.m);
    const int nb = (int)((n + 255) / 256);
    if (max_depth > 0) k_disp_flags_depth<<<nb, 256, 0, ctx->stream>>>(d_d, w, n, min_valid_x16, Q, max_depth, flags);
    else k_disp_flags<<<nb, 256, 0, ctx->stream>>>(d_d, n, min_valid_x16, flags);
    if (int src = dev_exclusive_scan<int>(ctx, ar, flags, scan, n)) return src;
    int ls = 0, lf = 0;
    if (bounds_out) {
        // arrays sized for the upper bound (every pixel valid); reprojection, pose and bounding box enqueued without a host read
        double *d_xyz = (double *)ar.get((size_t)n * 24);
        int *d_pix = want_pix ? (int *)ar.get((size_t)n * 4) : nullptr;
        if (ar.rc) return ar.rc;
        k_reproject<<<nb, 256, 0, ctx->stream>>>(d_d, flags, scan, w, n, Q, d_xyz, d_pix);
        if (pose4x4) {
            double *d_t = (double *)ar.get((size_t)n * 24);
            if (ar.rc) return ar.rc;
            k_transform_dn<<<nb, 256, 0, ctx->stream>>>(d_xyz, scan + (n - 1), flags + (n - 1), to_rigid(pose4x4), d_t);
            d_xyz = d_t;
        }
        const double *d_box = nullptr;
        if (int brc = cloud_bbox_enqueue_dn(ctx, ar, d_xyz, n, scan + (n - 1), flags + (n - 1), &d_box)) return brc;
        {
            PinRead rd(ctx);
            int prc;
            if ((prc = rd.add(&ls, scan + (n - 1), 4)) || (prc = rd.add(&lf, flags + (n - 1), 4)) || (prc = rd.add(bounds_out, d_box, 48)) || (prc = rd.wait())) return prc;
        }
        const int64_t m = (int64_t)ls + lf;
        *m_out = m;
        *d_xyz_out = m ? d_xyz : nullptr;
        if (d_pix_out) *d_pix_out = m ? d_pix : nullptr;
        return m ? bbox_check(ctx, bounds_out, bounds_out + 3) : R3D_OK;
    }
    {
        PinRead rd(ctx);
        int prc;
        if ((prc = rd.add(&ls, scan + (n - 1), 4)) || (prc = rd.add(&lf, flags + (n - 1), 4)) || (prc = rd.wait())) return prc;
    }
    const int64_t m = (int64_t)ls + lf;
    *m_out = m;
    *d_xyz_out = nullptr;
    if (d_pix_out) *d_pix_out = nullptr;
    if (m == 0) return R3D_OK;
    double *d_xyz = (double *)ar.get((size_t)m * 24);
    int *d_pix = want_pix ? (int *)ar.get((size_t)m * 4) : nullptr;
    if (ar.rc) return ar.rc;
    k_reproject<<<nb, 256, 0, ctx->stream>>>(d_d, flags, scan, w, n, Q, d_xyz, d_pix);
    R3D_HIP(ctx, hipGetLastError());
    *d_xyz_out = d_xyz;
    if (d_pix_out) *d_pix_out = d_pix;
    return R3D_OK;
}

// depth image (host) -> compacted xyz / colours (device); everything below r3d_backproject_depth's argument checks
int backproject_core(r3d_ctx *ctx, DevArena &ar, const uint16_t *depth, int w, int h, int stride, const r3d_depth_camera *cam,
                     const uint8_t *color, int cstride, bool want_pix, double **d_xyz_out, double **d_rgb_out, int **d_pix_out, int64_t *m_out,
                     double *bounds_out = nullptr /* [6]: the cloud's bounding box, read back together with the count */,
                     const PinExtra *extra = nullptr /* rides along in that read-back */) {
    const int64_t n = (int64_t)w * h;
    if (n > 0x7fffffff) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "backproject_depth: image too large");
    unsigned short *d_d = (unsigned short *)ar.get((size_t)stride * h * 2);
    unsigned char *d_c = color ? (unsigned char *)ar.get((size_t)cstride * h) : nullptr;
    int *flags = (int *)ar.get((size_t)n * 4), *scan = (int *)ar.get((size_t)n * 4);
    if (ar.rc) return ar.rc;
    // the caller's last row need not be padded to the stride: (h-1) full strides + w elements is all that is guaranteed readable
    R3D_HIP(ctx, hipMemcpyAsync(d_d, depth, ((size_t)stride * (h - 1) + w) * 2, hipMemcpyHostToDevice, ctx->stream));
    if (color) R3D_HIP(ctx, hipMemcpyAsync(d_c, color, (size_t)cstride * (h - 1) + (size_t)w * 3, hipMemcpyHostToDevice, ctx->stream));
    DepthCam c{cam->fx, cam->fy, cam->ppx, cam->ppy, cam->depth_trunc, (float)cam->depth_scale, cam->flip_yz};
    const int nb = (int)((n + 255) / 256);
    k_depth_flags<<<nb, 256, 0, ctx->stream>>>(d_d, w, stride, n, c, flags);
    if (int src = dev_exclusive_scan<int>(ctx, ar, flags, scan, n)) return src;
    int ls = 0, lf = 0;
    if (bounds_out) {   // as reproject_core: arrays for the upper bound, back-projection and bounding box enqueued, ONE read-back
        double *d_xyz = (double *)ar.get((size_t)n * 24), *d_rgb = color ? (double *)ar.get((size_t)n * 24) : nullptr;
        int *d_pix = want_pix ? (int *)ar.get((size_t)n * 4) : nullptr;
        if (ar.rc) return ar.rc;
        k_backproject<<<nb, 256, 0, ctx->stream>>>(d_d, d_c, flags, scan, w, stride, cstride, n, c, d_xyz, d_rgb, d_pix);
        const double *d_box = nullptr;
        if (int brc = cloud_bbox_enqueue_dn(ctx, ar, d_xyz, n, scan + (n - 1), flags + (n - 1), &d_box)) return brc;
        {
            PinRead rd(ctx);
            int prc;
            if ((prc = rd.add(&ls, scan + (n - 1), 4)) || (prc = rd.add(&lf, flags + (n - 1), 4)) || (prc = rd.add(bounds_out, d_box, 48)) ||
                (extra && extra->host && (prc = rd.add(extra->host, extra->dev, extra->bytes))) || (prc = rd.wait())) return prc;
        }
        const int64_t m = (int64_t)ls + lf;
        *m_out = m;
        *d_xyz_out = m ? d_xyz : nullptr;
        if (d_rgb_out) *d_rgb_out = m ? d_rgb : nullptr;
        if (d_pix_out) *d_pix_out = m ? d_pix : nullptr;
        return m ? bbox_check(ctx, bounds_out, bounds_out + 3) : R3D_OK;
    }
    {
        PinRead rd(ctx);
        int prc;
        if ((prc = rd.add(&ls, scan + (n - 1), 4)) || (prc = rd.add(&lf, flags + (n - 1), 4)) || (prc = rd.wait())) return prc;
    }
    const int64_t m = (int64_t)ls + lf;
    *m_out = m;
    *d_xyz_out = nullptr;
    if (d_rgb_out) *d_rgb_out = nullptr;
    if (d_pix_out) *d_pix_out = nullptr;
    if (m == 0) return R3D_OK;
    double *d_xyz = (double *)ar.get((size_t)m * 24), *d_rgb = color ? (double *)ar.get((size_t)m * 24) : nullptr;
    int *d_pix = want_pix ? (int *)ar.get((size_t)m * 4) : nullptr;
    if (ar.rc) return ar.rc;
    k_backproject<<<nb, 256, 0, ctx->stream>>>(d_d, d_c, flags, scan, w, stride, cstride, n, c, d_xyz, d_rgb, d_pix);
    R3D_HIP(ctx, hipGetLastError());
    *d_xyz_out = d_xyz;
    if (d_rgb_out) *d_rgb_out = d_rgb;
    if (d_pix_out) *d_pix_out = d_pix;
    return R3D_OK;
}
static int depth_args_ok(r3d_ctx *ctx, const uint16_t *depth, int w, int h, int stride, const r3d_depth_camera *cam, const uint8_t *color,
                         int cstride, const char *who) {
    if (!depth || !cam || w <= 0 || h <= 0 || stride < w || (color && cstride < 3 * w)) return r3d_fail(ctx, R3D_E_BADARG, "%s: bad argument", who);
    if (!(cam->fx != 0) || !(cam->fy != 0) || !(cam->depth_scale > 0) || !(cam->depth_trunc > 0))
        return r3d_fail(ctx, R3D_E_BADARG, "%s: focal lengths must be non-zero, depth_scale and depth_trunc positive", who);
    return R3D_OK;
}

// voxel grid of a device cloud: segments of equal legacy voxel index, in lexicographic index order
struct VoxelSegs {
    int *idx = nullptr, *starts = nullptr;
    double *sorted = nullptr;   // the points in (voxel, index) order (written by the sort)
    int64_t nseg = 0;
};
// fixed_org / fixed_max (legacy grid only): use this origin and size the key space for points up to fixed_max instead of deriving
// both from the cloud's own bounding box (the incremental model table: a slice of the model keyed in the MODEL's grid; no
// bounding-box pass).  bounds_out[6] receives min / max of the cloud when its box was computed here.
// known_bounds[6] (min, max of exactly these points, already on the host): no bounding-box pass, same grid.
int voxel_segments(r3d_ctx *ctx, DevArena &ar, const double *d_p, int64_t n, double voxel, VoxelSegs &V, bool tensor_grid = false,
                   const double *fixed_org = nullptr, const double *fixed_max = nullptr, double *bounds_out = nullptr,
                   const double *known_bounds = nullptr) {
    double mn[3], mx[3];
    int rc;
    if (fixed_org) {
        for (int a = 0; a < 3; a++) { mn[a] = fixed_org[a] + 0.5 * voxel; mx[a] = fixed_max[a]; }
    } else if (known_bounds) {
        for (int a = 0; a < 3; a++) { mn[a] = known_bounds[a]; mx[a] = known_bounds[3 + a]; }
        if (bounds_out) for (int a = 0; a < 6; a++) bounds_out[a] = known_bounds[a];
    } else {
        if ((rc = cloud_bbox(ctx, ar, d_p, n, mn, mx))) return rc;
        if (bounds_out) for (int a = 0; a < 3; a++) { bounds_out[a] = mn[a]; bounds_out[3 + a] = mx[a]; }
    }
    double org[3];
    int dims[3];
    double total = 1;
    for (int a = 0; a < 3; a++) {
        if (tensor_grid) {   // o3d.t: key = floor(float(p) / float(voxel)), origin 0; float division by a positive constant is monotonic
            const double k0 = std::floor((float)mn[a] / (float)voxel), k1 = std::floor((float)mx[a] / (float)voxel);
            if (!(std::fabs(k0) < 9e15 && std::fabs(k1) < 9e15 && k1 - k0 < 2.0e9)) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "voxel_downsample: voxel grid too fine for the extent of the cloud");
            org[a] = k0;
            dims[a] = (int)(k1 - k0) + 1;
        } else {
            org[a] = fixed_org ? fixed_org[a] : mn[a] - 0.5 * voxel;  // legacy: voxel_min_bound = min_bound - voxel_size * 0.5
            dims[a] = (int)std::floor((mx[a] - org[a]) / voxel) + 2;
        }
        total *= dims[a];
    }
    if (total >= 1.8e19) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "voxel_downsample: voxel grid exceeds 2^64 cells");
    void *keys;
    bool keys32;
    // key_order 1: exact legacy index floor((p - origin) / voxel), z fastest => output in lexicographic (kx,ky,kz) order
    V.sorted = (double *)ar.get((size_t)n * 24);
    if (ar.rc) return ar.rc;
    // The sort's own round trip (run count + fallback test) is deferred and travels with the segment count: ONE read-back for the
    // whole voxel grid.  If the test then says the counting sort had to decline (rare: thousands of runs in one bucket), the
    // arrays are garbage: sort again with the library radix sort and repeat the segmentation.
    int *flags = (int *)ar.get((size_t)n * 4), *scan = (int *)ar.get((size_t)n * 4);
    V.starts = (int *)ar.get((size_t)n * 4);
    if (ar.rc) return ar.rc;
    const int nb = (int)((n + 255) / 256);
    for (int attempt = 0; attempt < 2; attempt++) {
        SortDefer df;
        if ((rc = sort_by_cell(ctx, ar, d_p, n, org, voxel, dims, tensor_grid ? 3 : 1, &keys, &keys32, &V.idx, V.sorted, attempt ? 1 : -1,
                               attempt ? nullptr : &df))) return rc;
        if (keys32) k_seg_flags<unsigned><<<nb, 256, 0, ctx->stream>>>((const unsigned *)keys, n, flags);
        else k_seg_flags<unsigned long long><<<nb, 256, 0, ctx->stream>>>((const unsigned long long *)keys, n, flags);
        if (int src = dev_exclusive_scan<int>(ctx, ar, flags, scan, n)) return src;
        k_seg_starts<<<nb, 256, 0, ctx->stream>>>(flags, scan, n, V.starts);
        int last_scan = 0, last_flag = 0;
        {
            PinRead rd(ctx);
            int prc;
            if ((prc = rd.add(&last_scan, scan + (n - 1), 4)) || (prc = rd.add(&last_flag, flags + (n - 1), 4)) || (prc = df.add_to(rd)) || (prc = rd.wait())) return prc;
        }
        V.nseg = (int64_t)last_scan + last_flag;
        if (df.ok()) return R3D_OK;
        static const bool dbg = [] { const char *e = getenv("R3D_SORT_DEBUG"); return e && *e == '1'; }();
        if (dbg) fprintf(stderr, "[r3d sort] voxel grid: deferred test failed (n=%lld runs=%d max_runs_per_bucket=%d): sorting again with the radix sort\n",
                         (long long)n, (int)(df.h_total >> 32), df.h_maxruns);
    }
    return r3d_fail(ctx, R3D_E_HIP, "voxel grid: the fallback sort did not complete");
}

// normals of a device cloud into a fresh device buffer
int normals_core(r3d_ctx *ctx, DevArena &ar, const double *d_p, int64_t n, double radius, int max_nn, const double *d_prev, double **d_n_out,
                 const double *enclosing = nullptr) {
    Grid G;
    const int k = (int)std::min<int64_t>(max_nn, n);
    int rc;
    double occ = std::max(2.0, k / 5.0);
    if (const char *oe = getenv("R3D_KNN_OCC")) { const double v = atof(oe); if (v >= 0.5 && v <= 256) occ = v; }
    if ((rc = grid_build(ctx, ar, d_p, n, radius, occ, G, false, enclosing))) return rc;
    double *d_n = (double *)ar.get((size_t)n * 24);
    if (ar.rc) return ar.rc;
    const size_t lds = (size_t)k * KNN_BLOCK * 12;
    R3D_HIP(ctx, hipFuncSetAttribute((const void *)k_normals, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_normals<<<(unsigned)((n + KNN_BLOCK - 1) / KNN_BLOCK), KNN_BLOCK, lds, ctx->stream>>>(G.v, n, k, radius, d_prev, d_n, nullptr);
    R3D_HIP(ctx, hipGetLastError());
    *d_n_out = d_n;
    return R3D_OK;
}

// ICP / point-to-plane / GICP loop on clouds that are already in device memory (d_sn / d_tn may be null where the mode
// allows it); everything below r3d_icp's argument checks and uploads
static int icp_core(r3d_ctx *ctx, DevArena &ar, const r3d_icp_params *p, double *d_s, int64_t ns, double *d_sn, double *d_t, int64_t nt,
                    double *d_tn, const double *init4x4, double *T4x4, r3d_icp_stats *stats,
                    std::chrono::steady_clock::time_point t_begin, const double *tgt_enclosing = nullptr /* grid_build's `enclosing` */,
                    double tgt_spacing = 0 /* grid_build's `spacing_hint`: voxel size the target was down-sampled with */) {
    int rc;
    Grid G;
    double occ = 3.0;   // points per occupied cell the search grid aims at (R3D_ICP_OCC: A/B)
    if (const char *oe = getenv("R3D_ICP_OCC")) { const double v = atof(oe); if (v >= 0.5 && v <= 64) occ = v; }
    if ((rc = grid_build(ctx, ar, d_t, nt, p->max_correspondence_distance, occ, G, true, tgt_enclosing, tgt_spacing))) return rc;   // dense table: one load per row in the loop
    // float32 copy for the two-stage search (R3D_ICP_IMPL=exact: all-float64 search, for A/B).  fe = 2 x bound on
    // |float distance - exact distance|: both end points are rounded to float (relative 2^-24 per coordinate), times a
    // safety factor of 2; skipped (exact search) when the coordinates are so large that the margin stops filtering.
    {
        const char *ie = getenv("R3D_ICP_IMPL");
        double M = 0;
        for (int a = 0; a < 3; a++) M = std::max(M, std::max(std::fabs(G.mn[a]), std::fabs(G.mx[a])));
        const double Mq = M + 2.0 * p->max_correspondence_distance + G.v.cell;
        const double e = std::sqrt(3.0) * (M + Mq) * std::ldexp(1.0, -24);
        const bool want_f32 = ie && strcmp(ie, "f32") == 0, want_exact = ie && (strcmp(ie, "exact") == 0 || strcmp(ie, "tiled") == 0);
        // packed search (default): the cell-relative quantisation needs (p - origin) / cell exact to well below 1/1024
        if (!want_f32 && !want_exact && Mq / G.v.cell < 1e9) {
            unsigned *q = (unsigned *)ar.get((size_t)(nt + 4) * sizeof(unsigned));
            if (ar.rc) return ar.rc;
            G.v.q10 = q;
            // R3D_ICP_POOL=0: every lane walks its own row lists (rounds 2-3; A/B)
            static const bool pool_env = [] { const char *e = getenv("R3D_ICP_POOL"); return !(e && !strcmp(e, "0")); }();
            G.v.pool = pool_env ? 1 : 0;
            k_pack_q10<<<(unsigned)((nt + 4 + 255) / 256), 256, 0, ctx->stream>>>(G.v, nt, q);
            R3D_HIP(ctx, hipGetLastError());
        } else if (!want_exact && 4.0 * e < 0.25 * G.v.cell && M < 1e30) {
            float *f = (float *)ar.get((size_t)(nt + 4) * 3 * sizeof(float));
            if (ar.rc) return ar.rc;
            k_soa_f32<<<(unsigned)((nt + 4 + 255) / 256), 256, 0, ctx->stream>>>(G.v.pts, nt, f, f + (nt + 4), f + 2 * (nt + 4));
            R3D_HIP(ctx, hipGetLastError());
            G.v.fx = f; G.v.fy = f + (nt + 4); G.v.fz = f + 2 * (nt + 4);
            G.v.fe = (float)(4.0 * e);
        }
    }
    // reach bitmap (GridView::reach): where the search walks outer shells (cell < radius) a query with NO target point within the
    // radius pays for the whole (2 s + 1)^3 neighbourhood, and the bitmap lets it leave at once.  MEASURED on the 76-frame scan:
    // no effect (397 vs 397 ms of loops) -- there the walkers are queries whose nearest point lies between one cell and the radius,
    // not queries far from the model; what helped is a cell as large as the radius (grid_build's spacing_hint: 390 -> 252 ms).
    // Off by default (R3D_ICP_REACH=1 builds it: three small kernels per registration).
    {
        static const bool reach_on = [] { const char *e = getenv("R3D_ICP_REACH"); return e && !strcmp(e, "1"); }();
        const int rs = (int)std::ceil(p->max_correspondence_distance * G.v.inv_cell);   // = the kernels' smax
        if (reach_on && rs >= 2) {
            const int rnx = (G.v.nx + rs - 1) / rs, rny = (G.v.ny + rs - 1) / rs, rnz = (G.v.nz + rs - 1) / rs;
            const int64_t rn = (int64_t)rnx * rny * rnz;
            unsigned char *ra = (unsigned char *)ar.get((size_t)rn), *rb = (unsigned char *)ar.get((size_t)rn);
            if (ar.rc) return ar.rc;
            R3D_HIP(ctx, hipMemsetAsync(ra, 0, (size_t)rn, ctx->stream));
            k_reach_mark<<<(unsigned)((nt + 255) / 256), 256, 0, ctx->stream>>>(G.v, nt, rs, rnx, rny, rnz, ra);
            const unsigned nbr = (unsigned)((rn + 255) / 256);
            k_reach_dilate<<<nbr, 256, 0, ctx->stream>>>(ra, rb, rn, 1, rnx);
            k_reach_dilate<<<nbr, 256, 0, ctx->stream>>>(rb, ra, rn, rnx, rny);
            k_reach_dilate<<<nbr, 256, 0, ctx->stream>>>(ra, rb, rn, (int64_t)rnx * rny, rnz);
            R3D_HIP(ctx, hipGetLastError());
            G.v.reach = rb; G.v.rs = rs; G.v.rnx = rnx; G.v.rny = rny; G.v.rnz = rnz;
        }
    }
    double *d_tns = nullptr;
    if (d_tn) {
        d_tns = (double *)ar.get((size_t)nt * 24);
        if (ar.rc) return ar.rc;
        k_gather3<<<(unsigned)((nt + 255) / 256), 256, 0, ctx->stream>>>(d_tn, G.v.idx, nt, d_tns);
    }
    // spatially sort the source once (Morton order of the target-grid cell of its initial pose) so that neighbouring threads
    // walk the same cells; sums are order-dependent only at the 1e-16 level and stay deterministic
    double T[16];
    if (init4x4) memcpy(T, init4x4, sizeof T);
    else for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0);
    {
        double *d_s0 = (double *)ar.get((size_t)ns * 24);
        if (ar.rc) return ar.rc;
        k_transform<<<(unsigned)((ns + 255) / 256), 256, 0, ctx->stream>>>(d_s, ns, to_rigid(T), 0, d_s0);
        int dims[3] = {G.v.nx, G.v.ny, G.v.nz};
        void *sk;
        bool sk32;
        int *sidx;
        if ((rc = sort_by_cell(ctx, ar, d_s0, ns, G.mn, G.v.cell, dims, 2, &sk, &sk32, &sidx))) return rc;
        double *d_ss = (double *)ar.get((size_t)ns * 24);
        if (ar.rc) return ar.rc;
        k_gather3<<<(unsigned)((ns + 255) / 256), 256, 0, ctx->stream>>>(d_s, sidx, ns, d_ss);
        d_s = d_ss;
        if (d_sn) {
            double *d_sns = (double *)ar.get((size_t)ns * 24);
            if (ar.rc) return ar.rc;
            k_gather3<<<(unsigned)((ns + 255) / 256), 256, 0, ctx->stream>>>(d_sn, sidx, ns, d_sns);
            d_sn = d_sns;
        }
    }
    // default: per-lane search straight from global memory (L1/L2-cached gathers), two-stage (float32 top-4, exact top-3:
    // nn_block_top4; R3D_ICP_IMPL=exact keeps every distance in float64); R3D_ICP_IMPL=tiled selects the LDS-tiled
    // kernel, kept for A/B: measured 0.31 vs 0.26 ms per GICP iteration at 1M points (staging + barriers cost more than
    // the gathers they replace once the source is Morton-ordered)
    const char *impl_env = getenv("R3D_ICP_IMPL");
    const bool tiled_impl = impl_env && strcmp(impl_env, "tiled") == 0;
    // one residency of the chip (3 workgroups per CU at the kernel's register count): every wave starts at once, walks its share
    // of the queries in a grid-stride loop and pays the 29-slot reduction once
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int nblocks = tiled_impl ? (int)std::min<int64_t>((ns + 63) / 64, 8192)
                                   : (int)std::min<int64_t>((ns + ICP_BLOCK - 1) / ICP_BLOCK, (int64_t)std::max(cus, 1) * 3);
    double *d_part = (double *)ar.get((size_t)nblocks * ICP_SLOTS * 8);
    IcpState *d_st = (IcpState *)ar.get(sizeof(IcpState));
    // small source clouds (at most two workgroups per CU: every gather of the scan is a full round trip) take the search form
    // that keeps three candidate groups in flight; R3D_ICP_DEEP=0 / 1 forces either form (A/B)
    static const int deep_env = [] { const char *e = getenv("R3D_ICP_DEEP"); return !e ? -1 : atoi(e) != 0; }();
    const bool deep_search = deep_env >= 0 ? deep_env == 1 : ns <= (int64_t)ICP_BLOCK * std::max(cus, 1) * 2;
    // R3D_ICP_SPLIT=1: search and accumulation as two kernels (A/B; same partial sums bit for bit)
    static const bool split_env = [] { const char *e = getenv("R3D_ICP_SPLIT"); return e && !strcmp(e, "1"); }();
    const bool split_impl = split_env && !tiled_impl;
    int *d_nn = split_impl ? (int *)ar.get((size_t)ns * 4) : nullptr;
    if (ar.rc) return ar.rc;
    static_assert(sizeof(IcpState) <= ICP_SLOTS * sizeof(double), "the pinned landing buffer holds one IcpState");
    if (!ctx->icp_ev) R3D_HIP(ctx, hipEventCreateWithFlags(&ctx->icp_ev, hipEventDisableTiming));
    if (!ctx->icp_host) R3D_HIP(ctx, hipHostMalloc((void **)&ctx->icp_host, ICP_SLOTS * sizeof(double), hipHostMallocMapped));
    IcpState *hst = (IcpState *)ctx->icp_host;   // pinned and mapped: k_icp_step publishes the state here, the host polls it
    // R3D_ICP_WAIT=event: the round-1/2 form (D2H copy of the state + an event polled with hipEventQuery), kept for A/B
    static const bool wait_event = [] { const char *e = getenv("R3D_ICP_WAIT"); return e && !strcmp(e, "event"); }();
    IcpState *d_mirror = nullptr;
    if (!wait_event) R3D_HIP(ctx, hipHostGetDevicePointer((void **)&d_mirror, hst, 0));
    memset(hst, 0, sizeof *hst);
    memcpy(hst->T, T, sizeof T);
    R3D_HIP(ctx, hipMemcpyAsync(d_st, hst, sizeof *hst, hipMemcpyHostToDevice, ctx->stream));
    const int max_it = p->max_iteration;
    // one evaluation + one step of the loop, enqueued without waiting (both return at once when the loop has ended)
    auto enqueue_eval = [&](bool publish) {
        const double eps = p->gicp_epsilon > 0 ? p->gicp_epsilon : 1e-3;
        const double md = p->max_correspondence_distance;
        if (tiled_impl) {
            switch (p->mode) {
                case MODE_P2P: k_icp_eval_t<MODE_P2P><<<nblocks, 64, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, md, eps, d_part, nullptr); break;
                case MODE_P2PLANE: k_icp_eval_t<MODE_P2PLANE><<<nblocks, 64, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, md, eps, d_part, nullptr); break;
                default: k_icp_eval_t<MODE_GICP><<<nblocks, 64, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, md, eps, d_part, nullptr); break;
            }
        } else {
#define R3D_ICP_LAUNCH(M, S) k_icp_eval<M, S><<<nblocks, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, md, eps, d_part, nullptr)
#define R3D_ICP_MODES(S)                                            \
    switch (p->mode) {                                              \
        case MODE_P2P: R3D_ICP_LAUNCH(MODE_P2P, S); break;          \
        case MODE_P2PLANE: R3D_ICP_LAUNCH(MODE_P2PLANE, S); break;  \
        default: R3D_ICP_LAUNCH(MODE_GICP, S); break;               \
    }
            if (split_impl) {
                // the search writes one slot per query, so its grid is free: one residency at ITS occupancy (4 workgroups per CU)
                const int nsearch = (int)std::min<int64_t>((ns + ICP_BLOCK - 1) / ICP_BLOCK, (int64_t)std::max(cus, 1) * 4);
                if (G.v.q10) k_icp_search<SEARCH_Q10><<<nsearch, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, ns, d_st, md, d_nn);
                else if (G.v.fx) k_icp_search<SEARCH_F32><<<nsearch, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, ns, d_st, md, d_nn);
                else k_icp_search<SEARCH_EXACT><<<nsearch, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, ns, d_st, md, d_nn);
                switch (p->mode) {
                    case MODE_P2P: k_icp_accum<MODE_P2P><<<nblocks, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, eps, d_nn, d_part); break;
                    case MODE_P2PLANE: k_icp_accum<MODE_P2PLANE><<<nblocks, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, eps, d_nn, d_part); break;
                    default: k_icp_accum<MODE_GICP><<<nblocks, ICP_BLOCK, 0, ctx->stream>>>(G.v, d_s, d_sn, d_tns, ns, d_st, eps, d_nn, d_part); break;
                }
            } else
            if (G.v.q10 && deep_search) { R3D_ICP_MODES(SEARCH_Q10_DEEP) }
            else if (G.v.q10) { R3D_ICP_MODES(SEARCH_Q10) }
            else if (G.v.fx) { R3D_ICP_MODES(SEARCH_F32) }
            else { R3D_ICP_MODES(SEARCH_EXACT) }
#undef R3D_ICP_MODES
#undef R3D_ICP_LAUNCH
        }
        k_icp_step<<<1, 1024, 0, ctx->stream>>>(d_part, nblocks, d_st, ns, p->mode, max_it, p->relative_fitness, p->relative_rmse, d_mirror,
                                                publish ? 1 : 0);
    };
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t_loop = std::chrono::steady_clock::now();
    // The loop of RegistrationICP (evaluate, test, update) runs on the device; the host enqueues ICP_BATCH evaluations at a
    // time and reads the state back once per batch (a host round trip per iteration cost ~50 us of a 0.22 ms iteration).
    // Polling an event instead of a blocking stream synchronise: the blocking wait may put the thread to sleep for a
    // scheduler tick (occasional 40-50 ms waits).
    // A WINDOW of evaluations stays enqueued: ICP_BATCH at first, then ICP_STRIDE more whenever all but ICP_STRIDE of them have
    // been consumed -- the device never runs dry while the host notices a batch's end and enqueues the next (that hand-over
    // left the GPU idle for ~40 us per 8 iterations: profiles/r04_icp_timeline.txt).  k_icp_step publishes its state to the
    // host mirror only at the steps the host waits for (every ICP_STRIDE-th and the last).
    // Large clouds (an evaluation is >= 50 us) get a window of 24, topped up by 8: the default 20- and 30-iteration loops are then
    // enqueued (almost) whole and do not depend on the host thread being scheduled in time -- the 10-50 ms "device queue not
    // served" loops of shared boxes were the device running dry behind a descheduled host; evaluations behind the step that ends
    // the loop return at their first instruction (at most 24 x ~8 us).  R3D_ICP_WINDOW=<batch> overrides (stride = batch / 2).
    static const int win_env = [] { const char *e = getenv("R3D_ICP_WINDOW"); const int v = e ? atoi(e) : 0; return v < 2 ? 0 : (v > 256 ? 256 : v); }();
    const bool big = ns >= 500000;
    const int ICP_BATCH = win_env ? win_env : big ? 24 : 8, ICP_STRIDE = win_env ? std::max(win_env / 2, 1) : big ? 8 : 4;
    bool loop_done = false;
    const int total = max_it + 1;
    for (int enq = 0; enq < total;) {
        for (int b = 0, nb = enq == 0 ? ICP_BATCH : ICP_STRIDE; b < nb && enq < total; b++, enq++) {
            // a step publishes its state to the host mirror (~20 posted writes and a system-scope fence) only if the host will WAIT
            // for it: the last one, and the steps `need` below names (ICP_BATCH - ICP_STRIDE, then every ICP_STRIDE-th, while more
            // evaluations remain to be enqueued behind them); a step that ENDS the loop publishes by itself
            const int sno = enq + 1, first_need = ICP_BATCH - ICP_STRIDE;
            const bool awaited = sno == total || (sno >= first_need && (sno - first_need) % ICP_STRIDE == 0 && sno + ICP_STRIDE < total);
            enqueue_eval(wait_event || awaited);
        }
        R3D_HIP(ctx, hipGetLastError());
        if (wait_event) {
            R3D_HIP(ctx, hipMemcpyAsync(hst, d_st, sizeof *hst, hipMemcpyDeviceToHost, ctx->stream));
            R3D_HIP(ctx, hipEventRecord(ctx->icp_ev, ctx->stream));
        }
        // evaluations that must have been consumed before the host goes on: everything at the end (and in event mode, where
        // the copy is ordered behind the whole batch), all but the last ICP_STRIDE otherwise
        const int need = (enq >= total || wait_event) ? enq : enq - ICP_STRIDE;
        // Bounded wait: either the state k_icp_step publishes in the mapped host buffer (`seq` reaches `need`, or the DONE bit),
        // or (R3D_ICP_WAIT=event) the event behind a D2H copy of the state.  A tight spin for
        // the first 200 us (a batch of a small cloud is shorter than a sleep), then the thread yields between polls (a bare
        // hipEventQuery spin hammers the runtime's stream lock, which a profiler's completion handlers also need).  The deadline
        // scales with the work enqueued (serialised counter passes are ~100x slower than a plain run); on expiry the call fails
        // with the last state read so the caller exits non-zero.
        const auto t_poll = std::chrono::steady_clock::now();
        const double deadline_s = 20.0 + 2e-6 * (double)(ns + nt) * ICP_BATCH;
        for (unsigned spins = 0;; spins++) {
            if (wait_event) {
                const hipError_t q = hipEventQuery(ctx->icp_ev);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return r3d_fail(ctx, R3D_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
            } else {
                // `seq` is the last word k_icp_step stores, behind a system-scope fence: once it shows the awaited step, or the
                // DONE bit, every other field of that step is visible
                const int seq = *(volatile int *)&hst->seq;
                std::atomic_thread_fence(std::memory_order_acquire);
                if ((seq & ~ICP_SEQ_DONE) >= need || (seq & ICP_SEQ_DONE)) { loop_done = (seq & ICP_SEQ_DONE) != 0; break; }
            }
            if ((spins & 63) != 63) continue;
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count();
            if (waited > deadline_s) {
                ctx->poisoned = true;   // the batch is still queued on arena buffers: no later call may reuse them
                return r3d_fail(ctx, R3D_E_HIP, "registration loop: no completion after %.0f s (%d evaluations enqueued, last state read: %d evaluations, done=%d)",
                                waited, enq, hst->evals, hst->done);
            }
            if (waited > 2e-4) std::this_thread::sleep_for(std::chrono::microseconds(waited > 5e-3 ? 200 : 20));
        }
        if (wait_event) loop_done = hst->done != 0;   // (the copy behind the event is complete: all fields are of one step)
        if (loop_done) break;
    }
    if (!loop_done) return r3d_fail(ctx, R3D_E_HIP, "registration loop did not finish (%d evaluations)", hst->evals);
    // (in mirror mode evaluations enqueued behind the step that ended the loop may still be in flight: they return at their first
    // instruction (`done`), and everything this call or the next enqueues is ordered behind them on the same stream)
    memcpy(T, hst->T, sizeof T);
    const int it = hst->iterations, converged = hst->converged;
    const double fit = hst->fit, rmse = hst->rmse, ncorr = hst->corr;
    memcpy(T4x4, T, sizeof T);
    if (stats) {
        stats->iterations = it;
        stats->converged = converged;
        stats->correspondences = (int64_t)ncorr;
        stats->fitness = fit;
        stats->inlier_rmse = rmse;
        const auto t_end = std::chrono::steady_clock::now();
        stats->setup_ms = std::chrono::duration<double, std::milli>(t_loop - t_begin).count();
        stats->loop_ms = std::chrono::duration<double, std::milli>(t_end - t_loop).count();
    }
    {
        static const bool dbg = [] { const char *e = getenv("R3D_ICP_DEBUG"); return e && *e == '1'; }();
        if (dbg) {
            const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count();
            const double dev_ms = (double)(hst->t_last - hst->t_first) * 1e-5;   // 100 MHz ticks, first step to last step
            if (host_ms > 10.0 || host_ms > 5.0 + 3.0 * dev_ms)
                fprintf(stderr, "[r3d icp] loop: host %.3f ms, device first-to-last step %.3f ms (%d evaluations): the device %s\n", host_ms, dev_ms,
                        hst->evals, dev_ms > 0.5 * host_ms ? "itself was stalled" : "finished on time, the completion reached the host late");
        }
    }
    return R3D_OK;
}

extern "C" {

static int voxel_downsample_impl(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                                 double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n, bool tensor_grid);

int r3d_voxel_downsample(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                         double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_voxel_downsample");
    return voxel_downsample_impl(ctx, xyz, colors, normals, n, voxel, out_xyz, out_colors, out_normals, out_n, false);
}

int r3d_voxel_downsample_tensor(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                                double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_voxel_downsample_tensor");
    return voxel_downsample_impl(ctx, xyz, colors, normals, n, voxel, out_xyz, out_colors, out_normals, out_n, true);
}

static int voxel_downsample_impl(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                                 double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n, bool tensor_grid) {
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !out_xyz || !out_n || n <= 0 || !(voxel > 0)) return r3d_fail(ctx, R3D_E_BADARG, "voxel_downsample: bad argument");
    if ((colors && !out_colors) || (normals && !out_normals)) return r3d_fail(ctx, R3D_E_BADARG, "voxel_downsample: missing output array");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p, *d_c = nullptr, *d_n = nullptr;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    if (colors && (rc = upload(ctx, ar, colors, n * 3, &d_c))) return rc;
    if (normals && (rc = upload(ctx, ar, normals, n * 3, &d_n))) return rc;
    VoxelSegs V;
    if ((rc = voxel_segments(ctx, ar, d_p, n, voxel, V, tensor_grid))) return rc;
    const int64_t nseg = V.nseg;
    double *d_out = (double *)ar.get((size_t)nseg * 24);
    if (ar.rc) return ar.rc;
    const double *ins[3] = {d_p, d_c, d_n};
    double *outs[3] = {out_xyz, out_colors, out_normals};
    for (int a = 0; a < 3; a++) {
        if (!ins[a]) continue;
        if ((rc = launch_voxel_mean(ctx, ar, tensor_grid, ins[a], V.idx, V.starts, nseg, n, d_out, a == 0 ? V.sorted : nullptr))) return rc;
        R3D_HIP(ctx, hipGetLastError());
        R3D_HIP(ctx, hipMemcpyAsync(outs[a], d_out, (size_t)nseg * 24, hipMemcpyDeviceToHost, ctx->stream));
        R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out_n = nseg;
    return R3D_OK;
}

int r3d_estimate_normals(r3d_ctx *ctx, const double *xyz, int64_t n, double radius, int32_t max_nn, const double *prev_normals,
                         double *normals) {
    R3D_ROCTX_RANGE("r3d_estimate_normals");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !normals || n <= 0 || max_nn < 1) return r3d_fail(ctx, R3D_E_BADARG, "estimate_normals: bad argument");
    if (max_nn > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "estimate_normals: max_nn > 128 not supported");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p, *d_prev = nullptr, *d_n;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    if (prev_normals && (rc = upload(ctx, ar, prev_normals, n * 3, &d_prev))) return rc;
    if ((rc = normals_core(ctx, ar, d_p, n, radius, max_nn, d_prev, &d_n))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(normals, d_n, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_neighbor_score(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double count_radius, double *score) {
    R3D_ROCTX_RANGE("r3d_neighbor_score");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !score || n <= 0 || (count_radius <= 0 && k < 1)) return r3d_fail(ctx, R3D_E_BADARG, "neighbor_score: bad argument");
    if (k > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "neighbor_score: k > 128 not supported");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    Grid G;
    const int kk = (int)std::min<int64_t>(std::max(k, 1), n);
    if ((rc = grid_build(ctx, ar, d_p, n, count_radius > 0 ? count_radius : -1.0, std::max(2.0, kk / 5.0), G))) return rc;
    double *d_s = (double *)ar.get((size_t)n * 8);
    if (ar.rc) return ar.rc;
    const size_t lds = (size_t)kk * KNN_BLOCK * 12;
    R3D_HIP(ctx, hipFuncSetAttribute((const void *)k_knn_score, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_knn_score<<<(unsigned)((n + KNN_BLOCK - 1) / KNN_BLOCK), KNN_BLOCK, lds, ctx->stream>>>(G.v, n, kk, count_radius, d_s);
    R3D_HIP(ctx, hipGetLastError());
    R3D_HIP(ctx, hipMemcpyAsync(score, d_s, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_reproject_disparity(r3d_ctx *ctx, const int16_t *disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                            double *out_xyz, int32_t *out_pixel, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_reproject_disparity");
    if (!ctx) return R3D_E_BADARG;
    if (!disp || !Q4x4 || !out_xyz || !out_n || w <= 0 || h <= 0) return r3d_fail(ctx, R3D_E_BADARG, "reproject_disparity: bad argument");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    const size_t n = (size_t)w * h;
    int16_t *d_d = (int16_t *)ar.get(n * 2);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(d_d, disp, n * 2, hipMemcpyHostToDevice, ctx->stream));
    double *d_xyz;
    int *d_pix;
    int64_t m;
    int rc;
    if ((rc = reproject_core(ctx, ar, d_d, w, h, Q4x4, min_valid_x16, 0.0, out_pixel != nullptr, &d_xyz, &d_pix, &m))) return rc;
    *out_n = m;
    if (m == 0) return R3D_OK;
    R3D_HIP(ctx, hipMemcpyAsync(out_xyz, d_xyz, (size_t)m * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (out_pixel) R3D_HIP(ctx, hipMemcpyAsync(out_pixel, d_pix, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

static int disparity_to_cloud_impl(r3d_ctx *ctx, const int16_t *d_disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                                   double max_depth, const double *pose4x4, double voxel, double normal_radius, int32_t max_nn,
                                   int64_t capacity, double *out_xyz, double *out_normals, int64_t *out_n, bool device_out) {
    if (!ctx) return R3D_E_BADARG;
    if (!d_disp || !Q4x4 || !out_xyz || !out_n || w <= 0 || h <= 0 || capacity < 0)
        return r3d_fail(ctx, R3D_E_BADARG, "disparity_to_cloud: bad argument");
    if (max_nn > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "disparity_to_cloud: max_nn > 128 not supported");
    if (max_nn > 0 && !out_normals) return r3d_fail(ctx, R3D_E_BADARG, "disparity_to_cloud: normals requested without an output array");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p;
    int64_t m;
    int rc;
    // depth filter off: still drop points at infinity (W = 0, e.g. disparity 0), which no voxel grid can hold
    // Host round trips of the chain (each ~25 us of idle GPU): (1) valid-pixel count + bounding box of the posed points, (2) the voxel
    // grid's segment count + its sort's fallback test, (3) the normal grid's sort -- three, was six: reprojection, pose and box are
    // enqueued over the upper bound w * h before the count is known; the voxel sort's test travels with the segment count; the
    // normal grid reuses the box (voxel means lie inside it)
    double box[6];
    if ((rc = reproject_core(ctx, ar, d_disp, w, h, Q4x4, min_valid_x16, max_depth > 0 ? max_depth : 1.0e300, false, &d_p, nullptr, &m, pose4x4, box))) return rc;
    *out_n = 0;
    if (m == 0) return R3D_OK;
    if (voxel > 0) {
        VoxelSegs V;
        if ((rc = voxel_segments(ctx, ar, d_p, m, voxel, V, false, nullptr, nullptr, nullptr, box))) return rc;
        double *d_v = (double *)ar.get((size_t)V.nseg * 24);
        if (ar.rc) return ar.rc;
        if ((rc = launch_voxel_mean(ctx, ar, false, d_p, V.idx, V.starts, V.nseg, m, d_v, V.sorted))) return rc;
        R3D_HIP(ctx, hipGetLastError());
        d_p = d_v;
        m = V.nseg;
    }
    *out_n = m;
    if (m > capacity) return r3d_fail(ctx, R3D_E_BADARG, "disparity_to_cloud: %lld points, output arrays hold %lld", (long long)m, (long long)capacity);
    double *d_n = nullptr;
    if (max_nn > 0 && (rc = normals_core(ctx, ar, d_p, m, normal_radius, max_nn, nullptr, &d_n, box))) return rc;
    const hipMemcpyKind kind = device_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    R3D_HIP(ctx, hipMemcpyAsync(out_xyz, d_p, (size_t)m * 24, kind, ctx->stream));
    if (d_n) R3D_HIP(ctx, hipMemcpyAsync(out_normals, d_n, (size_t)m * 24, kind, ctx->stream));
    // device outputs stay ordered on the context stream (the caller's next kernel or collective on that stream sees them);
    // the arena is only reused by later calls on the same stream, so no wait is needed here either
    if (!device_out) R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_disparity_to_cloud_dev(r3d_ctx *ctx, const int16_t *d_disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                               double max_depth, const double *pose4x4, double voxel, double normal_radius, int32_t max_nn,
                               int64_t capacity, double *out_xyz, double *out_normals, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_disparity_to_cloud_dev");
    return disparity_to_cloud_impl(ctx, d_disp, w, h, Q4x4, min_valid_x16, max_depth, pose4x4, voxel, normal_radius, max_nn, capacity,
                                   out_xyz, out_normals, out_n, false);
}

int r3d_disparity_to_cloud_resident(r3d_ctx *ctx, const int16_t *d_disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                                    double max_depth, const double *pose4x4, double voxel, double normal_radius, int32_t max_nn,
                                    int64_t capacity, double *d_out_xyz, double *d_out_normals, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_disparity_to_cloud_resident");
    return disparity_to_cloud_impl(ctx, d_disp, w, h, Q4x4, min_valid_x16, max_depth, pose4x4, voxel, normal_radius, max_nn, capacity,
                                   d_out_xyz, d_out_normals, out_n, true);
}

int r3d_align_point_clouds(r3d_ctx *ctx, const r3d_align_params *p, const double *src, const double *src_colors, int64_t ns,
                           const double *tgt, int64_t nt, const double *init4x4, double *out_xyz, double *out_colors,
                           double *out_normals, int64_t *out_n, double *T4x4, r3d_icp_stats *stats) {
    R3D_ROCTX_RANGE("r3d_align_point_clouds");
    if (!ctx) return R3D_E_BADARG;
    if (!p || !src || !tgt || !out_xyz || !out_n || !T4x4 || ns <= 0 || nt <= 0) return r3d_fail(ctx, R3D_E_BADARG, "align: bad argument");
    if (src_colors && !out_colors) return r3d_fail(ctx, R3D_E_BADARG, "align: colours given without an output array");
    const r3d_icp_params *ip = &p->icp;
    if (ip->mode < 0 || ip->mode > 2) return r3d_fail(ctx, R3D_E_BADARG, "align: mode must be 0 (P2P), 1 (P2PLANE) or 2 (GICP)");
    if (!(ip->max_correspondence_distance > 0)) return r3d_fail(ctx, R3D_E_BADARG, "align: max_correspondence_distance must be > 0");
    if (p->normal_max_nn > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "align: normal_max_nn > 128 not supported");
    if (ip->mode != MODE_P2P && p->normal_max_nn <= 0) return r3d_fail(ctx, R3D_E_BADARG, "align: this mode needs normals (normal_max_nn > 0)");
    if (p->normal_max_nn > 0 && !out_normals) return r3d_fail(ctx, R3D_E_BADARG, "align: normals requested without an output array");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    int rc;
    double *d_s, *d_t, *d_c = nullptr;
    if ((rc = upload(ctx, ar, src, ns * 3, &d_s))) return rc;
    if ((rc = upload(ctx, ar, tgt, nt * 3, &d_t))) return rc;
    if (src_colors && (rc = upload(ctx, ar, src_colors, ns * 3, &d_c))) return rc;
    int64_t ms = ns, mt = nt;
    double tbox[6];
    const double *tgt_box = nullptr;
    if (p->voxel_size > 0) {  // pointcloud_alignment.py:22-23
        VoxelSegs V;
        if ((rc = voxel_segments(ctx, ar, d_s, ns, p->voxel_size, V))) return rc;
        double *d_sv = (double *)ar.get((size_t)V.nseg * 24), *d_cv = d_c ? (double *)ar.get((size_t)V.nseg * 24) : nullptr;
        if (ar.rc) return ar.rc;
        if ((rc = launch_voxel_mean(ctx, ar, false, d_s, V.idx, V.starts, V.nseg, ns, d_sv, V.sorted))) return rc;
        if (d_c) if ((rc = launch_voxel_mean(ctx, ar, false, d_c, V.idx, V.starts, V.nseg, ns, d_cv))) return rc;
        R3D_HIP(ctx, hipGetLastError());
        d_s = d_sv;
        d_c = d_cv;
        ms = V.nseg;
        VoxelSegs Vt;
        if ((rc = voxel_segments(ctx, ar, d_t, nt, p->voxel_size, Vt, false, nullptr, nullptr, tbox))) return rc;
        tgt_box = tbox;   // the search grid of the down-sampled target is laid out in the box of the target's own points (as the resident model does)
        double *d_tv = (double *)ar.get((size_t)Vt.nseg * 24);
        if (ar.rc) return ar.rc;
        if ((rc = launch_voxel_mean(ctx, ar, false, d_t, Vt.idx, Vt.starts, Vt.nseg, nt, d_tv, Vt.sorted))) return rc;
        R3D_HIP(ctx, hipGetLastError());
        d_t = d_tv;
        mt = Vt.nseg;
    }
    double *d_sn = nullptr, *d_tn = nullptr;
    if (p->normal_max_nn > 0) {  // pointcloud_alignment.py:27-28
        if ((rc = normals_core(ctx, ar, d_s, ms, p->normal_radius, p->normal_max_nn, nullptr, &d_sn))) return rc;
        if ((rc = normals_core(ctx, ar, d_t, mt, p->normal_radius, p->normal_max_nn, nullptr, &d_tn))) return rc;
    }
    double T[16];
    if ((rc = icp_core(ctx, ar, ip, d_s, ms, d_sn, d_t, mt, d_tn, init4x4, T, stats, t_begin, tgt_box, p->voxel_size > 0 ? p->voxel_size : 0))) return rc;  // :35-39
    memcpy(T4x4, T, sizeof T);
    double *d_o = (double *)ar.get((size_t)ms * 24), *d_on = d_sn ? (double *)ar.get((size_t)ms * 24) : nullptr;
    if (ar.rc) return ar.rc;
    const unsigned nb = (unsigned)((ms + 255) / 256);
    k_transform<<<nb, 256, 0, ctx->stream>>>(d_s, ms, to_rigid(T), 0, d_o);  // :42 source.transform
    if (d_sn) k_transform<<<nb, 256, 0, ctx->stream>>>(d_sn, ms, to_rigid(T), 1, d_on);
    R3D_HIP(ctx, hipGetLastError());
    R3D_HIP(ctx, hipMemcpyAsync(out_xyz, d_o, (size_t)ms * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (d_sn) R3D_HIP(ctx, hipMemcpyAsync(out_normals, d_on, (size_t)ms * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (d_c) R3D_HIP(ctx, hipMemcpyAsync(out_colors, d_c, (size_t)ms * 24, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_n = ms;
    return R3D_OK;
}

int r3d_knn_graph(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double radius, int32_t *nbr, double *d2) {
    R3D_ROCTX_RANGE("r3d_knn_graph");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !nbr || n <= 0 || k < 1) return r3d_fail(ctx, R3D_E_BADARG, "knn_graph: bad argument");
    if (k > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "knn_graph: k > 128 not supported");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    Grid G;
    if ((rc = grid_build(ctx, ar, d_p, n, radius, std::max(2.0, k / 5.0), G))) return rc;
    int *d_nb = (int *)ar.get((size_t)n * k * 4);
    double *d_d2 = d2 ? (double *)ar.get((size_t)n * k * 8) : nullptr;
    if (ar.rc) return ar.rc;
    const size_t lds = (size_t)k * KNN_BLOCK * 12;
    R3D_HIP(ctx, hipFuncSetAttribute((const void *)k_knn_graph, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_knn_graph<<<(unsigned)((n + KNN_BLOCK - 1) / KNN_BLOCK), KNN_BLOCK, lds, ctx->stream>>>(G.v, n, k, radius, d_nb, d_d2);
    R3D_HIP(ctx, hipGetLastError());
    R3D_HIP(ctx, hipMemcpyAsync(nbr, d_nb, (size_t)n * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d2) R3D_HIP(ctx, hipMemcpyAsync(d2, d_d2, (size_t)n * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

}  // extern "C" (reopened below)

// orient_normals_consistent_tangent_plane(k)  (normal_estimation.py:21; Open3D legacy OrientNormalsConsistentTangentPlane
// [recalled], which the tensor method delegates to).  Device: the k-NN graph.  Host, inside the library (sequential by nature):
// two Kruskal spanning trees and the breadth-first sign propagation.  Edges are visited in the total order (weight, a, b).
//   with a Delaunay edge list (r3d_orient_normals_graph): Euclidean MST of those edges (weight = squared length), re-weighted
//   1 - |n_a . n_b|; plus the k-NN edges (the point itself counts as one of the k) that are NOT Delaunay edges -- the original
//   tests membership against the set of ALL Delaunay edges, not only the EMST ones, and so does this; spanning tree of that
//   graph; propagation from the point of maximum z (first on ties), whose normal is turned towards +z.
//   without one (r3d_orient_normals): k-NN edges only, every connected component rooted at its own highest point.
namespace {
struct WEdge { double w; int a, b; };
inline bool wedge_less(const WEdge &x, const WEdge &y) {
    if (x.w != y.w) return x.w < y.w;
    if (x.a != y.a) return x.a < y.a;
    return x.b < y.b;
}
// keeps the spanning-forest edges of `edges` (sorted in place); parent is the caller's disjoint-set array (reset here)
void kruskal_forest(std::vector<WEdge> &edges, std::vector<int> &parent, std::vector<WEdge> &forest) {
    std::sort(edges.begin(), edges.end(), wedge_less);
    const int64_t n = (int64_t)parent.size();
    for (int64_t i = 0; i < n; i++) parent[i] = (int)i;
    auto find = [&](int v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
    forest.clear();
    forest.reserve(n);
    for (const WEdge &e : edges) {
        const int ra = find(e.a), rb = find(e.b);
        if (ra == rb) continue;
        parent[ra] = rb;
        forest.push_back(e);
        if ((int64_t)forest.size() == n - 1) break;
    }
}

int orient_core(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, const int32_t *del_edges, int64_t n_del, double *normals) {
    auto dotn = [&](int a, int b) {
        return (normals[(size_t)a * 3] * normals[(size_t)b * 3] + normals[(size_t)a * 3 + 1] * normals[(size_t)b * 3 + 1]) +
               normals[(size_t)a * 3 + 2] * normals[(size_t)b * 3 + 2];
    };
    auto flip = [&](int v) { for (int a = 0; a < 3; a++) normals[(size_t)v * 3 + a] = -normals[(size_t)v * 3 + a]; };
    // the k nearest INCLUDING the point itself (KDTreeFlann::SearchKNN(points_[v0], k)), from the device
    const int kk = (int)std::min<int64_t>((int64_t)k, std::min<int64_t>(n, 128));
    std::vector<int32_t> nbr((size_t)n * kk);
    int rc = r3d_knn_graph(ctx, xyz, n, kk, -1.0, nbr.data(), nullptr);
    if (rc) return rc;
    std::vector<int> parent(n);
    std::vector<WEdge> graph, forest;
    std::vector<unsigned long long> del_keys;   // EdgeIndex = min * n + max of every Delaunay edge, sorted
    if (del_edges) {
        std::vector<WEdge> del;
        del.reserve((size_t)n_del);
        del_keys.reserve((size_t)n_del);
        for (int64_t e = 0; e < n_del; e++) {
            const int64_t u = del_edges[2 * e], v = del_edges[2 * e + 1];
            if (u < 0 || v < 0 || u >= n || v >= n) return r3d_fail(ctx, R3D_E_BADARG, "orient_normals: edge %lld names a point outside the cloud", (long long)e);
            if (u == v) continue;
            del_keys.push_back((unsigned long long)std::min(u, v) * (unsigned long long)n + (unsigned long long)std::max(u, v));
        }
        std::sort(del_keys.begin(), del_keys.end());
        del_keys.erase(std::unique(del_keys.begin(), del_keys.end()), del_keys.end());
        for (unsigned long long key : del_keys) {
            const int a = (int)(key / (unsigned long long)n), b = (int)(key % (unsigned long long)n);
            const double dx = xyz[(size_t)a * 3] - xyz[(size_t)b * 3], dy = xyz[(size_t)a * 3 + 1] - xyz[(size_t)b * 3 + 1],
                         dz = xyz[(size_t)a * 3 + 2] - xyz[(size_t)b * 3 + 2];
            del.push_back({(dx * dx + dy * dy) + dz * dz, a, b});
        }
        kruskal_forest(del, parent, forest);                                           // Euclidean minimum spanning tree
        graph.reserve(forest.size() + (size_t)n * (kk > 1 ? kk - 1 : 0));
        for (const WEdge &e : forest) graph.push_back({1.0 - std::fabs(dotn(e.a, e.b)), e.a, e.b});
    } else {
        graph.reserve((size_t)n * (kk > 1 ? kk - 1 : 0));
    }
    const size_t n_tree_edges = graph.size();
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < kk; j++) {
            const int q = nbr[(size_t)i * kk + j];
            if (q < 0 || q == i) continue;
            const int a = (int)std::min<int64_t>(i, q), b = (int)std::max<int64_t>(i, q);
            if (del_edges && std::binary_search(del_keys.begin(), del_keys.end(), (unsigned long long)a * (unsigned long long)n + (unsigned long long)b)) continue;
            graph.push_back({0.0, a, b});
        }
    {   // an undirected k-NN edge appears once (graph_edges.insert in the original); weights after the de-duplication
        auto first = graph.begin() + (std::ptrdiff_t)n_tree_edges;
        std::sort(first, graph.end(), [](const WEdge &x, const WEdge &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        graph.erase(std::unique(first, graph.end(), [](const WEdge &x, const WEdge &y) { return x.a == y.a && x.b == y.b; }), graph.end());
        for (size_t e = n_tree_edges; e < graph.size(); e++) graph[e].w = 1.0 - std::fabs(dotn(graph[e].a, graph[e].b));
    }
    kruskal_forest(graph, parent, forest);
    // adjacency of the tree (CSR)
    std::vector<int> deg(n + 1, 0);
    for (const WEdge &e : forest) { deg[e.a + 1]++; deg[e.b + 1]++; }
    for (int64_t i = 0; i < n; i++) deg[i + 1] += deg[i];
    std::vector<int> adj(forest.size() * 2), fill(deg.begin(), deg.end() - 1);
    for (const WEdge &e : forest) { adj[fill[e.a]++] = e.b; adj[fill[e.b]++] = e.a; }
    // roots: the highest point of the cloud (Delaunay graph: connected), or of every component (k-NN graph only)
    std::vector<int> roots;
    if (del_edges) {
        int top = 0;
        for (int64_t i = 1; i < n; i++) if (xyz[(size_t)i * 3 + 2] > xyz[(size_t)top * 3 + 2]) top = (int)i;
        roots.push_back(top);
    } else {
        auto find = [&](int v) { while (parent[v] != v) v = parent[v]; return v; };
        std::vector<int> comp_root(n, -1);
        for (int64_t i = 0; i < n; i++) {
            const int r = find((int)i);
            if (comp_root[r] < 0 || xyz[(size_t)i * 3 + 2] > xyz[(size_t)comp_root[r] * 3 + 2]) comp_root[r] = (int)i;
        }
        for (int64_t r = 0; r < n; r++) if (comp_root[r] >= 0) roots.push_back(comp_root[r]);
    }
    std::vector<char> seen(n, 0);
    std::vector<int> queue;
    queue.reserve(n);
    for (const int root : roots) {
        if (normals[(size_t)root * 3 + 2] < 0) flip(root);      // TestAndOrientNormal((0,0,1), n_root)
        queue.clear();
        queue.push_back(root);
        seen[root] = 1;
        for (size_t h = 0; h < queue.size(); h++) {
            const int v = queue[h];
            for (int e = deg[v]; e < deg[v + 1]; e++) {
                const int u = adj[e];
                if (seen[u]) continue;
                seen[u] = 1;
                if (dotn(u, v) < 0) flip(u);
                queue.push_back(u);
            }
        }
    }
    return R3D_OK;
}
}  // namespace

extern "C" {

int r3d_orient_normals(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double *normals) {
    R3D_ROCTX_RANGE("r3d_orient_normals");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !normals || n <= 0 || k < 1) return r3d_fail(ctx, R3D_E_BADARG, "orient_normals: bad argument");
    if (n == 1) { if (normals[2] < 0) for (int a = 0; a < 3; a++) normals[a] = -normals[a]; return R3D_OK; }
    return orient_core(ctx, xyz, n, k, nullptr, 0, normals);
}

int r3d_orient_normals_graph(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, const int32_t *delaunay_edges, int64_t n_edges,
                             double *normals) {
    R3D_ROCTX_RANGE("r3d_orient_normals_graph");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !normals || !delaunay_edges || n <= 0 || k < 1 || n_edges < 0) return r3d_fail(ctx, R3D_E_BADARG, "orient_normals_graph: bad argument");
    if (n < 4) return r3d_fail(ctx, R3D_E_BADARG, "orient_normals_graph: a tetrahedralisation needs at least 4 points");
    return orient_core(ctx, xyz, n, k, delaunay_edges, n_edges, normals);
}

int r3d_transform_points(r3d_ctx *ctx, const double *xyz, int64_t n, const double *T4x4, int32_t rotate_only, double *out) {
    R3D_ROCTX_RANGE("r3d_transform_points");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !out || !T4x4 || n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "transform_points: bad argument");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    double *d_o = (double *)ar.get((size_t)n * 24);
    if (ar.rc) return ar.rc;
    k_transform<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_p, n, to_rigid(T4x4), rotate_only, d_o);
    R3D_HIP(ctx, hipGetLastError());
    R3D_HIP(ctx, hipMemcpyAsync(out, d_o, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_icp(r3d_ctx *ctx, const r3d_icp_params *p, const double *src, int64_t ns, const double *src_normals, const double *tgt,
            int64_t nt, const double *tgt_normals, const double *init4x4, double *T4x4, r3d_icp_stats *stats) {
    R3D_ROCTX_RANGE("r3d_icp");
    if (!ctx) return R3D_E_BADARG;
    if (!p || !src || !tgt || !T4x4 || ns <= 0 || nt <= 0) return r3d_fail(ctx, R3D_E_BADARG, "icp: bad argument");
    if (p->mode < 0 || p->mode > 2) return r3d_fail(ctx, R3D_E_BADARG, "icp: mode must be 0 (P2P), 1 (P2PLANE) or 2 (GICP)");
    if (!(p->max_correspondence_distance > 0)) return r3d_fail(ctx, R3D_E_BADARG, "icp: max_correspondence_distance must be > 0");
    if (p->mode != MODE_P2P && !tgt_normals) return r3d_fail(ctx, R3D_E_BADARG, "icp: target normals required for this mode");
    if (p->mode == MODE_GICP && !src_normals) return r3d_fail(ctx, R3D_E_BADARG, "icp: source normals required for GICP");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    int rc;
    double *d_t, *d_tn = nullptr, *d_s, *d_sn = nullptr;
    if ((rc = upload(ctx, ar, tgt, nt * 3, &d_t))) return rc;
    if ((rc = upload(ctx, ar, src, ns * 3, &d_s))) return rc;
    if (src_normals && (rc = upload(ctx, ar, src_normals, ns * 3, &d_sn))) return rc;
    if (tgt_normals && (rc = upload(ctx, ar, tgt_normals, nt * 3, &d_tn))) return rc;
    return icp_core(ctx, ar, p, d_s, ns, d_sn, d_t, nt, d_tn, init4x4, T4x4, stats, t_begin);
}

// ---- resident scan-loop model (main.py:34-54, test/GICP1.py:134-155) ---------------------------------
// The reference keeps the growing model in an Open3D cloud and hands ALL of it to align_point_clouds for every frame, which
// voxel-down-samples it again (pointcloud_alignment.py:22-23).  Here the model's points / colours / normals live in device
// buffers owned by an r3d_model: a frame goes up, the 4x4 and the statistics come down, nothing else crosses PCIe until
// r3d_model_download.  Every step runs the same kernels in the same order as the one-shot entry points on the same values
// (the model is re-voxelised from its resident points, in their original order), so the results are identical to them.
}  // extern "C"
struct r3d_model {
    r3d_ctx *ctx = nullptr;
    r3d_buf pts, cols, nrms;
    int64_t n = 0;
    bool has_colors = false, has_normals = false;
    // incremental legacy voxel table of the model's points (k_vt_*): valid for voxel size vt_voxel while the model's minimum
    // corner is vt_min; covers model rows [0, vt_pts); vt_n voxels in buffers [vt_cur] (double-buffered for the merge)
    bool vt_valid = false;
    double vt_voxel = 0, vt_min[3] = {0, 0, 0}, vt_max[3] = {0, 0, 0};
    int64_t vt_n = 0, vt_pts = 0;
    r3d_buf vt_keys[2], vt_sums[2], vt_cnt[2];
    int vt_cur = 0;
    int vt_rebuilds = 0, vt_updates = 0;   // statistics (r3d_model_voxel_table_stats)
    // bounding box of the rows [pend_lo, pend_hi) the last alignment appended: computed on the device when they were appended,
    // fetched with the NEXT call's first read-back (no round trip of its own); pend_host_ok: pend_box holds it already
    double *d_pend = nullptr;
    int64_t pend_lo = 0, pend_hi = 0;
    bool pend_host_ok = false;
    double pend_box[6] = {0, 0, 0, 0, 0, 0};
};
namespace {
// grows a model buffer to hold `rows` triplets, keeping the first `keep` rows
int model_reserve(r3d_model *m, r3d_buf &b, int64_t rows, int64_t keep) {
    r3d_ctx *ctx = m->ctx;
    const size_t bytes = (size_t)rows * 24;
    if (bytes <= b.cap) return R3D_OK;
    const size_t want = bytes + bytes / 2 + 4096;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return r3d_fail(ctx, R3D_E_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    if (b.p && keep > 0) {
        e = hipMemcpyAsync(np, b.p, (size_t)keep * 24, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(np); return r3d_fail(ctx, R3D_E_HIP, "model grow copy failed: %s", hipGetErrorString(e)); }
    }
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
    }
    b.p = np;
    b.cap = want;
    return R3D_OK;
}
// appends n rows (device or host source) with the attribute rule of the legacy operator+= (an attribute survives only when
// the model is empty or already carries it AND the appended cloud carries it)
int model_append(r3d_model *m, const double *xyz, const double *colors, const double *normals, int64_t n, hipMemcpyKind kind) {
    r3d_ctx *ctx = m->ctx;
    const bool keep_c = (m->n == 0 || m->has_colors) && colors, keep_n = (m->n == 0 || m->has_normals) && normals;
    int rc;
    if ((rc = model_reserve(m, m->pts, m->n + n, m->n))) return rc;
    if (keep_c && (rc = model_reserve(m, m->cols, m->n + n, m->n))) return rc;
    if (keep_n && (rc = model_reserve(m, m->nrms, m->n + n, m->n))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync((double *)m->pts.p + m->n * 3, xyz, (size_t)n * 24, kind, ctx->stream));
    if (keep_c) R3D_HIP(ctx, hipMemcpyAsync((double *)m->cols.p + m->n * 3, colors, (size_t)n * 24, kind, ctx->stream));
    if (keep_n) R3D_HIP(ctx, hipMemcpyAsync((double *)m->nrms.p + m->n * 3, normals, (size_t)n * 24, kind, ctx->stream));
    if (kind == hipMemcpyHostToDevice) R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the caller may reuse its arrays
    m->has_colors = keep_c;
    m->has_normals = keep_n;
    m->n += n;
    return R3D_OK;
}
}  // namespace
extern "C" {

int r3d_model_create(r3d_ctx *ctx, r3d_model **out) {
    if (!ctx || !out) return R3D_E_BADARG;
    r3d_model *m = new (std::nothrow) r3d_model;
    if (!m) return r3d_fail(ctx, R3D_E_OOM, "model_create: out of host memory");
    m->ctx = ctx;
    *out = m;
    return R3D_OK;
}

void r3d_model_destroy(r3d_model *m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (r3d_buf *b : {&m->pts, &m->cols, &m->nrms, &m->vt_keys[0], &m->vt_keys[1], &m->vt_sums[0], &m->vt_sums[1], &m->vt_cnt[0], &m->vt_cnt[1]})
        if (b->p) (void)hipFree(b->p);
    if (m->d_pend) (void)hipFree(m->d_pend);
    delete m;
}

int r3d_model_clear(r3d_model *m) {
    if (!m) return R3D_E_BADARG;
    m->n = 0;
    m->has_colors = m->has_normals = false;
    m->vt_valid = false;
    m->vt_n = m->vt_pts = 0;
    m->pend_lo = m->pend_hi = 0;
    m->pend_host_ok = false;
    return R3D_OK;
}

int r3d_model_size(r3d_model *m, int64_t *n, int32_t *has_colors, int32_t *has_normals) {
    if (!m) return R3D_E_BADARG;
    if (n) *n = m->n;
    if (has_colors) *has_colors = m->has_colors;
    if (has_normals) *has_normals = m->has_normals;
    return R3D_OK;
}

int r3d_model_voxel_table_stats(r3d_model *m, int64_t *voxels, int32_t *rebuilds, int32_t *updates) {
    if (!m) return R3D_E_BADARG;
    if (voxels) *voxels = m->vt_valid ? m->vt_n : 0;
    if (rebuilds) *rebuilds = m->vt_rebuilds;
    if (updates) *updates = m->vt_updates;
    return R3D_OK;
}

int r3d_model_append(r3d_model *m, const double *xyz, const double *colors, const double *normals, int64_t n) {
    R3D_ROCTX_RANGE("r3d_model_append");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (!xyz || n < 0) return r3d_fail(ctx, R3D_E_BADARG, "model_append: bad argument");
    if (n == 0) return R3D_OK;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    return model_append(m, xyz, colors, normals, n, hipMemcpyHostToDevice);
}

}  // extern "C"
namespace {
// grow-only table buffer for `rows` voxels
int vt_reserve(r3d_model *m, r3d_buf &b, int64_t rows, size_t elem) {
    const size_t bytes = (size_t)rows * elem;
    if (bytes <= b.cap) return R3D_OK;
    const size_t want = bytes + bytes / 2 + 4096;
    if (b.p) { (void)hipStreamSynchronize(m->ctx->stream); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return r3d_fail(m->ctx, R3D_E_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    b.cap = want;
    return R3D_OK;
}
// pointcloud_alignment.py:23 for the resident model: the legacy voxel grid of ALL model points, as means in lexicographic voxel
// order in a fresh arena buffer.  Default: the incremental table above (full rebuild whenever the model's minimum corner moved, the
// voxel size changed or the table does not exist); R3D_MODEL_IMPL=rebuild: the full sort of every model point, every frame.
int model_target_voxels(r3d_model *m, DevArena &ar, double voxel, double **d_tv_out, int64_t *mt_out, double *box_out /* [6]: the model's box */) {
    r3d_ctx *ctx = m->ctx;
    int rc;
    double *d_t = (double *)m->pts.p;
    static const bool always_rebuild = [] { const char *e = getenv("R3D_MODEL_IMPL"); return e && !strcmp(e, "rebuild"); }();
    if (always_rebuild) {
        VoxelSegs Vt;
        if ((rc = voxel_segments(ctx, ar, d_t, m->n, voxel, Vt, false, nullptr, nullptr, box_out))) return rc;
        double *d_tv = (double *)ar.get((size_t)Vt.nseg * 24);
        if (ar.rc) return ar.rc;
        if ((rc = launch_voxel_mean(ctx, ar, false, d_t, Vt.idx, Vt.starts, Vt.nseg, m->n, d_tv, Vt.sorted))) return rc;
        *d_tv_out = d_tv;
        *mt_out = Vt.nseg;
        return R3D_OK;
    }
    bool rebuild = !m->vt_valid || m->vt_voxel != voxel || m->vt_pts > m->n;
    const int64_t ns = m->n - m->vt_pts;
    double smn[3], smx[3];
    if (!rebuild && ns > 0) {
        if (m->pend_host_ok && m->pend_lo == m->vt_pts && m->pend_hi == m->n) {   // the slice is the block the last alignment appended
            for (int a = 0; a < 3; a++) { smn[a] = m->pend_box[a]; smx[a] = m->pend_box[3 + a]; }
            if ((rc = bbox_check(ctx, smn, smx))) return rc;
        } else if ((rc = cloud_bbox(ctx, ar, d_t + m->vt_pts * 3, ns, smn, smx))) return rc;
        for (int a = 0; a < 3; a++) if (smn[a] < m->vt_min[a]) rebuild = true;   // the grid origin moves: every key changes
    }
    auto key_space_ok = [&](const double *mn, const double *mx) {
        for (int a = 0; a < 3; a++) if (!((mx[a] - (mn[a] - 0.5 * voxel)) / voxel < 2097000.0)) return false;
        return true;
    };
    const unsigned long long *tkeys = nullptr;
    if (rebuild) {
        VoxelSegs Vt;
        double bounds[6];
        if ((rc = voxel_segments(ctx, ar, d_t, m->n, voxel, Vt, false, nullptr, nullptr, bounds))) return rc;
        for (int a = 0; a < 6; a++) box_out[a] = bounds[a];
        if (!key_space_ok(bounds, bounds + 3)) {          // more than 2^21 voxels along an axis: no table, the plain full pass
            m->vt_valid = false;
            double *d_tv = (double *)ar.get((size_t)Vt.nseg * 24);
            if (ar.rc) return ar.rc;
            if ((rc = launch_voxel_mean(ctx, ar, false, d_t, Vt.idx, Vt.starts, Vt.nseg, m->n, d_tv, Vt.sorted))) return rc;
            *d_tv_out = d_tv;
            *mt_out = Vt.nseg;
            return R3D_OK;
        }
        const int c = m->vt_cur;
        if ((rc = vt_reserve(m, m->vt_keys[c], Vt.nseg, 8)) || (rc = vt_reserve(m, m->vt_sums[c], Vt.nseg, 24)) || (rc = vt_reserve(m, m->vt_cnt[c], Vt.nseg, 4))) return rc;
        k_vt_build<8><<<(unsigned)((Vt.nseg + 255) / 256), 256, 0, ctx->stream>>>(Vt.sorted, Vt.starts, Vt.nseg, m->n, bounds[0] - 0.5 * voxel,
                                                                               bounds[1] - 0.5 * voxel, bounds[2] - 0.5 * voxel, voxel,
                                                                               (unsigned long long *)m->vt_keys[c].p, (double *)m->vt_sums[c].p, (int *)m->vt_cnt[c].p);
        R3D_HIP(ctx, hipGetLastError());
        for (int a = 0; a < 3; a++) { m->vt_min[a] = bounds[a]; m->vt_max[a] = bounds[3 + a]; }
        m->vt_n = Vt.nseg;
        m->vt_voxel = voxel;
        m->vt_valid = true;
        m->vt_rebuilds++;
    } else if (ns > 0) {
        double nmx[3], org[3];
        for (int a = 0; a < 3; a++) { nmx[a] = std::max(m->vt_max[a], smx[a]); org[a] = m->vt_min[a] - 0.5 * voxel; }
        if (!key_space_ok(m->vt_min, nmx)) { m->vt_valid = false; return model_target_voxels(m, ar, voxel, d_tv_out, mt_out, box_out); }
        VoxelSegs V;
        if ((rc = voxel_segments(ctx, ar, d_t + m->vt_pts * 3, ns, voxel, V, false, org, nmx))) return rc;
        const int c = m->vt_cur, o = c ^ 1;
        const int64_t cap = m->vt_n + V.nseg;
        if ((rc = vt_reserve(m, m->vt_keys[o], cap, 8)) || (rc = vt_reserve(m, m->vt_sums[o], cap, 24)) || (rc = vt_reserve(m, m->vt_cnt[o], cap, 4))) return rc;
        unsigned long long *segkey = (unsigned long long *)ar.get((size_t)V.nseg * 8);
        int64_t *ins = (int64_t *)ar.get((size_t)V.nseg * 8);
        double *nsum = (double *)ar.get((size_t)V.nseg * 24);
        int *is_new = (int *)ar.get((size_t)V.nseg * 4), *rank = (int *)ar.get((size_t)V.nseg * 4), *ncnt = (int *)ar.get((size_t)V.nseg * 4);
        if (ar.rc) return ar.rc;
        const unsigned nbs = (unsigned)((V.nseg + 255) / 256);
        k_vt_update<<<nbs, 256, 0, ctx->stream>>>(V.sorted, V.starts, V.nseg, ns, org[0], org[1], org[2], voxel, (const unsigned long long *)m->vt_keys[c].p,
                                                  m->vt_n, (double *)m->vt_sums[c].p, (int *)m->vt_cnt[c].p, segkey, is_new, ins, nsum, ncnt);
        if (int src = dev_exclusive_scan<int>(ctx, ar, is_new, rank, V.nseg)) return src;
        int last_rank = 0, last_flag = 0;
        {
            PinRead rd(ctx);
            int prc;
            if ((prc = rd.add(&last_rank, rank + (V.nseg - 1), 4)) || (prc = rd.add(&last_flag, is_new + (V.nseg - 1), 4)) || (prc = rd.wait())) return prc;
        }
        const int total_new = last_rank + last_flag;
        if (total_new > 0) {
            k_vt_merge<<<(unsigned)((m->vt_n + V.nseg + 255) / 256), 256, 0, ctx->stream>>>(
                (const unsigned long long *)m->vt_keys[c].p, (const double *)m->vt_sums[c].p, (const int *)m->vt_cnt[c].p, m->vt_n, segkey, is_new, rank, ins, nsum,
                ncnt, V.nseg, total_new, (unsigned long long *)m->vt_keys[o].p, (double *)m->vt_sums[o].p, (int *)m->vt_cnt[o].p);
            m->vt_cur = o;
            m->vt_n += total_new;
        }
        R3D_HIP(ctx, hipGetLastError());
        for (int a = 0; a < 3; a++) m->vt_max[a] = nmx[a];
        m->vt_updates++;
    }
    m->vt_pts = m->n;
    tkeys = (const unsigned long long *)m->vt_keys[m->vt_cur].p;
    (void)tkeys;
    double *d_tv = (double *)ar.get((size_t)m->vt_n * 24);
    if (ar.rc) return ar.rc;
    k_vt_means<<<(unsigned)((m->vt_n + 255) / 256), 256, 0, ctx->stream>>>((const double *)m->vt_sums[m->vt_cur].p, (const int *)m->vt_cnt[m->vt_cur].p, m->vt_n, d_tv);
    R3D_HIP(ctx, hipGetLastError());
    for (int a = 0; a < 3; a++) { box_out[a] = m->vt_min[a]; box_out[3 + a] = m->vt_max[a]; }
    *d_tv_out = d_tv;
    *mt_out = m->vt_n;
    return R3D_OK;
}
int model_align_checks(r3d_model *m, const r3d_align_params *p, const char *who) {
    r3d_ctx *ctx = m->ctx;
    if (m->n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "%s: the model is empty (append the first frame)", who);
    const r3d_icp_params *ip = &p->icp;
    if (ip->mode < 0 || ip->mode > 2) return r3d_fail(ctx, R3D_E_BADARG, "%s: mode must be 0 (P2P), 1 (P2PLANE) or 2 (GICP)", who);
    if (!(ip->max_correspondence_distance > 0)) return r3d_fail(ctx, R3D_E_BADARG, "%s: max_correspondence_distance must be > 0", who);
    if (p->normal_max_nn > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "%s: normal_max_nn > 128 not supported", who);
    if (ip->mode != MODE_P2P && p->normal_max_nn <= 0) return r3d_fail(ctx, R3D_E_BADARG, "%s: this mode needs normals (normal_max_nn > 0)", who);
    return R3D_OK;
}
// the frame (d_s, optional colours d_c, ns points) is already on the device in arena `ar`
int model_align_core(r3d_model *m, DevArena &ar, const r3d_align_params *p, double *d_s, double *d_c, int64_t ns, double *T4x4,
                     r3d_icp_stats *stats, int64_t *appended, std::chrono::steady_clock::time_point t_begin,
                     const double *frame_box = nullptr /* bounding box of the frame's points when the caller already has it */) {
    r3d_ctx *ctx = m->ctx;
    const r3d_icp_params *ip = &p->icp;
    int rc;
    double *d_t = (double *)m->pts.p;
    int64_t ms = ns, mt = m->n;
    const double *tgt_box = nullptr;
    double tbox[6];
    if (p->voxel_size > 0) {  // pointcloud_alignment.py:22-23 on the frame and on the WHOLE resident model
        VoxelSegs V;
        if ((rc = voxel_segments(ctx, ar, d_s, ns, p->voxel_size, V, false, nullptr, nullptr, nullptr, frame_box))) return rc;
        double *d_sv = (double *)ar.get((size_t)V.nseg * 24), *d_cv = d_c ? (double *)ar.get((size_t)V.nseg * 24) : nullptr;
        if (ar.rc) return ar.rc;
        if ((rc = launch_voxel_mean(ctx, ar, false, d_s, V.idx, V.starts, V.nseg, ns, d_sv, V.sorted))) return rc;
        if (d_c) if ((rc = launch_voxel_mean(ctx, ar, false, d_c, V.idx, V.starts, V.nseg, ns, d_cv))) return rc;
        R3D_HIP(ctx, hipGetLastError());
        d_s = d_sv;
        d_c = d_cv;
        ms = V.nseg;
        double *d_tv = nullptr;
        if ((rc = model_target_voxels(m, ar, p->voxel_size, &d_tv, &mt, tbox))) return rc;
        d_t = d_tv;
        tgt_box = tbox;   // the model's own box (the table keeps it): the voxel means lie inside it, no bounding-box pass for the search grid
    }
    // pointcloud_alignment.py:27-28 estimates normals on both clouds.  The point-to-point estimator never reads them and the
    // model's operator+= drops the frame's normals unless the model carries normals itself, so they are computed only where
    // they can reach a result: target normals for the plane / GICP estimators, source normals for GICP or a model with normals.
    double *d_sn = nullptr, *d_tn = nullptr;
    const bool want_sn = p->normal_max_nn > 0 && (ip->mode == MODE_GICP || m->has_normals);
    if (want_sn && (rc = normals_core(ctx, ar, d_s, ms, p->normal_radius, p->normal_max_nn, nullptr, &d_sn))) return rc;
    if (ip->mode != MODE_P2P && (rc = normals_core(ctx, ar, d_t, mt, p->normal_radius, p->normal_max_nn, nullptr, &d_tn))) return rc;
    double T[16];
    if ((rc = icp_core(ctx, ar, ip, d_s, ms, d_sn, d_t, mt, d_tn, nullptr, T, stats, t_begin, tgt_box, p->voxel_size > 0 ? p->voxel_size : 0))) return rc;  // :35-39, from identity
    memcpy(T4x4, T, sizeof T);
    double *d_o = (double *)ar.get((size_t)ms * 24), *d_on = d_sn ? (double *)ar.get((size_t)ms * 24) : nullptr;
    if (ar.rc) return ar.rc;
    const unsigned nb = (unsigned)((ms + 255) / 256);
    k_transform<<<nb, 256, 0, ctx->stream>>>(d_s, ms, to_rigid(T), 0, d_o);  // :42 source.transform
    if (d_sn) k_transform<<<nb, 256, 0, ctx->stream>>>(d_sn, ms, to_rigid(T), 1, d_on);
    R3D_HIP(ctx, hipGetLastError());
    const int64_t n_before = m->n;
    if ((rc = model_append(m, d_o, d_c, d_on, ms, hipMemcpyDeviceToDevice))) return rc;   // main.py:49 combined += aligned
    if (appended) *appended = ms;
    // the appended block's bounding box, for the next frame's table update: enqueued now, fetched with that call's first read-back
    m->pend_host_ok = false;
    m->pend_lo = m->pend_hi = 0;
    if (p->voxel_size > 0 && ms > 0) {
        if (!m->d_pend) {
            hipError_t e = hipMalloc((void **)&m->d_pend, 64);
            if (e != hipSuccess) { m->d_pend = nullptr; return R3D_OK; }       // no pending box: the next call computes it itself
        }
        if (cloud_bbox_enqueue_dn(ctx, ar, d_o, ms, nullptr, nullptr, nullptr, m->d_pend) == R3D_OK) { m->pend_lo = n_before; m->pend_hi = m->n; }
    }
    return R3D_OK;
}
}  // namespace
extern "C" {

int r3d_model_align_append(r3d_model *m, const r3d_align_params *p, const double *src, const double *src_colors, int64_t ns,
                           double *T4x4, r3d_icp_stats *stats, int64_t *appended) {
    R3D_ROCTX_RANGE("r3d_model_align_append");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (!p || !src || !T4x4 || ns <= 0) return r3d_fail(ctx, R3D_E_BADARG, "model_align_append: bad argument");
    int rc;
    if ((rc = model_align_checks(m, p, "model_align_append"))) return rc;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    double *d_s, *d_c = nullptr;
    if ((rc = upload(ctx, ar, src, ns * 3, &d_s))) return rc;
    if (src_colors && (rc = upload(ctx, ar, src_colors, ns * 3, &d_c))) return rc;
    return model_align_core(m, ar, p, d_s, d_c, ns, T4x4, stats, appended, t_begin);
}

// The same two steps of main.py:34-49 for a frame that is still a DEPTH IMAGE (the recorded frames of test/output84, a RealSense
// z16 frame): create_from_rgbd_image + flip (test/check84.py:155-159,172-178) runs on the device, so a 0.6 MB image goes up
// instead of 6.8 MB of float64 points.  color (may be NULL): uint8 [h][w][3], channels taken in the order given.
int r3d_model_append_depth(r3d_model *m, const uint16_t *depth, int32_t w, int32_t h, int32_t stride, const r3d_depth_camera *cam,
                           const uint8_t *color, int32_t color_stride, int64_t *appended) {
    R3D_ROCTX_RANGE("r3d_model_append_depth");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    int rc;
    if ((rc = depth_args_ok(ctx, depth, w, h, stride, cam, color, color_stride, "model_append_depth"))) return rc;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p, *d_c;
    int64_t n;
    if ((rc = backproject_core(ctx, ar, depth, w, h, stride, cam, color, color_stride, false, &d_p, &d_c, nullptr, &n))) return rc;
    if (appended) *appended = n;
    if (n == 0) return R3D_OK;
    return model_append(m, d_p, d_c, nullptr, n, hipMemcpyDeviceToDevice);
}

int r3d_model_align_append_depth(r3d_model *m, const r3d_align_params *p, const uint16_t *depth, int32_t w, int32_t h, int32_t stride,
                                 const r3d_depth_camera *cam, const uint8_t *color, int32_t color_stride, double *T4x4, r3d_icp_stats *stats,
                                 int64_t *frame_points, int64_t *appended) {
    R3D_ROCTX_RANGE("r3d_model_align_append_depth");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (!p || !T4x4) return r3d_fail(ctx, R3D_E_BADARG, "model_align_append_depth: bad argument");
    int rc;
    if ((rc = depth_args_ok(ctx, depth, w, h, stride, cam, color, color_stride, "model_align_append_depth"))) return rc;
    if ((rc = model_align_checks(m, p, "model_align_append_depth"))) return rc;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    double *d_p, *d_c;
    int64_t n;
    double fbox[6];
    PinExtra pend;
    if (m->d_pend && m->pend_hi > m->pend_lo && !m->pend_host_ok) { pend.host = m->pend_box; pend.dev = m->d_pend; pend.bytes = 48; }
    if ((rc = backproject_core(ctx, ar, depth, w, h, stride, cam, color, color_stride, false, &d_p, &d_c, nullptr, &n, fbox, &pend))) return rc;
    if (pend.host) m->pend_host_ok = true;
    if (frame_points) *frame_points = n;
    if (appended) *appended = 0;
    if (n == 0) {   // no valid pixel: a failed capture, skipped like main.py:53-54 (identity, empty statistics, nothing appended)
        for (int i = 0; i < 16; i++) T4x4[i] = (i % 5 == 0);
        if (stats) memset(stats, 0, sizeof *stats);
        return R3D_OK;
    }
    return model_align_core(m, ar, p, d_p, d_c, n, T4x4, stats, appended, t_begin, fbox);
}

int r3d_backproject_depth(r3d_ctx *ctx, const uint16_t *depth, int32_t w, int32_t h, int32_t stride, const r3d_depth_camera *cam,
                          const uint8_t *color, int32_t color_stride, double *out_xyz, double *out_colors, int32_t *out_pixel, int64_t *out_n) {
    R3D_ROCTX_RANGE("r3d_backproject_depth");
    if (!ctx) return R3D_E_BADARG;
    if (!out_xyz || !out_n || (color && !out_colors)) return r3d_fail(ctx, R3D_E_BADARG, "backproject_depth: bad argument");
    int rc;
    if ((rc = depth_args_ok(ctx, depth, w, h, stride, cam, color, color_stride, "backproject_depth"))) return rc;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p, *d_c;
    int *d_pix;
    int64_t n;
    if ((rc = backproject_core(ctx, ar, depth, w, h, stride, cam, color, color_stride, out_pixel != nullptr, &d_p, &d_c, &d_pix, &n))) return rc;
    *out_n = n;
    if (n == 0) return R3D_OK;
    R3D_HIP(ctx, hipMemcpyAsync(out_xyz, d_p, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (d_c) R3D_HIP(ctx, hipMemcpyAsync(out_colors, d_c, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (out_pixel) R3D_HIP(ctx, hipMemcpyAsync(out_pixel, d_pix, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_model_register_append(r3d_model *m, const r3d_icp_params *p, const double *src, const double *src_colors, const double *src_normals,
                              int64_t ns, double *T4x4, r3d_icp_stats *stats) {
    R3D_ROCTX_RANGE("r3d_model_register_append");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (!p || !src || !T4x4 || ns <= 0) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: bad argument");
    if (m->n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: the model is empty (append the first frame)");
    if (p->mode < 0 || p->mode > 2) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: mode must be 0 (P2P), 1 (P2PLANE) or 2 (GICP)");
    if (!(p->max_correspondence_distance > 0)) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: max_correspondence_distance must be > 0");
    if (p->mode != MODE_P2P && !m->has_normals) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: the model has no normals (r3d_model_estimate_normals)");
    if (p->mode == MODE_GICP && !src_normals) return r3d_fail(ctx, R3D_E_BADARG, "model_register_append: source normals required for GICP");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    int rc;
    double *d_s, *d_c = nullptr, *d_sn = nullptr;
    if ((rc = upload(ctx, ar, src, ns * 3, &d_s))) return rc;
    if (src_colors && (rc = upload(ctx, ar, src_colors, ns * 3, &d_c))) return rc;
    if (src_normals && (rc = upload(ctx, ar, src_normals, ns * 3, &d_sn))) return rc;
    double T[16];
    if ((rc = icp_core(ctx, ar, p, d_s, ns, d_sn, (double *)m->pts.p, m->n, m->has_normals ? (double *)m->nrms.p : nullptr, nullptr, T, stats, t_begin)))
        return rc;                                                                     // test/GICP1.py:99-102, from identity
    memcpy(T4x4, T, sizeof T);
    double *d_o = (double *)ar.get((size_t)ns * 24), *d_on = d_sn ? (double *)ar.get((size_t)ns * 24) : nullptr;
    if (ar.rc) return ar.rc;
    const unsigned nb = (unsigned)((ns + 255) / 256);
    k_transform<<<nb, 256, 0, ctx->stream>>>(d_s, ns, to_rigid(T), 0, d_o);            // :103 source.transform
    if (d_sn) k_transform<<<nb, 256, 0, ctx->stream>>>(d_sn, ns, to_rigid(T), 1, d_on);
    R3D_HIP(ctx, hipGetLastError());
    return model_append(m, d_o, d_c, d_on, ns, hipMemcpyDeviceToDevice);               // :146 combined += aligned
}

int r3d_model_estimate_normals(r3d_model *m, double radius, int32_t max_nn) {
    R3D_ROCTX_RANGE("r3d_model_estimate_normals");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (max_nn < 1) return r3d_fail(ctx, R3D_E_BADARG, "model_estimate_normals: bad argument");
    if (max_nn > 128) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "model_estimate_normals: max_nn > 128 not supported");
    if (m->n == 0) return R3D_OK;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    int rc;
    double *d_n;
    // legacy EstimateNormals on a cloud that already has normals keeps their orientation (test/GICP1.py:148)
    if ((rc = normals_core(ctx, ar, (double *)m->pts.p, m->n, radius, max_nn, m->has_normals ? (double *)m->nrms.p : nullptr, &d_n))) return rc;
    if ((rc = model_reserve(m, m->nrms, m->n, m->has_normals ? m->n : 0))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(m->nrms.p, d_n, (size_t)m->n * 24, hipMemcpyDeviceToDevice, ctx->stream));
    m->has_normals = true;
    return R3D_OK;
}

int r3d_model_download(r3d_model *m, double *xyz, double *colors, double *normals) {
    R3D_ROCTX_RANGE("r3d_model_download");
    if (!m) return R3D_E_BADARG;
    r3d_ctx *ctx = m->ctx;
    if (!xyz) return r3d_fail(ctx, R3D_E_BADARG, "model_download: bad argument");
    if (m->n == 0) return R3D_OK;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    R3D_HIP(ctx, hipMemcpyAsync(xyz, m->pts.p, (size_t)m->n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (colors && m->has_colors) R3D_HIP(ctx, hipMemcpyAsync(colors, m->cols.p, (size_t)m->n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (normals && m->has_normals) R3D_HIP(ctx, hipMemcpyAsync(normals, m->nrms.p, (size_t)m->n * 24, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

// Device-pointer forms for callers that keep their clouds in HBM (the multi-view exchange of config C5: clouds produced by
// r3d_disparity_to_cloud_resident, gathered by RCCL, registered and transformed without touching the host).  Work is enqueued on
// the context stream; r3d_icp_dev returns when the loop has finished (it reads the state back), r3d_transform_points_dev is
// asynchronous.
int r3d_icp_dev(r3d_ctx *ctx, const r3d_icp_params *p, const double *d_src, int64_t ns, const double *d_src_normals, const double *d_tgt,
                int64_t nt, const double *d_tgt_normals, const double *init4x4, double *T4x4, r3d_icp_stats *stats) {
    R3D_ROCTX_RANGE("r3d_icp_dev");
    if (!ctx) return R3D_E_BADARG;
    if (!p || !d_src || !d_tgt || !T4x4 || ns <= 0 || nt <= 0) return r3d_fail(ctx, R3D_E_BADARG, "icp_dev: bad argument");
    if (p->mode < 0 || p->mode > 2) return r3d_fail(ctx, R3D_E_BADARG, "icp_dev: mode must be 0 (P2P), 1 (P2PLANE) or 2 (GICP)");
    if (!(p->max_correspondence_distance > 0)) return r3d_fail(ctx, R3D_E_BADARG, "icp_dev: max_correspondence_distance must be > 0");
    if (p->mode != MODE_P2P && !d_tgt_normals) return r3d_fail(ctx, R3D_E_BADARG, "icp_dev: target normals required for this mode");
    if (p->mode == MODE_GICP && !d_src_normals) return r3d_fail(ctx, R3D_E_BADARG, "icp_dev: source normals required for GICP");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    DevArena ar(ctx);
    return icp_core(ctx, ar, p, const_cast<double *>(d_src), ns, const_cast<double *>(d_src_normals), const_cast<double *>(d_tgt), nt,
                    const_cast<double *>(d_tgt_normals), init4x4, T4x4, stats, t_begin);
}

int r3d_transform_points_dev(r3d_ctx *ctx, const double *d_xyz, int64_t n, const double *T4x4, int32_t rotate_only, double *d_out) {
    R3D_ROCTX_RANGE("r3d_transform_points_dev");
    if (!ctx) return R3D_E_BADARG;
    if (!d_xyz || !d_out || !T4x4 || n < 0) return r3d_fail(ctx, R3D_E_BADARG, "transform_points_dev: bad argument");
    if (n == 0) return R3D_OK;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    k_transform<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_xyz, n, to_rigid(T4x4), rotate_only, d_out);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

int r3d_transform_blocks_dev(r3d_ctx *ctx, int32_t n_blocks, const double *const *d_xyz, const int64_t *counts, const double *T4x4s,
                             const int32_t *rotate_only, double *const *d_out) {
    R3D_ROCTX_RANGE("r3d_transform_blocks_dev");
    if (!ctx) return R3D_E_BADARG;
    if (n_blocks < 0 || (n_blocks > 0 && (!d_xyz || !counts || !T4x4s || !d_out))) return r3d_fail(ctx, R3D_E_BADARG, "transform_blocks_dev: bad argument");
    for (int b = 0; b < n_blocks; b++)
        if (counts[b] < 0 || (counts[b] > 0 && (!d_xyz[b] || !d_out[b]))) return r3d_fail(ctx, R3D_E_BADARG, "transform_blocks_dev: block %d: bad argument", b);
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    for (int b = 0; b < n_blocks;) {
        TransformBlocks B;
        B.n = 0;
        long long end = 0;
        for (; b < n_blocks && B.n < TB_MAX; b++) {   // the next TB_MAX non-empty blocks
            if (counts[b] == 0) continue;
            const double *T = T4x4s + (size_t)b * 16;
            const bool ro = rotate_only && rotate_only[b];
            const int q = B.n++;
            B.in[q] = d_xyz[b];
            B.out[q] = d_out[b];
            end += counts[b];
            B.end[q] = end;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) B.r[q][i * 3 + j] = T[i * 4 + j];
                B.r[q][9 + i] = ro ? 0.0 : T[i * 4 + 3];
            }
        }
        if (B.n == 0) continue;
        for (int q = B.n; q < TB_MAX; q++) { B.in[q] = nullptr; B.out[q] = nullptr; B.end[q] = end; }
        k_transform_blocks<<<(unsigned)((end + 255) / 256), 256, 0, ctx->stream>>>(B);
        R3D_HIP(ctx, hipGetLastError());
    }
    return R3D_OK;
}

}  // extern "C"

// diagnostic: the cell sort on its own (tests compare both implementations with a host lexsort)
#ifdef R3D_ICP_STATS
extern "C" int r3d_debug_icp_stats(uint64_t *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_icp_stats), 16 * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_icp_stats), z, 16 * 8) != hipSuccess) return -1; }
    return 0;
}
#endif
extern "C" int r3d_debug_sort_by_cell(r3d_ctx *ctx, const double *xyz, int64_t n, const double *org3, double cell, const int32_t *dims3, int32_t key_order,
                                      int32_t impl, int32_t *idx_out, uint64_t *keys_out) {
    R3D_ROCTX_RANGE("r3d_debug_sort_by_cell");
    if (!ctx) return R3D_E_BADARG;
    if (!xyz || !org3 || !dims3 || !idx_out || n <= 0 || !(cell > 0) || key_order < 0 || key_order > 3 || impl < 0 || impl > 1)
        return r3d_fail(ctx, R3D_E_BADARG, "debug_sort_by_cell: bad argument");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    double *d_p;
    int rc;
    if ((rc = upload(ctx, ar, xyz, n * 3, &d_p))) return rc;
    const int dims[3] = {dims3[0], dims3[1], dims3[2]};
    void *keys;
    bool k32;
    int *idx;
    if ((rc = sort_by_cell(ctx, ar, d_p, n, org3, cell, dims, key_order, &keys, &k32, &idx, nullptr, impl))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(idx_out, idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<unsigned> k4;
    if (keys_out) {
        if (k32) {
            k4.resize((size_t)n);
            R3D_HIP(ctx, hipMemcpyAsync(k4.data(), keys, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else R3D_HIP(ctx, hipMemcpyAsync(keys_out, keys, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keys_out && k32)
        for (int64_t i = 0; i < n; i++) keys_out[i] = k4[(size_t)i];
    return R3D_OK;
}

extern "C" int r3d_debug_exclusive_scan(r3d_ctx *ctx, const int32_t *in, int64_t n, int32_t op, int32_t *out) {
    R3D_ROCTX_RANGE("r3d_debug_exclusive_scan");
    if (!ctx) return R3D_E_BADARG;
    if (!in || !out || n <= 0 || op < 0 || op > 1) return r3d_fail(ctx, R3D_E_BADARG, "debug_exclusive_scan: bad argument");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    DevArena ar(ctx);
    int *d_in = (int *)ar.get((size_t)n * 4), *d_out = (int *)ar.get((size_t)n * 4);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = op == 1 ? dev_exclusive_scan<int, true>(ctx, ar, d_in, d_out, n) : dev_exclusive_scan<int>(ctx, ar, d_in, d_out, n);
    if (rc) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}
