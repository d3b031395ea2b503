// Per-frame stages either side of the SGM matcher in Calib_depth/depth*.py, MI355X (gfx950):
//   rectification maps (initUndistortRectifyMap, CV_16SC2)          depth2.py:125-128
//   remap INTER_LINEAR with fixed-point maps (+ fused grey output)   depth2.py:243-244
//   cvtColor BGR2GRAY                                                 depth2.py:247-248
//   disparity WLS filter (confidence + fast global smoother)          depth2.py:164-166,255
//   normalize NORM_MINMAX                                             depth2.py:256
// All HBM-bound byte / float work except the smoother, whose tridiagonal solves are latency chains: one lane per image
// row (tiles transposed through LDS so that global accesses stay coalesced) or per image column.
#include <math.h>

#include <algorithm>
#include <cmath>

#include "r3d_internal.h"

namespace {

struct PPArena {  // bump allocator over ctx->pp_bufs (grow-only, reused across calls)
    r3d_ctx *ctx;
    size_t next = 0;
    int rc = R3D_OK;
    explicit PPArena(r3d_ctx *c) : ctx(c) {}
    void *get(size_t bytes) {
        if (rc) return nullptr;
        if (next >= ctx->pp_bufs.size()) ctx->pp_bufs.emplace_back();
        r3d_buf &b = ctx->pp_bufs[next++];
        rc = r3d_reserve(ctx, b, bytes ? bytes : 16);
        return rc ? nullptr : b.p;
    }
};

// ------------------------------------------------------------------------------------------ rectification maps

struct RectParams {
    double ir[9];
    double k[12];  // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    double u0, v0, fx, fy;
};

// one thread per destination row: the reference accumulates _x/_y/_w along the row (+= ir[0], ir[3], ir[6]), which is a
// sequential fp64 chain; rows are independent
__global__ void k_rectify_map(RectParams P, int W, int H, int16_t *__restrict__ map1, uint16_t *__restrict__ map2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const double k1 = P.k[0], k2 = P.k[1], p1 = P.k[2], p2 = P.k[3], k3 = P.k[4], k4 = P.k[5], k5 = P.k[6], k6 = P.k[7];
    const double s1 = P.k[8], s2 = P.k[9], s3 = P.k[10], s4 = P.k[11];
    double _x = i * P.ir[1] + P.ir[2], _y = i * P.ir[4] + P.ir[5], _w = i * P.ir[7] + P.ir[8];
    int16_t *m1 = map1 + (size_t)i * W * 2;
    uint16_t *m2 = map2 + (size_t)i * W;
    for (int j = 0; j < W; j++, _x += P.ir[0], _y += P.ir[3], _w += P.ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
        const double u = P.fx * 1.0 * xd + P.u0;
        const double v = P.fy * 1.0 * yd + P.v0;
        const double lim = 2147483648.0;
        const double ru = fmin(fmax(rint(u * 32.0), -lim), lim - 1.0), rv = fmin(fmax(rint(v * 32.0), -lim), lim - 1.0);
        const int iu = (int)ru, iv = (int)rv;
        m1[j * 2] = (int16_t)(iu >> 5);
        m1[j * 2 + 1] = (int16_t)(iv >> 5);
        m2[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
    }
}

// ---------------------------------------------------------------------------------------------------- remap

// fixed-point bilinear weights of table entry (fy, fx): products of 5-bit fractions scaled to 2^15; the (0,0) entry is
// (32767, 0, 0, 1) because 32768 saturates in a short and the missing unit goes to the last tap
__device__ __forceinline__ void bilinear_w(int frac, int &w0, int &w1, int &w2, int &w3) {
    const int fx = frac & 31, fy = (frac >> 5) & 31;
    w0 = (32 - fy) * (32 - fx) * 32;
    w1 = (32 - fy) * fx * 32;
    w2 = fy * (32 - fx) * 32;
    w3 = fy * fx * 32;
    if (frac == 0) { w0 = 32767; w3 = 1; }
}

__device__ __forceinline__ int gray_of(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14; }

template <int CN>
__global__ void k_remap(const uint8_t *__restrict__ src, int sw, int sh, int sstride, const int16_t *__restrict__ map1,
                        const uint16_t *__restrict__ map2, int dw, int dh, int border, uint8_t *__restrict__ dst,
                        uint8_t *__restrict__ gray) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const size_t p = (size_t)y * dw + x;
    const int sx = map1[p * 2], sy = map1[p * 2 + 1];
    int w0, w1, w2, w3;
    bilinear_w(map2[p] & 1023, w0, w1, w2, w3);
    int out[CN];
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
#pragma unroll
        for (int c = 0; c < CN; c++) out[c] = border;
    } else {
        const bool x0 = sx >= 0, x1 = sx + 1 < sw, y0 = sy >= 0, y1 = sy + 1 < sh;
        const uint8_t *r0 = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * CN;
        const uint8_t *r1 = r0 + sstride;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const int t0 = (x0 && y0) ? r0[c] : border, t1 = (x1 && y0) ? r0[CN + c] : border;
            const int t2 = (x0 && y1) ? r1[c] : border, t3 = (x1 && y1) ? r1[CN + c] : border;
            out[c] = (t0 * w0 + t1 * w1 + t2 * w2 + t3 * w3 + (1 << 14)) >> 15;
        }
    }
#pragma unroll
    for (int c = 0; c < CN; c++) dst[p * CN + c] = (uint8_t)min(max(out[c], 0), 255);
    if (CN >= 3 && gray) gray[p] = (uint8_t)gray_of(min(max(out[0], 0), 255), min(max(out[1], 0), 255), min(max(out[2], 0), 255));
}

__global__ void k_bgr2gray(const uint8_t *__restrict__ bgr, int w, int h, int stride, int cn, uint8_t *__restrict__ gray) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *p = bgr + (size_t)y * stride + (size_t)x * cn;
    gray[(size_t)y * w + x] = (uint8_t)gray_of(p[0], p[1], p[2]);
}

// ------------------------------------------------------------------------------------------------ normalize

__global__ void k_minmax_init(int *mm) { mm[0] = 2147483647; mm[1] = -2147483647 - 1; }

__global__ void k_minmax_s16(const int16_t *__restrict__ a, int64_t n, int *mm) {
    int lo = 32767, hi = -32768;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t head = std::min<int64_t>(n, (int64_t)((16 - ((uintptr_t)a & 15)) & 15) / 2);  // elements before 16-B alignment
    const int64_t nv = (n - head) / 8;
    const int4 *v = (const int4 *)(a + head);
    for (int64_t i = tid; i < nv; i += nth) {
        const int4 q = v[i];
        const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e0 = (int)(int16_t)(w[j] & 0xffff), e1 = w[j] >> 16;
            lo = min(lo, min(e0, e1));
            hi = max(hi, max(e0, e1));
        }
    }
    for (int64_t i = tid; i < head; i += nth) { lo = min(lo, (int)a[i]); hi = max(hi, (int)a[i]); }
    for (int64_t i = head + nv * 8 + tid; i < n; i += nth) { lo = min(lo, (int)a[i]); hi = max(hi, (int)a[i]); }
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    __shared__ int slo[4], shi[4];  // one global atomic pair per workgroup: same-address atomics serialise at the memory side
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) { lo = min(lo, slo[w]); hi = max(hi, shi[w]); }
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

__global__ void k_normalize_s16(const int16_t *__restrict__ a, int64_t n, const int *mm, double lo, double hi,
                                int16_t *__restrict__ out) {
    const double smin = mm[0], smax = mm[1];
    const double scale = (hi - lo) * ((smax - smin) > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.);
    const double shift = lo - smin * scale;
    const float fa = (float)scale, fb = (float)shift;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = rintf((float)a[i] * fa + fb);
        out[i] = (int16_t)fminf(fmaxf(v, -32768.f), 32767.f);
    }
}

// ------------------------------------------------------------------------------------------------ WLS filter

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// depth-discontinuity confidence of one view inside its ROI [x0, x0+rw): 1 - roll_off * box-variance of the raw (x16)
// disparities, clipped at 0; box sums are integers (exact), mean = float(double(sum) * (1/k^2)) like a 32F boxFilter with
// fp64 accumulators; BORDER_REFLECT_101 at the ROI edge (the ROI is converted into its own matrix first)
#define DD_STRIP 32
// One wave per 64 columns x 32 rows; RT > 0: compile-time radius (tap loops unroll, loads batch), RT = 0: any radius.  Vertical running sums of horizontal (2r+1)-tap sums (integers: exact in any
// order); each row's horizontal sums are computed once and parked in an LDS ring of 2r+1 slots until they leave the window.
template <int RT>
__global__ __launch_bounds__(64) void k_wls_dd(const int16_t *__restrict__ disp, int W, int H, int x0, int rw, int radius_rt,
                                               float roll_off, float *__restrict__ dd) {
    extern __shared__ long long dd_ring[];  // [2r+1][64] sums of squares, then [2r+1][64] int sums
    const int radius = RT > 0 ? RT : radius_rt;
    const int k = 2 * radius + 1;
    long long *ring2 = dd_ring;
    int *ring1 = (int *)(dd_ring + (size_t)k * 64);
    const int lane = threadIdx.x, x = blockIdx.x * 64 + lane, ys = blockIdx.y * DD_STRIP;
    const bool live = x < rw;
    const int xc = min(x, rw - 1);
    const bool inner = xc - radius >= 0 && xc + radius < rw;
    auto hsum = [&](int yy, int &s, long long &s2) {
        const int16_t *row = disp + (size_t)reflect101(yy, H) * W + x0;
        int a = 0;
        long long a2 = 0;
        auto taps = [&](auto index) {
            if constexpr (RT > 0) {
#pragma unroll
                for (int dx = -RT; dx <= RT; dx++) { const int v = row[index(dx)]; a += v; a2 += (long long)(v * v); }
            } else {
                for (int dx = -radius; dx <= radius; dx++) { const int v = row[index(dx)]; a += v; a2 += (long long)(v * v); }
            }
        };
        if (inner) taps([&](int dx) { return xc + dx; });
        else taps([&](int dx) { return reflect101(xc + dx, rw); });
        s = a;
        s2 = a2;
    };
    long long S = 0, S2 = 0;
    for (int i = 0; i < k; i++) {  // rows ys-r .. ys+r; ring slot of row yy is (yy - (ys - r)) mod k
        int s;
        long long s2;
        hsum(ys - radius + i, s, s2);
        ring1[i * 64 + lane] = s;
        ring2[i * 64 + lane] = s2;
        S += s;
        S2 += s2;
    }
    const double sc = 1.0 / (double)(k * k);
    const int ye = min(ys + DD_STRIP, H);
    int slot = 0;  // slot of the oldest row (y - r)
    for (int y = ys; y < ye; y++) {
        const float mean = (float)((double)S * sc), sq = (float)((double)S2 * sc);
        const float var = sq - mean * mean;
        const float v = 1.0f - roll_off * var;
        if (live) dd[(size_t)y * W + x0 + x] = v > 0.f ? v : 0.f;
        if (y + 1 < ye) {
            int s;
            long long s2;
            hsum(y + radius + 1, s, s2);
            S += s - ring1[slot * 64 + lane];
            S2 += s2 - ring2[slot * 64 + lane];
            ring1[slot * 64 + lane] = s;
            ring2[slot * 64 + lane] = s2;
            slot = slot + 1 == k ? 0 : slot + 1;
        }
    }
}

// LR-consistency confidence over the left ROI, x255; writes the two planes the smoother runs on (ROI-sized, pitch lw):
// sig[0] = conf * disparity, sig[1] = conf; conf_full (image-sized, may be null) keeps the map for getConfidenceMap()
__global__ void k_wls_conf(const int16_t *__restrict__ dl, const int16_t *__restrict__ dr, const float *__restrict__ ddl,
                           const float *__restrict__ ddr, int W, int H, int lx, int lw, int rx, int rw, int lrc,
                           float *__restrict__ sig0, float *__restrict__ sig1, float *__restrict__ conf_full) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= lw || y >= H) return;
    const int j = lx + x;
    const size_t row = (size_t)y * W;
    const int d = dl[row + j];
    float c = ddl[row + j];
    const int ridx = j - (d >> 4);
    if (ridx >= rx && ridx < rx + rw) {
        const int s = d + (int)dr[row + ridx];
        c = (s < 0 ? -s : s) < lrc ? fminf(c, ddr[row + ridx]) : 0.f;
    }
    c = 255.0f * c;
    sig0[(size_t)y * lw + x] = c * (float)d;
    sig1[(size_t)y * lw + x] = c;
    if (conf_full) conf_full[row + j] = c;
}

// smoothness weights from the guide inside the ROI: ch[y][x] = LUT[|g(x)-g(x+1)|^2] (0 in the last column),
// cv[y][x] = LUT[|g(y)-g(y+1)|^2] (0 in the last row); LUT holds -exp(-sqrt(i)/sigma)
template <int CN>
__global__ void k_fgs_weights(const uint8_t *__restrict__ guide, int gstride, int lx, int lw, int H, const float *__restrict__ lut,
                              float *__restrict__ ch, float *__restrict__ cv) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= lw || y >= H) return;
    const uint8_t *p = guide + (size_t)y * gstride + (size_t)(lx + x) * CN;
    int dh = 0, dv = 0;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        if (x + 1 < lw) { const int t = (int)p[c] - (int)p[CN + c]; dh += t * t; }
        if (y + 1 < H) { const int t = (int)p[c] - (int)p[gstride + c]; dv += t * t; }
    }
    ch[(size_t)y * lw + x] = x + 1 < lw ? lut[dh] : 0.f;
    cv[(size_t)y * lw + x] = y + 1 < H ? lut[dv] : 0.f;
}

// One Thomas step, float32, fixed operation order (shared with oracle/prepost_oracle.py:_fgs_pass):
//   a = lam*C[j-1]; c = lam*C[j]; denom = ((1 - a) - c) - a*cc[j-1]; cc[j] = c/denom; f[j] = (f[j] - a*f[j-1])/denom
struct ThomasState {
    float cprev = 0.f, ccprev = 0.f, f0 = 0.f, f1 = 0.f;
};
__device__ __forceinline__ void thomas_fwd(ThomasState &s, float lam, float craw, float &v0, float &v1, float &cc) {
    const float a = lam * s.cprev, c = lam * craw;
    const float denom = ((1.0f - a) - c) - a * s.ccprev;
    cc = c / denom;
    v0 = (v0 - a * s.f0) / denom;
    v1 = (v1 - a * s.f1) / denom;
    s.cprev = craw;
    s.ccprev = cc;
    s.f0 = v0;
    s.f1 = v1;
}

#define FGS_TW 32  // tile width (columns per chunk) of the horizontal pass
#define FGS_TP 33  // padded pitch

// Horizontal pass: one wave per 64 image rows, lane = row.  Chunks of 32 columns go global -> registers -> LDS tile
// (coalesced 128-B row segments), the lane walks its row inside the tile, results go back the same way; the next
// chunk's loads are issued before the current chunk's chain so that the solve hides their latency.
__global__ __launch_bounds__(64) void k_fgs_h(const float *__restrict__ C, float *__restrict__ cc_out, float *__restrict__ s0,
                                              float *__restrict__ s1, int W, int H, float lam) {
    __shared__ float tC[64 * FGS_TP], t0[64 * FGS_TP], t1[64 * FGS_TP];
    const int lane = threadIdx.x, r0 = blockIdx.x * 64;
    const int half = lane >> 5, col = lane & 31;  // load shape: two rows per instruction, 32 columns each
    const int nchunk = (W + FGS_TW - 1) / FGS_TW;
    float rc[32], ra[32], rb[32];

    auto load_regs = [&](const float *pc, int x0) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int row = r0 + 2 * i + half, x = x0 + col;
            const bool ok = row < H && x < W;
            const size_t o = (size_t)row * W + x;
            rc[i] = ok ? pc[o] : 0.f;
            ra[i] = ok ? s0[o] : 0.f;
            rb[i] = ok ? s1[o] : 0.f;
        }
    };
    auto regs_to_lds = [&]() {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int o = (2 * i + half) * FGS_TP + col;
            tC[o] = rc[i];
            t0[o] = ra[i];
            t1[o] = rb[i];
        }
    };
    auto lds_to_global = [&](int x0, bool with_cc) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int row = r0 + 2 * i + half, x = x0 + col;
            if (row < H && x < W) {
                const size_t o = (size_t)row * W + x;
                const int t = (2 * i + half) * FGS_TP + col;
                if (with_cc) cc_out[o] = tC[t];
                s0[o] = t0[t];
                s1[o] = t1[t];
            }
        }
    };

    // forward elimination, left to right
    ThomasState st;
    load_regs(C, 0);
    regs_to_lds();
    __syncthreads();
    for (int k = 0; k < nchunk; k++) {
        const int x0 = k * FGS_TW, n = min(FGS_TW, W - x0);
        if (k + 1 < nchunk) load_regs(C, x0 + FGS_TW);
        for (int j = 0; j < n; j++) {
            const int t = lane * FGS_TP + j;
            float v0 = t0[t], v1 = t1[t], cc;
            thomas_fwd(st, lam, tC[t], v0, v1, cc);
            tC[t] = cc;
            t0[t] = v0;
            t1[t] = v1;
        }
        __syncthreads();
        lds_to_global(x0, true);
        __syncthreads();
        if (k + 1 < nchunk) {
            regs_to_lds();
            __syncthreads();
        }
    }
    // back substitution, right to left: f[j] -= cc[j]*f[j+1]  (cc of the last column is 0)
    __threadfence_block();
    float n0 = 0.f, n1 = 0.f;
    load_regs(cc_out, (nchunk - 1) * FGS_TW);
    regs_to_lds();
    __syncthreads();
    for (int k = nchunk - 1; k >= 0; k--) {
        const int x0 = k * FGS_TW, n = min(FGS_TW, W - x0);
        if (k > 0) load_regs(cc_out, x0 - FGS_TW);
        for (int j = n - 1; j >= 0; j--) {
            const int t = lane * FGS_TP + j;
            const float cc = tC[t];
            n0 = t0[t] - cc * n0;
            n1 = t1[t] - cc * n1;
            t0[t] = n0;
            t1[t] = n1;
        }
        __syncthreads();
        lds_to_global(x0, false);
        __syncthreads();
        if (k > 0) {
            regs_to_lds();
            __syncthreads();
        }
    }
}

// Vertical pass: lane = column (coalesced rows), rows walked in blocks of 8 with the next block's loads in flight.
#define FGS_VB 8
__global__ __launch_bounds__(64) void k_fgs_v(const float *__restrict__ C, float *__restrict__ cc_out, float *__restrict__ s0,
                                              float *__restrict__ s1, int W, int H, float lam) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= W) return;
    const int nblk = (H + FGS_VB - 1) / FGS_VB;
    float c[FGS_VB], a[FGS_VB], b[FGS_VB], nc[FGS_VB], na[FGS_VB], nb[FGS_VB];
    auto load = [&](const float *pc, int y0, float *qc, float *qa, float *qb) {
#pragma unroll
        for (int i = 0; i < FGS_VB; i++) {
            const int y = y0 + i;
            const size_t o = (size_t)y * W + x;
            qc[i] = y < H ? pc[o] : 0.f;
            qa[i] = y < H ? s0[o] : 0.f;
            qb[i] = y < H ? s1[o] : 0.f;
        }
    };
    ThomasState st;
    load(C, 0, c, a, b);
    for (int k = 0; k < nblk; k++) {
        const int y0 = k * FGS_VB;
        if (k + 1 < nblk) load(C, y0 + FGS_VB, nc, na, nb);
#pragma unroll
        for (int i = 0; i < FGS_VB; i++) {
            if (y0 + i < H) {
                float cc;
                thomas_fwd(st, lam, c[i], a[i], b[i], cc);
                const size_t o = (size_t)(y0 + i) * W + x;
                cc_out[o] = cc;
                s0[o] = a[i];
                s1[o] = b[i];
            }
        }
#pragma unroll
        for (int i = 0; i < FGS_VB; i++) { c[i] = nc[i]; a[i] = na[i]; b[i] = nb[i]; }
    }
    float n0 = 0.f, n1 = 0.f;
    load(cc_out, (nblk - 1) * FGS_VB, c, a, b);
    for (int k = nblk - 1; k >= 0; k--) {
        const int y0 = k * FGS_VB;
        if (k > 0) load(cc_out, y0 - FGS_VB, nc, na, nb);
#pragma unroll
        for (int i = FGS_VB - 1; i >= 0; i--) {
            if (y0 + i < H) {
                n0 = a[i] - c[i] * n0;
                n1 = b[i] - c[i] * n1;
                const size_t o = (size_t)(y0 + i) * W + x;
                s0[o] = n0;
                s1[o] = n1;
            }
        }
#pragma unroll
        for (int i = 0; i < FGS_VB; i++) { c[i] = nc[i]; a[i] = na[i]; b[i] = nb[i]; }
    }
}

// ---- partitioned solver (default): the same tridiagonal systems, solved block-parallel -----------------------
// Every 32nd unknown of a line (row for the horizontal pass, column for the vertical one) is a separator; the 31
// unknowns between two separators form a block.  Given the separator values the blocks decouple, and the separators
// satisfy a tridiagonal Schur-complement system of their own:
//   A: per block, local Thomas solves of  T y = f (both signals),  T v = a_first e_first,  T w = c_last e_last;
//      only the end values go to memory (record of 12 floats per line and block)
//   R: per line, the reduced tridiagonal system over the separators (<= N/32 unknowns), both signals
//   B: per block, f_first -= a_first*S_left, f_last -= c_last*S_right, local Thomas solve, result written in place
// Exact in exact arithmetic; in float32 it differs from the sequential order by rounding only (tests state the bound).
// A and B work on 64-line x 32-unknown tiles in LDS: 4 000 waves at 8 MP instead of 39, global traffic 8 floats/pixel.
#define PT 32
#define PREC 12  // record fields: yF1 yL1 yF2 yL2 vF vL wF wL a_s c_s fs1 fs2
#define TILE_F (64 * 33)

template <bool VERT>
__device__ __forceinline__ int lidx(int lane, int j) { return VERT ? j * 64 + lane : lane * 33 + j; }

// tile of 64 lines x 32 unknowns starting at unknown p of lines line0..: global (pitch W, H rows) -> LDS
template <bool VERT>
__device__ __forceinline__ void tile_load(const float *__restrict__ g, float *t, int W, int H, int line0, int p, int lane) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
        int row, col, o;
        if (VERT) { row = p + i; col = line0 + lane; o = i * 64 + lane; }
        else { row = line0 + 2 * i + (lane >> 5); col = p + (lane & 31); o = (2 * i + (lane >> 5)) * 33 + (lane & 31); }
        // clamped address + bit mask instead of a select: a select gets turned into a branch around the load, and a
        // branch per load serialises the 96 loads of a tile (one s_waitcnt vmcnt(0) each)
        const float v = g[(size_t)min(row, H - 1) * W + min(col, W - 1)];
        t[o] = __int_as_float(__float_as_int(v) & ((row < H && col < W) ? -1 : 0));
    }
}
template <bool VERT>
__device__ __forceinline__ void tile_store(float *__restrict__ g, const float *t, int W, int H, int line0, int p, int lane) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
        int row, col, o;
        if (VERT) { row = p + i; col = line0 + lane; o = i * 64 + lane; }
        else { row = line0 + 2 * i + (lane >> 5); col = p + (lane & 31); o = (2 * i + (lane >> 5)) * 33 + (lane & 31); }
        if (row < H && col < W) g[(size_t)row * W + col] = t[o];
    }
}

// The block solves keep a line's 32 tile values in registers (static indexing, fully unrolled; n = block length is
// wave-uniform), so that the elimination chain never waits on LDS or memory; one IEEE division per step gives 1/denom,
// the right-hand sides are multiplied by it.
struct BlockEnds {
    float yF1, yL1, yF2, yL2, vF, vL, wF, wL, a_s, c_s, fs1, fs2;
};

// c[j]: raw weights (c[31]: the separator's), y1/y2: the two signals.  cl = raw weight left of the block.
// Blocks are always solved at full length: positions past the end of a line hold zero weights and zero data (tile loads
// pad with 0), which makes them decoupled identity equations (the last real weight of a line is 0 by construction).
__device__ __forceinline__ void block_solve_A(float (&c)[32], float (&y1)[32], float (&y2)[32], float cl, float lam, BlockEnds &e) {
    float v[31];
    const float a0 = lam * cl;
    float cprev = cl, ccprev = 0.f, y1p = 0.f, y2p = 0.f, vp = 0.f, rinv = 1.f;
#pragma unroll
    for (int j = 0; j < 31; j++) {
        const float craw = c[j], a = lam * cprev, cj = lam * craw;
        const float diag = (1.0f - a) - cj;
        const float denom = j > 0 ? diag - a * ccprev : diag;
        rinv = 1.0f / denom;
        y1p = (j > 0 ? y1[j] - a * y1p : y1[j]) * rinv;
        y2p = (j > 0 ? y2[j] - a * y2p : y2[j]) * rinv;
        vp = (j > 0 ? 0.f - a * vp : a0) * rinv;
        ccprev = j < 30 ? cj * rinv : 0.f;
        c[j] = ccprev;
        y1[j] = y1p;
        y2[j] = y2p;
        v[j] = vp;
        cprev = craw;
    }
    const float c_last = lam * cprev;  // coupling of the block's last unknown to its separator
    float wn = c_last * rinv;
    e.yL1 = y1p; e.yL2 = y2p; e.vL = vp; e.wL = wn;
#pragma unroll
    for (int j = 29; j >= 0; j--) {
        const float cc = c[j];
        y1p = y1[j] - cc * y1p;
        y2p = y2[j] - cc * y2p;
        vp = v[j] - cc * vp;
        wn = 0.f - cc * wn;
    }
    e.yF1 = y1p; e.yF2 = y2p; e.vF = vp; e.wF = wn;
    e.a_s = c_last;
    e.c_s = lam * c[31];
    e.fs1 = y1[31];
    e.fs2 = y2[31];
}

// sl*/sr*: separator values left / right of the block (0 where there is none); result in y1/y2[0..31), separator copied to [31]
__device__ __forceinline__ void block_solve_B(float (&c)[32], float (&y1)[32], float (&y2)[32], float cl, float lam, float sl1,
                                              float sl2, float sr1, float sr2) {
    const float a0 = lam * cl;
    float cprev = cl, ccprev = 0.f, y1p = 0.f, y2p = 0.f;
#pragma unroll
    for (int j = 0; j < 31; j++) {
        const float craw = c[j], a = lam * cprev, cj = lam * craw;
        const float diag = (1.0f - a) - cj;
        const float denom = j > 0 ? diag - a * ccprev : diag;
        const float rinv = 1.0f / denom;
        float f1 = y1[j], f2 = y2[j];
        if (j == 30) {  // coupling to the right separator moves to the right-hand side
            f1 = f1 - cj * sr1;
            f2 = f2 - cj * sr2;
        }
        y1p = (j > 0 ? f1 - a * y1p : f1 - a0 * sl1) * rinv;
        y2p = (j > 0 ? f2 - a * y2p : f2 - a0 * sl2) * rinv;
        ccprev = j < 30 ? cj * rinv : 0.f;
        c[j] = ccprev;
        y1[j] = y1p;
        y2[j] = y2p;
        cprev = craw;
    }
#pragma unroll
    for (int j = 29; j >= 0; j--) {
        y1p = y1[j] - c[j] * y1p;
        y2p = y2[j] - c[j] * y2p;
        y1[j] = y1p;
        y2[j] = y2p;
    }
    y1[31] = sr1;
    y2[31] = sr2;
}

template <bool VERT>
__device__ __forceinline__ float left_weight(const float *__restrict__ C, int W, int p, int line) {
    return p > 0 ? (VERT ? C[(size_t)(p - 1) * W + line] : C[(size_t)line * W + p - 1]) : 0.f;
}

// horizontal pass, phase A: tiles transposed through LDS (coalesced 128-B row segments), then one row per lane
__global__ __launch_bounds__(64) void k_fgs_hA(const float *__restrict__ C, const float *__restrict__ s0, const float *__restrict__ s1,
                                               float *__restrict__ rec, int W, int H, float lam) {
    __shared__ float tC[TILE_F], t0[TILE_F], t1[TILE_F];
    const int lane = threadIdx.x, k = blockIdx.x, line0 = blockIdx.y * 64;
    const int p = k * PT;
    tile_load<false>(C, tC, W, H, line0, p, lane);
    tile_load<false>(s0, t0, W, H, line0, p, lane);
    tile_load<false>(s1, t1, W, H, line0, p, lane);
    __syncthreads();
    const int line = line0 + lane;
    if (line >= H) return;
    float c[32], y1[32], y2[32];
#pragma unroll
    for (int j = 0; j < 32; j++) { c[j] = tC[lane * 33 + j]; y1[j] = t0[lane * 33 + j]; y2[j] = t1[lane * 33 + j]; }
    BlockEnds e;
    block_solve_A(c, y1, y2, left_weight<false>(C, W, p, line), lam, e);
    float *r = rec + ((size_t)k * PREC) * H + line;
    const size_t NL = H;
    r[0 * NL] = e.yF1; r[1 * NL] = e.yL1; r[2 * NL] = e.yF2; r[3 * NL] = e.yL2; r[4 * NL] = e.vF; r[5 * NL] = e.vL;
    r[6 * NL] = e.wF; r[7 * NL] = e.wL; r[8 * NL] = e.a_s; r[9 * NL] = e.c_s; r[10 * NL] = e.fs1; r[11 * NL] = e.fs2;
}

// vertical pass, phase A: one column per lane, the 32 rows of the block straight from global memory into registers
__global__ __launch_bounds__(64) void k_fgs_vA(const float *__restrict__ C, const float *__restrict__ s0, const float *__restrict__ s1,
                                               float *__restrict__ rec, int W, int H, float lam) {
    const int k = blockIdx.x, line = blockIdx.y * 64 + threadIdx.x;
    const int p = k * PT;
    if (line >= W) return;
    float c[32], y1[32], y2[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int m = p + j < H ? -1 : 0;
        const size_t o = (size_t)min(p + j, H - 1) * W + line;  // clamped address + bit mask, see tile_load
        c[j] = __int_as_float(__float_as_int(C[o]) & m);
        y1[j] = __int_as_float(__float_as_int(s0[o]) & m);
        y2[j] = __int_as_float(__float_as_int(s1[o]) & m);
    }
    BlockEnds e;
    block_solve_A(c, y1, y2, left_weight<true>(C, W, p, line), lam, e);
    float *r = rec + ((size_t)k * PREC) * W + line;
    const size_t NL = W;
    r[0 * NL] = e.yF1; r[1 * NL] = e.yL1; r[2 * NL] = e.yF2; r[3 * NL] = e.yL2; r[4 * NL] = e.vF; r[5 * NL] = e.vL;
    r[6 * NL] = e.wF; r[7 * NL] = e.wL; r[8 * NL] = e.a_s; r[9 * NL] = e.c_s; r[10 * NL] = e.fs1; r[11 * NL] = e.fs2;
}

// coefficients of the reduced (separator) system, one thread per (line, separator):
//   coef[(k*5 + {ra, rb, rc, r1, r2}) * NL + line]
__global__ void k_fgs_pRc(const float *__restrict__ rec, float *__restrict__ coef, int N, int NL) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (line >= NL) return;
    const int nb = (N + PT - 1) / PT;
    const size_t L = NL;
    const float *r = rec + ((size_t)k * PREC) * L + line;
    const float a_s = r[8 * L], c_s = r[9 * L];
    float r1 = r[10 * L] - a_s * r[1 * L], r2 = r[11 * L] - a_s * r[3 * L];
    const float ra = 0.f - a_s * r[5 * L];
    float rb = ((1.0f - a_s) - c_s) - a_s * r[7 * L], rc = 0.f;
    if (k + 1 < nb) {
        const float *rn = r + (size_t)PREC * L;
        rb = rb - c_s * rn[4 * L];
        rc = 0.f - c_s * rn[6 * L];
        r1 = r1 - c_s * rn[0 * L];
        r2 = r2 - c_s * rn[2 * L];
    }
    float *o = coef + ((size_t)k * 5) * L + line;
    o[0] = ra; o[L] = rb; o[2 * L] = rc; o[3 * L] = r1; o[4 * L] = r2;
}

// reduced tridiagonal system over the separators of each line (one lane per line, P = N/32 unknowns);
// sol[(k*3 + {0: cc, 1: S of signal 0, 2: S of signal 1}) * NL + line].  Loads run one chunk of 8 steps ahead of the chain.
#define RCH 8
__global__ __launch_bounds__(64) void k_fgs_pR(const float *__restrict__ coef, float *__restrict__ sol, int N, int NL) {
    const int line = blockIdx.x * 64 + threadIdx.x;
    if (line >= NL) return;
    const int P = N / PT;
    const size_t L = NL;
    float cur[RCH][5], nxt[RCH][5];
    auto load = [&](int k0, float (&d)[RCH][5]) {
#pragma unroll
        for (int i = 0; i < RCH; i++)
#pragma unroll
            for (int f = 0; f < 5; f++) d[i][f] = coef[((size_t)min(k0 + i, P - 1) * 5 + f) * L + line];  // clamped, never a branch
    };
    float ccprev = 0.f, s1p = 0.f, s2p = 0.f;
    load(0, cur);
    for (int k0 = 0; k0 < P; k0 += RCH) {
        if (k0 + RCH < P) load(k0 + RCH, nxt);
#pragma unroll
        for (int i = 0; i < RCH; i++) {
            if (k0 + i < P) {
                const float ra = cur[i][0], denom = cur[i][1] - ra * ccprev, rinv = 1.0f / denom;
                ccprev = cur[i][2] * rinv;
                s1p = (cur[i][3] - ra * s1p) * rinv;
                s2p = (cur[i][4] - ra * s2p) * rinv;
                float *o = sol + ((size_t)(k0 + i) * 3) * L + line;
                o[0] = ccprev; o[L] = s1p; o[2 * L] = s2p;
            }
        }
#pragma unroll
        for (int i = 0; i < RCH; i++)
#pragma unroll
            for (int f = 0; f < 5; f++) cur[i][f] = nxt[i][f];
    }
    // back substitution: S[k] -= cc[k]*S[k+1]; the last separator is final already
    float b[RCH][3], nb3[RCH][3];
    auto loadb = [&](int khi, float (&d)[RCH][3]) {  // steps khi, khi-1, ...
#pragma unroll
        for (int i = 0; i < RCH; i++)
#pragma unroll
            for (int f = 0; f < 3; f++) d[i][f] = sol[((size_t)max(khi - i, 0) * 3 + f) * L + line];
    };
    loadb(P - 2, b);
    for (int khi = P - 2; khi >= 0; khi -= RCH) {
        if (khi - RCH >= 0) loadb(khi - RCH, nb3);
#pragma unroll
        for (int i = 0; i < RCH; i++) {
            if (khi - i >= 0) {
                s1p = b[i][1] - b[i][0] * s1p;
                s2p = b[i][2] - b[i][0] * s2p;
                float *o = sol + ((size_t)(khi - i) * 3) * L + line;
                o[L] = s1p; o[2 * L] = s2p;
            }
        }
#pragma unroll
        for (int i = 0; i < RCH; i++)
#pragma unroll
            for (int f = 0; f < 3; f++) b[i][f] = nb3[i][f];
    }
}

template <bool VERT>
__device__ __forceinline__ void sep_values(const float *__restrict__ sol, int k, bool has_sep, int NL, int line, float &sl1, float &sl2,
                                           float &sr1, float &sr2) {
    sl1 = sl2 = sr1 = sr2 = 0.f;
    const size_t L = NL;
    if (k > 0) {
        const float *o = sol + ((size_t)(k - 1) * 3) * L + line;
        sl1 = o[L];
        sl2 = o[2 * L];
    }
    if (has_sep) {
        const float *o = sol + ((size_t)k * 3) * L + line;
        sr1 = o[L];
        sr2 = o[2 * L];
    }
}

__global__ __launch_bounds__(64) void k_fgs_hB(const float *__restrict__ C, float *__restrict__ s0, float *__restrict__ s1,
                                               const float *__restrict__ sol, int W, int H, float lam) {
    __shared__ float tC[TILE_F], t0[TILE_F], t1[TILE_F];
    const int lane = threadIdx.x, k = blockIdx.x, line0 = blockIdx.y * 64;
    const int p = k * PT, q = min(p + PT - 1, W);
    tile_load<false>(C, tC, W, H, line0, p, lane);
    tile_load<false>(s0, t0, W, H, line0, p, lane);
    tile_load<false>(s1, t1, W, H, line0, p, lane);
    __syncthreads();
    const int line = line0 + lane;
    if (line < H) {
        float c[32], y1[32], y2[32];
#pragma unroll
        for (int j = 0; j < 32; j++) { c[j] = tC[lane * 33 + j]; y1[j] = t0[lane * 33 + j]; y2[j] = t1[lane * 33 + j]; }
        float sl1, sl2, sr1, sr2;
        sep_values<false>(sol, k, q < W, H, line, sl1, sl2, sr1, sr2);
        block_solve_B(c, y1, y2, left_weight<false>(C, W, p, line), lam, sl1, sl2, sr1, sr2);
#pragma unroll
        for (int j = 0; j < 32; j++) { t0[lane * 33 + j] = y1[j]; t1[lane * 33 + j] = y2[j]; }
    }
    __syncthreads();
    tile_store<false>(s0, t0, W, H, line0, p, lane);
    tile_store<false>(s1, t1, W, H, line0, p, lane);
}

__global__ __launch_bounds__(64) void k_fgs_vB(const float *__restrict__ C, float *__restrict__ s0, float *__restrict__ s1,
                                               const float *__restrict__ sol, int W, int H, float lam) {
    const int k = blockIdx.x, line = blockIdx.y * 64 + threadIdx.x;
    const int p = k * PT, q = min(p + PT - 1, H);
    if (line >= W) return;
    float c[32], y1[32], y2[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int m = p + j < H ? -1 : 0;
        const size_t o = (size_t)min(p + j, H - 1) * W + line;  // clamped address + bit mask, see tile_load
        c[j] = __int_as_float(__float_as_int(C[o]) & m);
        y1[j] = __int_as_float(__float_as_int(s0[o]) & m);
        y2[j] = __int_as_float(__float_as_int(s1[o]) & m);
    }
    float sl1, sl2, sr1, sr2;
    sep_values<true>(sol, k, q < H, W, line, sl1, sl2, sr1, sr2);
    block_solve_B(c, y1, y2, left_weight<true>(C, W, p, line), lam, sl1, sl2, sr1, sr2);
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (p + j < H) {
            const size_t o = (size_t)(p + j) * W + line;
            s0[o] = y1[j];
            s1[o] = y2[j];
        }
    }
}

// out = short(round_half_even(sig0 * (1 / (sig1 + 1e-5)))) inside the ROI, 16*(minD-1) elsewhere
__global__ void k_wls_finish(const float *__restrict__ sig0, const float *__restrict__ sig1, int W, int H, int lx, int lw,
                             int fill, int16_t *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    int v = fill;
    if (x >= lx && x < lx + lw) {
        const size_t o = (size_t)y * lw + (x - lx);
        const float q = sig0[o] * (1.0f / (sig1[o] + 0.00001f));
        v = (int)fminf(fmaxf(rintf(q), -32768.f), 32767.f);
    }
    out[(size_t)y * W + x] = (int16_t)v;
}

int check_ctx(r3d_ctx *ctx) {
    if (!ctx) return R3D_E_BADARG;
    (void)hipSetDevice(ctx->device);
    return R3D_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------- C ABI

extern "C" int r3d_init_undistort_rectify_map(r3d_ctx *ctx, const double *camera3x3, const double *dist, int32_t n_dist,
                                              const double *R3x3, const double *new_camera, int32_t new_camera_cols, int32_t w,
                                              int32_t h, int16_t *map1, uint16_t *map2) {
    R3D_ROCTX_RANGE("r3d_init_undistort_rectify_map");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!camera3x3 || !new_camera || !map1 || !map2 || w <= 0 || h <= 0 || (new_camera_cols != 3 && new_camera_cols != 4))
        return r3d_fail(ctx, R3D_E_BADARG, "init_undistort_rectify_map: bad argument");
    if (n_dist != 0 && n_dist != 4 && n_dist != 5 && n_dist != 8 && n_dist != 12 && n_dist != 14)
        return r3d_fail(ctx, R3D_E_BADARG, "init_undistort_rectify_map: %d distortion coefficients (want 0/4/5/8/12/14)", n_dist);
    double d[14] = {0};
    for (int i = 0; i < n_dist; i++) d[i] = dist[i];
    if (d[12] != 0 || d[13] != 0) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "init_undistort_rectify_map: tilted sensor model");
    RectParams P;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R3x3) std::copy(R3x3, R3x3 + 9, R);
    double M[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double *a = new_camera + (size_t)i * new_camera_cols;
            M[i * 3 + j] = a[0] * R[j] + a[1] * R[3 + j] + a[2] * R[6 + j];
        }
    {  // closed-form 3x3 inverse (cofactors / det), the form cv::invert takes for n = 3
        const double *S = M;
        double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
        if (det == 0) return r3d_fail(ctx, R3D_E_BADARG, "init_undistort_rectify_map: singular P*R");
        det = 1. / det;
        P.ir[0] = (S[4] * S[8] - S[5] * S[7]) * det;
        P.ir[1] = (S[2] * S[7] - S[1] * S[8]) * det;
        P.ir[2] = (S[1] * S[5] - S[2] * S[4]) * det;
        P.ir[3] = (S[5] * S[6] - S[3] * S[8]) * det;
        P.ir[4] = (S[0] * S[8] - S[2] * S[6]) * det;
        P.ir[5] = (S[2] * S[3] - S[0] * S[5]) * det;
        P.ir[6] = (S[3] * S[7] - S[4] * S[6]) * det;
        P.ir[7] = (S[1] * S[6] - S[0] * S[7]) * det;
        P.ir[8] = (S[0] * S[4] - S[1] * S[3]) * det;
    }
    const int order[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};  // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    for (int i = 0; i < 12; i++) P.k[i] = d[order[i]];
    P.u0 = camera3x3[2];
    P.v0 = camera3x3[5];
    P.fx = camera3x3[0];
    P.fy = camera3x3[4];
    PPArena ar(ctx);
    const size_t n = (size_t)w * h;
    int16_t *d1 = (int16_t *)ar.get(n * 4);
    uint16_t *d2 = (uint16_t *)ar.get(n * 2);
    if (ar.rc) return ar.rc;
    k_rectify_map<<<(h + 63) / 64, 64, 0, ctx->stream>>>(P, w, h, d1, d2);
    R3D_HIP(ctx, hipGetLastError());
    R3D_HIP(ctx, hipMemcpyAsync(map1, d1, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(map2, d2, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

extern "C" int r3d_remap_u8_dev(r3d_ctx *ctx, const uint8_t *d_src, int32_t sw, int32_t sh, int32_t sstride, int32_t cn,
                                const int16_t *d_map1, const uint16_t *d_map2, int32_t dw, int32_t dh, int32_t border_value,
                                uint8_t *d_dst, uint8_t *d_gray) {
    R3D_ROCTX_RANGE("r3d_remap_u8_dev");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!d_src || !d_map1 || !d_map2 || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || sstride < sw * cn)
        return r3d_fail(ctx, R3D_E_BADARG, "remap: bad argument");
    if (cn != 1 && cn != 3 && cn != 4) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "remap: %d channels (1, 3 or 4)", cn);
    if (d_gray && cn < 3) return r3d_fail(ctx, R3D_E_BADARG, "remap: grey output needs a BGR(A) source");
    if (dh > 65535) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "remap: more than 65535 rows");
    const dim3 grid((dw + 255) / 256, dh), block(256);
    if (cn == 1) k_remap<1><<<grid, block, 0, ctx->stream>>>(d_src, sw, sh, sstride, d_map1, d_map2, dw, dh, border_value, d_dst, d_gray);
    if (cn == 3) k_remap<3><<<grid, block, 0, ctx->stream>>>(d_src, sw, sh, sstride, d_map1, d_map2, dw, dh, border_value, d_dst, d_gray);
    if (cn == 4) k_remap<4><<<grid, block, 0, ctx->stream>>>(d_src, sw, sh, sstride, d_map1, d_map2, dw, dh, border_value, d_dst, d_gray);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

extern "C" int r3d_remap_u8(r3d_ctx *ctx, const uint8_t *src, int32_t sw, int32_t sh, int32_t sstride, int32_t cn,
                            const int16_t *map1, const uint16_t *map2, int32_t dw, int32_t dh, int32_t border_value, uint8_t *dst,
                            uint8_t *gray) {
    R3D_ROCTX_RANGE("r3d_remap_u8");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!src || !map1 || !map2 || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || cn < 1 || cn > 4 || sstride < sw * cn)
        return r3d_fail(ctx, R3D_E_BADARG, "remap: bad argument");
    PPArena ar(ctx);
    const size_t ns = (size_t)sstride * sh, nd = (size_t)dw * dh;
    uint8_t *ds = (uint8_t *)ar.get(ns), *dd = (uint8_t *)ar.get(nd * cn), *dg = (uint8_t *)ar.get(nd);
    int16_t *d1 = (int16_t *)ar.get(nd * 4);
    uint16_t *d2 = (uint16_t *)ar.get(nd * 2);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(ds, src, ns, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(d1, map1, nd * 4, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(d2, map2, nd * 2, hipMemcpyHostToDevice, ctx->stream));
    const int rc = r3d_remap_u8_dev(ctx, ds, sw, sh, sstride, cn, d1, d2, dw, dh, border_value, dd, gray ? dg : nullptr);
    if (rc) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(dst, dd, nd * cn, hipMemcpyDeviceToHost, ctx->stream));
    if (gray) R3D_HIP(ctx, hipMemcpyAsync(gray, dg, nd, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

extern "C" int r3d_bgr2gray_dev(r3d_ctx *ctx, const uint8_t *d_bgr, int32_t w, int32_t h, int32_t stride, int32_t cn, uint8_t *d_gray) {
    R3D_ROCTX_RANGE("r3d_bgr2gray_dev");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!d_bgr || !d_gray || w <= 0 || h <= 0 || (cn != 3 && cn != 4) || stride < w * cn || h > 65535)
        return r3d_fail(ctx, R3D_E_BADARG, "bgr2gray: bad argument");
    k_bgr2gray<<<dim3((w + 255) / 256, h), 256, 0, ctx->stream>>>(d_bgr, w, h, stride, cn, d_gray);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

extern "C" int r3d_bgr2gray(r3d_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int32_t stride, int32_t cn, uint8_t *gray) {
    R3D_ROCTX_RANGE("r3d_bgr2gray");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!bgr || !gray || w <= 0 || h <= 0 || (cn != 3 && cn != 4) || stride < w * cn) return r3d_fail(ctx, R3D_E_BADARG, "bgr2gray: bad argument");
    PPArena ar(ctx);
    const size_t ns = (size_t)stride * h, nd = (size_t)w * h;
    uint8_t *ds = (uint8_t *)ar.get(ns), *dg = (uint8_t *)ar.get(nd);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(ds, bgr, ns, hipMemcpyHostToDevice, ctx->stream));
    const int rc = r3d_bgr2gray_dev(ctx, ds, w, h, stride, cn, dg);
    if (rc) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(gray, dg, nd, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

extern "C" int r3d_normalize_minmax_s16_dev(r3d_ctx *ctx, const int16_t *d_src, int64_t n, double alpha, double beta, int16_t *d_dst) {
    R3D_ROCTX_RANGE("r3d_normalize_minmax_s16_dev");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!d_src || !d_dst || n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "normalize: bad argument");
    r3d_buf &mmb = ctx->pp_minmax;
    const int rc = r3d_reserve(ctx, mmb, 16);
    if (rc) return rc;
    int *mm = (int *)mmb.p;
    const int nb = (int)std::min<int64_t>((n + 2047) / 2048, 1024);
    k_minmax_init<<<1, 1, 0, ctx->stream>>>(mm);
    k_minmax_s16<<<nb, 256, 0, ctx->stream>>>(d_src, n, mm);
    k_normalize_s16<<<nb, 256, 0, ctx->stream>>>(d_src, n, mm, std::min(alpha, beta), std::max(alpha, beta), d_dst);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

extern "C" int r3d_normalize_minmax_s16(r3d_ctx *ctx, const int16_t *src, int64_t n, double alpha, double beta, int16_t *dst) {
    R3D_ROCTX_RANGE("r3d_normalize_minmax_s16");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    if (!src || !dst || n <= 0) return r3d_fail(ctx, R3D_E_BADARG, "normalize: bad argument");
    PPArena ar(ctx);
    int16_t *ds = (int16_t *)ar.get((size_t)n * 2), *dd = (int16_t *)ar.get((size_t)n * 2);
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(ds, src, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
    const int rc = r3d_normalize_minmax_s16_dev(ctx, ds, n, alpha, beta, dd);
    if (rc) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(dst, dd, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

static int wls_run(r3d_ctx *ctx, PPArena &ar, const r3d_wls_params *p, const int16_t *d_dl, const int16_t *d_dr,
                   const uint8_t *d_guide, int gcn, int gstride, int w, int h, int16_t *d_out, float *d_conf) {
    const int lo = std::max(0, p->min_disparity + p->num_disparities), ro = std::max(0, -p->min_disparity);
    const int lw = w - lo - ro, lx = lo, rx = w - (lo + lw);
    const int fill = 16 * (p->min_disparity - 1);
    hipStream_t st = ctx->stream;
    const dim3 blk(256);
    if (d_conf) R3D_HIP(ctx, hipMemsetAsync(d_conf, 0, (size_t)w * h * 4, st));
    if (lw <= 0) {
        float *z = (float *)ar.get(16);
        if (ar.rc) return ar.rc;
        k_wls_finish<<<dim3((w + 255) / 256, h), blk, 0, st>>>(z, z, w, h, 0, 0, fill, d_out);
        R3D_HIP(ctx, hipGetLastError());
        return R3D_OK;
    }
    const size_t nfull = (size_t)w * h, nroi = (size_t)lw * h;
    float *ddl = (float *)ar.get(nfull * 4), *ddr = (float *)ar.get(nfull * 4);
    float *s0 = (float *)ar.get(nroi * 4), *s1 = (float *)ar.get(nroi * 4);
    float *ch = (float *)ar.get(nroi * 4), *cv = (float *)ar.get(nroi * 4), *cc = (float *)ar.get(nroi * 4);
    if (ar.rc) return ar.rc;
    // weights_LUT[i] = -exp(-sqrt(i)/sigma): parameter table, fp64 on the host, rounded once to float
    const int nlut = gcn * 255 * 255 + 1;
    if (ctx->pp_lut_sigma != p->sigma_color || ctx->pp_lut_n != nlut) {
        const int rc = r3d_reserve(ctx, ctx->pp_lut, (size_t)nlut * 4);
        if (rc) return rc;
        std::vector<float> hl((size_t)nlut);
        for (int i = 0; i < nlut; i++) hl[i] = (float)(-std::exp(-std::sqrt((double)i) / p->sigma_color));
        R3D_HIP(ctx, hipMemcpyAsync(ctx->pp_lut.p, hl.data(), (size_t)nlut * 4, hipMemcpyHostToDevice, st));
        R3D_HIP(ctx, hipStreamSynchronize(st));
        ctx->pp_lut_sigma = p->sigma_color;
        ctx->pp_lut_n = nlut;
    }
    const float *lut = (const float *)ctx->pp_lut.p;
    const dim3 groi((lw + 255) / 256, h), gdd((lw + 63) / 64, (h + DD_STRIP - 1) / DD_STRIP);
    const size_t ddsh = (size_t)(2 * p->discontinuity_radius + 1) * 64 * 12;
    for (int view = 0; view < 2; view++) {
        const int16_t *dv = view ? d_dr : d_dl;
        float *out = view ? ddr : ddl;
        const int vx = view ? rx : lx, r = p->discontinuity_radius;
        const float ro = (float)p->discontinuity_roll_off;
        switch (r) {  // radii of blockSize 1..11 (ceil(0.5*bs)) get unrolled tap loops
            case 1: k_wls_dd<1><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            case 2: k_wls_dd<2><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            case 3: k_wls_dd<3><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            case 4: k_wls_dd<4><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            case 5: k_wls_dd<5><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            case 6: k_wls_dd<6><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
            default: k_wls_dd<0><<<gdd, 64, ddsh, st>>>(dv, w, h, vx, lw, r, ro, out); break;
        }
    }
    k_wls_conf<<<groi, blk, 0, st>>>(d_dl, d_dr, ddl, ddr, w, h, lx, lw, rx, lw, p->lrc_thresh, s0, s1, d_conf);
    if (gcn == 1) k_fgs_weights<1><<<groi, blk, 0, st>>>(d_guide, gstride, lx, lw, h, lut, ch, cv);
    else k_fgs_weights<3><<<groi, blk, 0, st>>>(d_guide, gstride, lx, lw, h, lut, ch, cv);
    float lam = (float)p->lambda;
    if (p->solver == R3D_WLS_SOLVER_SEQUENTIAL) {
        for (int it = 0; it < p->num_iter; it++) {
            k_fgs_h<<<(h + 63) / 64, 64, 0, st>>>(ch, cc, s0, s1, lw, h, lam);
            k_fgs_v<<<(lw + 63) / 64, 64, 0, st>>>(cv, cc, s0, s1, lw, h, lam);
            lam = lam * (float)p->lambda_attenuation;
        }
    } else {
        // records, reduced-system coefficients and separator solutions: 12 + 5 + 3 floats per line and block
        const int nbh = (lw + PT - 1) / PT, nbv = (h + PT - 1) / PT, Ph = lw / PT, Pv = h / PT;
        const size_t nrec = std::max((size_t)nbh * h, (size_t)nbv * lw);
        float *rec = (float *)ar.get(nrec * PREC * 4), *coef = (float *)ar.get(nrec * 5 * 4), *sol = (float *)ar.get(nrec * 3 * 4);
        if (ar.rc) return ar.rc;
        for (int it = 0; it < p->num_iter; it++) {
            k_fgs_hA<<<dim3(nbh, (h + 63) / 64), 64, 0, st>>>(ch, s0, s1, rec, lw, h, lam);
            if (Ph > 0) {
                k_fgs_pRc<<<dim3((h + 255) / 256, Ph), 256, 0, st>>>(rec, coef, lw, h);
                k_fgs_pR<<<(h + 63) / 64, 64, 0, st>>>(coef, sol, lw, h);
            }
            k_fgs_hB<<<dim3(nbh, (h + 63) / 64), 64, 0, st>>>(ch, s0, s1, sol, lw, h, lam);
            k_fgs_vA<<<dim3(nbv, (lw + 63) / 64), 64, 0, st>>>(cv, s0, s1, rec, lw, h, lam);
            if (Pv > 0) {
                k_fgs_pRc<<<dim3((lw + 255) / 256, Pv), 256, 0, st>>>(rec, coef, h, lw);
                k_fgs_pR<<<(lw + 63) / 64, 64, 0, st>>>(coef, sol, h, lw);
            }
            k_fgs_vB<<<dim3(nbv, (lw + 63) / 64), 64, 0, st>>>(cv, s0, s1, sol, lw, h, lam);
            lam = lam * (float)p->lambda_attenuation;
        }
    }
    k_wls_finish<<<dim3((w + 255) / 256, h), blk, 0, st>>>(s0, s1, w, h, lx, lw, fill, d_out);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

static int wls_check(r3d_ctx *ctx, const r3d_wls_params *p, int gcn, int w, int h) {
    if (!p || w <= 0 || h <= 0) return r3d_fail(ctx, R3D_E_BADARG, "wls_filter: bad argument");
    if (gcn != 1 && gcn != 3) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "wls_filter: guide with %d channels (1 or 3)", gcn);
    if (!(p->lambda >= 0) || !(p->sigma_color > 0) || p->num_iter < 1 || p->num_iter > 16 || p->discontinuity_radius < 0 ||
        p->discontinuity_radius > 32 || p->num_disparities < 0 || (p->solver != R3D_WLS_SOLVER_PARTITIONED && p->solver != R3D_WLS_SOLVER_SEQUENTIAL))
        return r3d_fail(ctx, R3D_E_BADARG, "wls_filter: bad parameter");
    if (h > 65535) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "wls_filter: more than 65535 rows");
    return R3D_OK;
}

extern "C" int r3d_wls_filter_dev(r3d_ctx *ctx, const r3d_wls_params *p, const int16_t *d_disp_left, const int16_t *d_disp_right,
                                  const uint8_t *d_guide, int32_t guide_cn, int32_t guide_stride, int32_t w, int32_t h,
                                  int16_t *d_out, float *d_confidence) {
    R3D_ROCTX_RANGE("r3d_wls_filter_dev");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    int rc = wls_check(ctx, p, guide_cn, w, h);
    if (rc) return rc;
    if (!d_disp_left || !d_disp_right || !d_guide || !d_out || guide_stride < w * guide_cn)
        return r3d_fail(ctx, R3D_E_BADARG, "wls_filter: bad argument");
    PPArena ar(ctx);
    return wls_run(ctx, ar, p, d_disp_left, d_disp_right, d_guide, guide_cn, guide_stride, w, h, d_out, d_confidence);
}

extern "C" int r3d_wls_filter(r3d_ctx *ctx, const r3d_wls_params *p, const int16_t *disp_left, const int16_t *disp_right,
                              const uint8_t *guide, int32_t guide_cn, int32_t guide_stride, int32_t w, int32_t h, int16_t *out,
                              float *confidence) {
    R3D_ROCTX_RANGE("r3d_wls_filter");
    if (check_ctx(ctx)) return R3D_E_BADARG;
    int rc = wls_check(ctx, p, guide_cn, w, h);
    if (rc) return rc;
    if (!disp_left || !disp_right || !guide || !out || guide_stride < w * guide_cn) return r3d_fail(ctx, R3D_E_BADARG, "wls_filter: bad argument");
    PPArena ar(ctx);
    const size_t n = (size_t)w * h, ng = (size_t)guide_stride * h;
    int16_t *dl = (int16_t *)ar.get(n * 2), *dr = (int16_t *)ar.get(n * 2), *dout = (int16_t *)ar.get(n * 2);
    uint8_t *dg = (uint8_t *)ar.get(ng);
    float *dc = confidence ? (float *)ar.get(n * 4) : nullptr;
    if (ar.rc) return ar.rc;
    R3D_HIP(ctx, hipMemcpyAsync(dl, disp_left, n * 2, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(dr, disp_right, n * 2, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(dg, guide, ng, hipMemcpyHostToDevice, ctx->stream));
    rc = wls_run(ctx, ar, p, dl, dr, dg, guide_cn, guide_stride, w, h, dout, dc);
    if (rc) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(out, dout, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (confidence) R3D_HIP(ctx, hipMemcpyAsync(confidence, dc, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}
