/*
 * oracle/sgbm3way.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the stereo matcher the reference drives through
 *   cv2.StereoSGBM_create(..., mode=cv2.STEREO_SGBM_MODE_SGBM_3WAY).compute(L, R)
 * (reference call sites: Calib_depth/depth1.py:202-214,331; depth2.py:146-158,251;
 *  depth3.py:234-246,339; depth4.py:156-168,254; depth_test.py:162-174,260).
 *
 * The arithmetic lives in a third-party dependency that is NOT vendored in
 * /root/reference and is not installed in this image: OpenCV 4.x,
 * modules/calib3d/src/stereosgbm.cpp (version un-pinned by the reference: it has
 * no requirements file).  This file restates that module's published algorithm:
 *   StereoSGBMImpl::compute -> computeDisparity3WAY -> SGBM3WayMainLoop
 *   (calcPixelCostBT, getRawMatchingCost, accumulateCostsLeftTop,
 *    accumulateCostsRight, uniqueness / disp2 / sub-pixel, pseudo LR check),
 *   then medianBlur(disp, 3) and, iff speckleWindowSize > 0, filterSpeckles.
 *
 * PARITY UNPINNED versus real OpenCV: the reference holds no stereo image pair
 * and no disparity map, and cv2 cannot be imported here or
 * on the GPU box.  The oracle is pinned only by analytic known-answer tests
 * (tests/test_sgbm_oracle.py).  Every behaviour that is an OpenCV quirk rather
 * than textbook SGM is marked QUIRK below so that it can be flipped once a box
 * with cv2 exists.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (lib3d_reconstruction_project_amd HIP library)
 * never calls it.
 *
 * Build: make -C oracle     (gcc -O3 -march=x86-64-v3 -fopenmp -shared)
 */
#include <limits.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int16_t cost_t;

typedef struct {
    int minDisparity;      /* StereoSGBM_create kwargs, same names as the reference passes */
    int numDisparities;
    int blockSize;
    int P1;
    int P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
} sgbm_oracle_params;

enum { DISP_SHIFT = 4, DISP_SCALE = 1 << DISP_SHIFT, NSTRIPES = 4 /* QUIRK: fixed, thread-count independent */ };

/* QUIRK_SMALL_IMAGE_STRIPES [recalled, computeDisparity3WAY / SGBM3WayMainLoop::operator()]: every stripe n writes row y of its
 * run at index dst_offset + (y - src_start) of a PRIVATE buffer of stripe_sz + overlap rows (dst_offset = overlap for n = 0, else
 * 0), and the map is assembled as out[i] = buffer[i / stripe_sz][overlap + i % stripe_sz].  That is consistent only while
 * src_start = n * stripe_sz - overlap is not clamped.  On very small images (stripe_sz < overlap: H <= 12 at blockSize 5) a
 * stripe n >= 1 has n * stripe_sz - overlap < 0: its run starts at row 0 with dst_offset 0, so buffer row r holds the disparity
 * of IMAGE row r of a run that started at row 0, and the assembly hands out[n * stripe_sz + j] = that run's row (overlap + j) --
 * a row from further down the image -- while buffer rows at or beyond the stripe's last source row were never written: the
 * original returns uninitialised memory there.  With the switch on (default) the oracle reproduces the shifted rows and writes
 * the invalid marker where the original's content is undefined (sgbm_oracle_undefined_rows reports those rows, so a comparison
 * with the real library can mask them and their median neighbours); off = every row at its own place (rounds 1-3). */
static int g_quirk_small_image_stripes = 1;
void sgbm_oracle_set_quirk_small_image_stripes(int on) { g_quirk_small_image_stripes = on != 0; }

#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))

static inline cost_t sat16(int v) { return (cost_t)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v)); }

/* ---- derived geometry (SGBM3WayMainLoop constructor) ------------------------------------ */
typedef struct {
    int W, H, minD, maxD, D, minX1, maxX1, W1, SW2, SH2, P1, P2, uniq, d12, ftzero;
    int stripe_sz, overlap;
} geom_t;

static int derive(const sgbm_oracle_params *p, int W, int H, geom_t *g) {
    if (p->numDisparities <= 0 || p->numDisparities % 16 != 0) return -1;
    if (p->blockSize < 1 || p->blockSize % 2 == 0) return -1;
    g->W = W; g->H = H;
    g->minD = p->minDisparity; g->maxD = g->minD + p->numDisparities; g->D = p->numDisparities;
    g->minX1 = IMAX(g->maxD, 0); g->maxX1 = W + IMIN(g->minD, 0); g->W1 = g->maxX1 - g->minX1;
    g->SW2 = g->SH2 = p->blockSize / 2;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    g->P2 = IMAX(p->P2 > 0 ? p->P2 : 5, g->P1 + 1);
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;   /* QUIRK: 3WAY never disables the LR check */
    g->ftzero = IMAX(p->preFilterCap, 15) | 1;
    g->stripe_sz = (H + NSTRIPES - 1) / NSTRIPES;
    /* stripe_overlap = (blockSize/2+1) + ceil(0.1*stripe_sz) */
    g->overlap = (p->blockSize / 2 + 1) + (g->stripe_sz + 9) / 10;
    if (g->W1 <= 0) return -2;
    return 0;
}

/* ---- calcPixelCostBT: one image row -> pix[xc*D + d], xc = x - minX1 ---------------------- */
/* Row scratch: 4 rows of W bytes for the left image (value, -, -, -) and, for the right image,
 * value / interval-min / interval-max stored MIRRORED (index W-1-x) so that the d loop walks
 * memory forwards -- the same trick the original uses to vectorise over d. */
typedef struct {
    uint8_t *l[2];      /* left  channel c: prefiltered gradient (c=0), raw intensity (c=1) */
    uint8_t *r[2];      /* right channel c, mirrored */
    uint8_t *r0[2], *r1[2]; /* right half-pixel interval min / max, mirrored */
} rowbuf_t;

static void pix_row(const uint8_t *L, const uint8_t *R, int ldL, int ldR, const geom_t *g, int y,
                    const uint8_t *tab /* tab[v], v in [-1024, 1279] */, rowbuf_t *rb, cost_t *pix) {
    const int W = g->W, H = g->H, D = g->D;
    const uint8_t *row1 = L + (size_t)y * ldL, *row2 = R + (size_t)y * ldR;
    /* rows above / below, clamped to the SAME row at the image top / bottom */
    const int n = y > 0 ? -1 : 0, s = y < H - 1 ? 1 : 0;
    const uint8_t *a1 = row1 + (ptrdiff_t)n * ldL, *b1 = row1 + (ptrdiff_t)s * ldL;
    const uint8_t *a2 = row2 + (ptrdiff_t)n * ldR, *b2 = row2 + (ptrdiff_t)s * ldR;

    /* QUIRK: columns 0 and W-1 of BOTH channels (gradient and raw intensity) are tab[0] = ftzero */
    for (int c = 0; c < 2; c++) {
        rb->l[c][0] = rb->l[c][W - 1] = tab[0];
        rb->r[c][0] = rb->r[c][W - 1] = tab[0];
    }
    for (int x = 1; x < W - 1; x++) {
        rb->l[0][x] = tab[(row1[x + 1] - row1[x - 1]) * 2 + a1[x + 1] - a1[x - 1] + b1[x + 1] - b1[x - 1]];
        rb->r[0][W - 1 - x] = tab[(row2[x + 1] - row2[x - 1]) * 2 + a2[x + 1] - a2[x - 1] + b2[x + 1] - b2[x - 1]];
        rb->l[1][x] = row1[x];
        rb->r[1][W - 1 - x] = row2[x];
    }
    memset(pix, 0, (size_t)g->W1 * D * sizeof(cost_t));

    for (int c = 0; c < 2; c++) {
        const int shift = c == 0 ? 0 : 2;          /* gradient full weight, raw intensity >> 2 */
        const uint8_t *pl = rb->l[c], *pr = rb->r[c];
        uint8_t *v0a = rb->r0[c], *v1a = rb->r1[c];
        for (int x = 0; x < W; x++) {              /* mirrored index; neighbours are symmetric */
            int v = pr[x];
            int vl = x > 0 ? (v + pr[x - 1]) / 2 : v;
            int vr = x < W - 1 ? (v + pr[x + 1]) / 2 : v;
            v0a[x] = (uint8_t)IMIN(IMIN(vl, vr), v);
            v1a[x] = (uint8_t)IMAX(IMAX(vl, vr), v);
        }
        for (int x = g->minX1; x < g->maxX1; x++) {
            int u = pl[x];
            int ul = x > 0 ? (u + pl[x - 1]) / 2 : u;
            int ur = x < W - 1 ? (u + pl[x + 1]) / 2 : u;
            int u0 = IMIN(IMIN(ul, ur), u), u1 = IMAX(IMAX(ul, ur), u);
            cost_t *dst = pix + (size_t)(x - g->minX1) * D;
            /* right column x - (minD + k)  <->  mirrored index W-1-x + minD + k */
            const uint8_t *pv = pr + (W - 1 - x + g->minD);
            const uint8_t *pv0 = v0a + (W - 1 - x + g->minD), *pv1 = v1a + (W - 1 - x + g->minD);
            for (int k = 0; k < D; k++) {
                int v = pv[k], v0 = pv0[k], v1 = pv1[k];
                int c0 = IMAX(IMAX(0, u - v1), v0 - u);
                int c1 = IMAX(IMAX(0, v - u1), u0 - v);
                dst[k] = (cost_t)(dst[k] + (IMIN(c0, c1) >> shift));
            }
        }
    }
}

/* horizontal box sum with replicated borders in COST coordinates [0, W1) */
static void hsum_row(const geom_t *g, const cost_t *pix, cost_t *hs) {
    const int D = g->D, W1 = g->W1, SW2 = g->SW2;
    for (int d = 0; d < D; d++) hs[d] = (cost_t)(pix[d] * (SW2 + 1));
    for (int i = 1; i <= SW2; i++) {
        const cost_t *p = pix + (size_t)IMIN(i, W1 - 1) * D;
        for (int d = 0; d < D; d++) hs[d] = (cost_t)(hs[d] + p[d]);
    }
    for (int x = 1; x < W1; x++) {
        const cost_t *add = pix + (size_t)IMIN(x + SW2, W1 - 1) * D;
        const cost_t *sub = pix + (size_t)IMAX(x - SW2 - 1, 0) * D;
        const cost_t *prev = hs + (size_t)(x - 1) * D;
        cost_t *cur = hs + (size_t)x * D;
        for (int d = 0; d < D; d++) cur[d] = (cost_t)(prev[d] + add[d] - sub[d]);
    }
}

/* one SGM step: out[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, minp+P2) - (minp+P2)
 * QUIRK: the subtrahend is (min + P2), not min, so aggregated costs run down to C - P2 (negative).
 * Lp is padded with SHRT_MAX at d=-1 and d=D.  Returns min_d out[d]. */
static inline int sgm_step(const cost_t *C, const cost_t *Lp /* padded, Lp[-1..D] */, cost_t *out, int D, int P1,
                           int minp_P2) {
    int mn = SHRT_MAX;
    for (int d = 0; d < D; d++) {
        int a = IMIN((int)Lp[d - 1], (int)Lp[d + 1]) + P1;
        int b = IMIN((int)Lp[d], minp_P2);
        int v = sat16(C[d] + IMIN(a, b) - minp_P2);
        out[d] = (cost_t)v;
        mn = IMIN(mn, v);
    }
    return mn;
}

/* ---- one stripe (SGBM3WayMainLoop::operator()) ------------------------------------------- */
static int run_stripe(const uint8_t *L, const uint8_t *R, int ldL, int ldR, const geom_t *g, int n,
                      const uint8_t *tab, int16_t *disp, int ldD) {
    const int W = g->W, H = g->H, D = g->D, W1 = g->W1, SH2 = g->SH2, P1 = g->P1, P2 = g->P2;
    const int INVALID = (g->minD - 1) * DISP_SCALE;
    const int src_start = IMAX(IMIN(n * g->stripe_sz - g->overlap, H), 0);
    const int src_end = IMIN((n + 1) * g->stripe_sz, H);
    const int out_start = IMIN(n * g->stripe_sz, H); /* rows < out_start only warm the vertical path up */
    /* QUIRK_SMALL_IMAGE_STRIPES: the run of this stripe was clamped to row 0; computed row y lands at out row y - shift_from */
    const int shifted = g_quirk_small_image_stripes && n >= 1 && n * g->stripe_sz - g->overlap < 0 && out_start < H;
    const int out_end = IMIN((n + 1) * g->stripe_sz, H);
    const int hrows = SH2 * 2 + 2;
    const int Dp = D + 2; /* padded path rows */
    const size_t rowN = (size_t)W1 * D;

    uint8_t *bytes = (uint8_t *)malloc((size_t)W * 8);
    cost_t *pix = (cost_t *)malloc(rowN * sizeof(cost_t));
    cost_t *hs = (cost_t *)malloc(rowN * hrows * sizeof(cost_t));
    cost_t *C = (cost_t *)calloc(rowN, sizeof(cost_t));
    cost_t *S = (cost_t *)malloc(rowN * sizeof(cost_t));            /* L_left, then L_left+L_top+L_right */
    cost_t *top = (cost_t *)malloc((size_t)W1 * Dp * sizeof(cost_t)); /* L_top of the previous row, padded */
    cost_t *topmin = (cost_t *)calloc(W1, sizeof(cost_t));
    cost_t *cur = (cost_t *)malloc(3 * (size_t)Dp * sizeof(cost_t));
    int16_t *disp2 = (int16_t *)malloc((size_t)W * sizeof(int16_t));
    int16_t *disp2cost = (int16_t *)malloc((size_t)W * sizeof(int16_t));
    if (!bytes || !pix || !hs || !C || !S || !top || !topmin || !cur || !disp2 || !disp2cost) return -3;
    rowbuf_t rb;
    rb.l[0] = bytes; rb.l[1] = bytes + W; rb.r[0] = bytes + 2 * W; rb.r[1] = bytes + 3 * W;
    rb.r0[0] = bytes + 4 * W; rb.r0[1] = bytes + 5 * W; rb.r1[0] = bytes + 6 * W; rb.r1[1] = bytes + 7 * W;

    for (int x = 0; x < W1; x++) { /* all path buffers start at zero at the stripe's first row */
        cost_t *t = top + (size_t)x * Dp;
        t[0] = t[Dp - 1] = SHRT_MAX;
        memset(t + 1, 0, D * sizeof(cost_t));
    }
    cost_t *lprev = cur, *lcur = cur + Dp, *tnew = cur + 2 * Dp;
    lprev[0] = lprev[Dp - 1] = lcur[0] = lcur[Dp - 1] = SHRT_MAX;

    for (int y = src_start; y < src_end; y++) {
        /* --- getRawMatchingCost: C(y) = sum of 2*SH2+1 hsum rows, first stripe row replicated upwards,
         *     rows >= H clamped to H-1 */
        if (y == src_start) {
            for (int k = src_start; k <= src_start + SH2; k++) {
                cost_t *hadd = hs + (size_t)(IMIN(k, H - 1) % hrows) * rowN;
                if (k < H) { pix_row(L, R, ldL, ldR, g, k, tab, &rb, pix); hsum_row(g, pix, hadd); }
                int scale = k == src_start ? SH2 + 1 : 1;
                for (size_t i = 0; i < rowN; i++) C[i] = (cost_t)(C[i] + hadd[i] * scale);
            }
        } else {
            int k = y + SH2;
            cost_t *hadd = hs + (size_t)(IMIN(k, H - 1) % hrows) * rowN;
            if (k < H) { pix_row(L, R, ldL, ldR, g, k, tab, &rb, pix); hsum_row(g, pix, hadd); }
            const cost_t *hsub = hs + (size_t)(IMAX(y - SH2 - 1, src_start) % hrows) * rowN;
            for (size_t i = 0; i < rowN; i++) C[i] = (cost_t)(C[i] + hadd[i] - hsub[i]);
        }

        for (int x = 0; x < W; x++) { disp2[x] = (int16_t)INVALID; disp2cost[x] = SHRT_MAX; }
        int16_t *drow = (y >= out_start) ? disp + (size_t)y * ldD : NULL;
        if (shifted) {   /* out[n * stripe_sz + j] = this run's row overlap + j */
            const int i = out_start + (y - g->overlap);
            drow = (y >= g->overlap && i < out_end) ? disp + (size_t)i * ldD : NULL;
        }

        /* --- forward pass: L_left (left -> right) and L_top (previous row -> this row, in place) */
        int lmin = 0;
        memset(lprev + 1, 0, D * sizeof(cost_t));
        for (int x = 0; x < W1; x++) {
            const cost_t *Cx = C + (size_t)x * D;
            lmin = sgm_step(Cx, lprev + 1, lcur + 1, D, P1, lmin + P2);
            memcpy(S + (size_t)x * D, lcur + 1, D * sizeof(cost_t));
            cost_t *sw = lprev; lprev = lcur; lcur = sw;
            cost_t *t = top + (size_t)x * Dp;
            int tm = sgm_step(Cx, t + 1, tnew + 1, D, P1, topmin[x] + P2);
            memcpy(t + 1, tnew + 1, D * sizeof(cost_t));
            topmin[x] = (cost_t)tm;
        }
        /* --- backward pass: L_right, S = L_left + L_top + L_right, WTA, uniqueness, disp2, sub-pixel */
        int rmin = 0;
        memset(lprev + 1, 0, D * sizeof(cost_t));
        for (int x = W1 - 1; x >= 0; x--) {
            const cost_t *Cx = C + (size_t)x * D;
            cost_t *Sx = S + (size_t)x * D;
            const cost_t *t = top + (size_t)x * Dp + 1;
            rmin = sgm_step(Cx, lprev + 1, lcur + 1, D, P1, rmin + P2);
            int best = 0, minS = SHRT_MAX;
            for (int d = 0; d < D; d++) {
                int s = sat16((int)Sx[d] + lcur[1 + d] + t[d]);
                Sx[d] = (cost_t)s;
                if (s < minS) { minS = s; best = d; }     /* first minimum wins */
            }
            cost_t *sw = lprev; lprev = lcur; lcur = sw;
            if (!drow) continue;                           /* warm-up row: result is discarded */

            if (g->uniq > 0) {
                int d;
                for (d = 0; d < D; d++)
                    if (Sx[d] * (100 - g->uniq) < minS * 100 && abs(d - best) > 1) break;
                if (d < D) continue;                       /* QUIRK: also skips the disp2 update */
            }
            int x2 = x + g->minX1 - best - g->minD;        /* matched right-image column */
            if (disp2cost[x2] > minS) { disp2cost[x2] = (int16_t)minS; disp2[x2] = (int16_t)(best + g->minD); }
            int dsp;
            if (0 < best && best < D - 1) {
                int den = IMAX(Sx[best - 1] + Sx[best + 1] - 2 * Sx[best], 1);
                dsp = best * DISP_SCALE + ((Sx[best - 1] - Sx[best + 1]) * DISP_SCALE + den) / (den * 2);
            } else
                dsp = best * DISP_SCALE;
            drow[x + g->minX1] = (int16_t)(dsp + g->minD * DISP_SCALE);
        }
        if (!drow) continue;
        /* --- pseudo left-right consistency check on this row.  QUIRK: disp2 was initialised with the SCALED invalid
         *     marker (minD-1)*16, which passes the ">= minD" test whenever minD >= 2 */
        for (int x = g->minX1; x < g->maxX1; x++) {
            int d1 = drow[x];
            if (d1 == INVALID) continue;
            int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
            int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < W && disp2[_x] >= g->minD && abs(disp2[_x] - _d) > g->d12 &&
                0 <= x_ && x_ < W && disp2[x_] >= g->minD && abs(disp2[x_] - d_) > g->d12)
                drow[x] = (int16_t)INVALID;
        }
    }
    free(bytes); free(pix); free(hs); free(C); free(S); free(top); free(topmin); free(cur); free(disp2); free(disp2cost);
    return 0;
}

/* ---- medianBlur(disp, 3) on int16, BORDER_REPLICATE -------------------------------------- */
static inline void srt(int *a, int *b) { if (*a > *b) { int t = *a; *a = *b; *b = t; } }
static void median3x3_s16(const int16_t *src, int16_t *dst, int W, int H, int ld) {
    for (int y = 0; y < H; y++) {
        const int16_t *r0 = src + (size_t)IMAX(y - 1, 0) * ld, *r1 = src + (size_t)y * ld, *r2 = src + (size_t)IMIN(y + 1, H - 1) * ld;
        for (int x = 0; x < W; x++) {
            int xl = IMAX(x - 1, 0), xr = IMIN(x + 1, W - 1);
            int p0 = r0[xl], p1 = r0[x], p2 = r0[xr], p3 = r1[xl], p4 = r1[x], p5 = r1[xr], p6 = r2[xl], p7 = r2[x], p8 = r2[xr];
            srt(&p1, &p2); srt(&p4, &p5); srt(&p7, &p8); srt(&p0, &p1); srt(&p3, &p4); srt(&p6, &p7);
            srt(&p1, &p2); srt(&p4, &p5); srt(&p7, &p8); srt(&p0, &p3); srt(&p5, &p8); srt(&p4, &p7);
            srt(&p3, &p6); srt(&p1, &p4); srt(&p2, &p5); srt(&p4, &p7); srt(&p4, &p2); srt(&p6, &p4);
            srt(&p4, &p2);
            dst[(size_t)y * ld + x] = (int16_t)p4;
        }
    }
}

/* ---- filterSpeckles(img, newVal, maxSpeckleSize, maxDiff): 4-connected flood fill --------- */
int sgbm_oracle_filter_speckles(int16_t *img, int W, int H, int ld, int newVal, int maxSpeckleSize, int maxDiff) {
    size_t np = (size_t)W * H;
    int *labels = (int *)calloc(np, sizeof(int));
    int *stack = (int *)malloc(np * sizeof(int));
    uint8_t *small = (uint8_t *)calloc(np + 1, 1);
    if (!labels || !stack || !small) return -3;
    int cur = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            int16_t *ds = img + (size_t)i * ld;
            int *ls = labels + (size_t)i * W;
            if (ds[j] == newVal) continue;
            if (ls[j]) { if (small[ls[j]]) ds[j] = (int16_t)newVal; continue; }
            int sp = 0, count = 0, p = i * W + j;
            ls[j] = ++cur;
            for (;;) {
                count++;
                int py = p / W, px = p % W;
                int dp = img[(size_t)py * ld + px];
#define TRY(cond, qy, qx)                                                                          \
    if (cond) { int q = (qy) * W + (qx); int dq = img[(size_t)(qy) * ld + (qx)];                   \
        if (!labels[q] && dq != newVal && abs(dp - dq) <= maxDiff) { labels[q] = cur; stack[sp++] = q; } }
                TRY(py < H - 1, py + 1, px) TRY(py > 0, py - 1, px) TRY(px < W - 1, py, px + 1) TRY(px > 0, py, px - 1)
#undef TRY
                if (sp == 0) break;
                p = stack[--sp];
            }
            if (count <= maxSpeckleSize) { small[cur] = 1; ds[j] = (int16_t)newVal; }
        }
    free(labels); free(stack); free(small);
    return 0;
}

/* ---- StereoSGBMImpl::compute -------------------------------------------------------------- */
/* disp: int16 H x W (x16 fixed point, invalid = (minD-1)*16).  raw_or_null receives the disparity
 * BEFORE medianBlur / filterSpeckles (stage boundary used by the kernel-level parity tests).
 * nthreads: stripes run in parallel (OpenCV: parallel_for_ over the 4 stripes). */
int sgbm_oracle_compute(const uint8_t *L, const uint8_t *R, int W, int H, int ldL, int ldR,
                        const sgbm_oracle_params *p, int16_t *disp, int16_t *raw_or_null, int nthreads) {
    geom_t g;
    int rc = derive(p, W, H, &g);
    if (rc == -2) {
        /* minX1 >= maxX1: the disparity range leaves no column to match; the original fills the map with the invalid
         * marker and returns [recalled: "if( minX1 >= maxX1 ) { disp1 = Scalar::all(INVALID_DISP_SCALED); return; }"] */
        const int16_t inv = (int16_t)((p->minDisparity - 1) * DISP_SCALE);
        for (size_t i = 0; i < (size_t)W * H; i++) disp[i] = inv;
        if (raw_or_null)
            for (size_t i = 0; i < (size_t)W * H; i++) raw_or_null[i] = inv;
        return 0;
    }
    if (rc) return rc;
    uint8_t tabmem[2304];
    for (int k = 0; k < 2304; k++) tabmem[k] = (uint8_t)(IMIN(IMAX(k - 1024, -g.ftzero), g.ftzero) + g.ftzero);
    const uint8_t *tab = tabmem + 1024;
    int16_t *tmp = (int16_t *)malloc((size_t)W * H * sizeof(int16_t));
    if (!tmp) return -3;
    const int INVALID = (g.minD - 1) * DISP_SCALE;
    for (size_t i = 0; i < (size_t)W * H; i++) tmp[i] = (int16_t)INVALID;
    int err = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int n = 0; n < NSTRIPES; n++) {
        int r = run_stripe(L, R, ldL, ldR, &g, n, tab, tmp, W);
        if (r) err = r;
    }
    if (err) { free(tmp); return err; }
    if (raw_or_null) memcpy(raw_or_null, tmp, (size_t)W * H * sizeof(int16_t));
    median3x3_s16(tmp, disp, W, H, W);
    free(tmp);
    if (p->speckleWindowSize > 0)
        return sgbm_oracle_filter_speckles(disp, W, H, W, INVALID, p->speckleWindowSize, DISP_SCALE * p->speckleRange);
    return 0;
}

/* rows of an H-row map whose content is UNDEFINED in the original (QUIRK_SMALL_IMAGE_STRIPES: assembled from stripe-buffer rows
 * that were never written); mask[i] = 1 for such rows, 0 otherwise.  Returns their number. */
int sgbm_oracle_undefined_rows(int H, const sgbm_oracle_params *p, uint8_t *mask) {
    const int stripe_sz = (H + NSTRIPES - 1) / NSTRIPES;
    const int overlap = (p->blockSize / 2 + 1) + (stripe_sz + 9) / 10;
    int cnt = 0;
    for (int i = 0; i < H; i++) {
        const int n = i / stripe_sz, j = i % stripe_sz;
        const int src_end = IMIN((n + 1) * stripe_sz, H);
        const int undef = g_quirk_small_image_stripes && n >= 1 && n * stripe_sz - overlap < 0 && overlap + j >= src_end;
        if (mask) mask[i] = (uint8_t)undef;
        cnt += undef;
    }
    return cnt;
}

/* Stage oracle for kernel-level tests: aggregated block cost C(y) of ONE stripe-free row band,
 * i.e. the exact 5x5 (bs x bs) box sum with row replication at `band_start` (what stripe n sees when
 * band_start = its src_start).  out: [(y1-y0)][W1][D]. */
int sgbm_oracle_cost_rows(const uint8_t *L, const uint8_t *R, int W, int H, int ldL, int ldR,
                          const sgbm_oracle_params *p, int band_start, int y0, int y1, int16_t *out) {
    geom_t g;
    int rc = derive(p, W, H, &g);
    if (rc) return rc;
    uint8_t tabmem[2304];
    for (int k = 0; k < 2304; k++) tabmem[k] = (uint8_t)(IMIN(IMAX(k - 1024, -g.ftzero), g.ftzero) + g.ftzero);
    const size_t rowN = (size_t)g.W1 * g.D;
    uint8_t *bytes = (uint8_t *)malloc((size_t)W * 8);
    cost_t *pix = (cost_t *)malloc(rowN * sizeof(cost_t)), *hs = (cost_t *)malloc(rowN * sizeof(cost_t));
    int *acc = (int *)malloc(rowN * sizeof(int));
    if (!bytes || !pix || !hs || !acc) return -3;
    rowbuf_t rb;
    rb.l[0] = bytes; rb.l[1] = bytes + W; rb.r[0] = bytes + 2 * W; rb.r[1] = bytes + 3 * W;
    rb.r0[0] = bytes + 4 * W; rb.r0[1] = bytes + 5 * W; rb.r1[0] = bytes + 6 * W; rb.r1[1] = bytes + 7 * W;
    for (int y = y0; y < y1; y++) {
        memset(acc, 0, rowN * sizeof(int));
        for (int j = -g.SH2; j <= g.SH2; j++) {
            int k = IMIN(IMAX(y + j, band_start), H - 1);
            pix_row(L, R, ldL, ldR, &g, k, tabmem + 1024, &rb, pix);
            hsum_row(&g, pix, hs);
            for (size_t i = 0; i < rowN; i++) acc[i] += hs[i];
        }
        for (size_t i = 0; i < rowN; i++) out[(size_t)(y - y0) * rowN + i] = (cost_t)acc[i];
    }
    free(bytes); free(pix); free(hs); free(acc);
    return 0;
}
