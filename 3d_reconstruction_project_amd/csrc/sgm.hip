// sgm.hip -- semi-global block matching (3-way) for gfx950, hand-written for 64-wide wavefronts.
//
// Replaces cv2.StereoSGBM(mode=MODE_SGBM_3WAY).compute (reference call sites: Calib_depth/depth2.py:146-158,251
// and the other depth*.py files); the arithmetic restated here is OpenCV's stereosgbm.cpp as pinned down in
// oracle/sgbm3way.c (every QUIRK listed there is reproduced bit for bit).
//
// Data layout in HBM (all int16 unless noted; "dp" = 128 for D<=128, 256 otherwise; w1 = maxX1-minX1):
//   rec_l/rec_r  uint2 [h][w]       per pixel: (g, g_lo, g_hi, i | i_lo, i_hi, 0, 0): prefiltered gradient, raw
//                                   intensity and their Birchfield-Tomasi half-pixel intervals
//   cost         [h][w1][dp]        aggregated block cost C (blockSize x blockSize box of the BT pixel cost)
//   cspec        [3][SH2][w1][dp]   C of the first SH2 rows of stripes 1..3 (box replicated at the stripe top)
//   hsum         [h][w1][dp]        L_left + L_right
//   raw / mins   [h][w]             WTA disparity (x16, before LR check) and its aggregated cost
// Lane mapping of every volume kernel: one wavefront owns ONE (row, column) disparity vector; lane l holds the
// 2*NP consecutive disparities d = 2*NP*l .. 2*NP*l+2*NP-1 as NP packed int16x2 registers, so a wave reads or
// writes 256*NP contiguous bytes per pixel and the d-1 / d+1 neighbours come from one DPP wave shift each way.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "r3d_internal.h"

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#ifndef R3D_EXP_COST
#define R3D_EXP_COST 0   /* timing experiments on k_cost2 (wrong results): 1 = no stores, 2 = no loads inside the loop */
#endif
#ifndef R3D_COST_STAGE
#define R3D_COST_STAGE 1   /* where k_cost2 stages the next input row: 1 right after the barrier (default: 0.67 -> 0.64 ms), 2 between the box sum and its stores (0.65) */
#endif
struct SgmGeom {
    int W, H, minD, D, NP, minX1, maxX1, W1, SW2, SH2, P1, P2, uniq, d12, ftzero, stripe_sz, overlap, invalid;
    int DP;  // disparity slots per cost-volume column: the smallest of 32 / 64 / 128 / 256 that holds D (v2 kernels; v1 and v3
             // only know 128 / 256 = NP * 128)
};

constexpr int PADPK = 0x7fff7fff;  // SHRT_MAX in both halves: the d=-1 / d=D padding of every path buffer

__device__ __forceinline__ s16x2 as_s(int v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u16x2 as_u(int v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ int as_i(s16x2 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int as_i(u16x2 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int pk_add(int a, int b) { return as_i(as_s(a) + as_s(b)); }
__device__ __forceinline__ int pk_sub(int a, int b) { return as_i(as_s(a) - as_s(b)); }
// Both halves known not to carry / borrow into each other (block costs: at most 121 * 189 = 22 869 per half): ONE 32-bit add, which
// gfx950 issues in ~2.5 cycles per wave against ~4.4 for v_pk_add_u16 (tools/micro/valu_rate.hip); same bits.
__device__ __forceinline__ int pk_add_nc(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int pk_sub_nb(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int pk_add_sat(int a, int b) { return as_i(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ int pk_min(int a, int b) { return as_i(__builtin_elementwise_min(as_s(a), as_s(b))); }
__device__ __forceinline__ int pk_umax(int a, int b) { return as_i(__builtin_elementwise_max(as_u(a), as_u(b))); }
__device__ __forceinline__ int pk_umin(int a, int b) { return as_i(__builtin_elementwise_min(as_u(a), as_u(b))); }
__device__ __forceinline__ int pk_usub_sat(int a, int b) { return as_i(__builtin_elementwise_sub_sat(as_u(a), as_u(b))); }
__device__ __forceinline__ int pk_dup(int v) { return (v & 0xffff) | (v << 16); }
__device__ __forceinline__ int lo16(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(int v) { return v >> 16; }

// lane i receives lane i-1 (lane 0 keeps `fill`) / lane i+1 (lane 63 keeps `fill`)
__device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

// butterfly all-reduce over the 64 lanes: every lane ends with the result (xor 1, 2 via quad_perm; 4, 8 via
// row_half_mirror / row_mirror; 16, 32 via the gfx950 permlane swaps)
#define R3D_BUTTERFLY(OP)                                                                   \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));       \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));       \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));      \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));      \
    { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = OP((T)r[0], (T)r[1]); } \
    { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = OP((T)r[0], (T)r[1]); }
__device__ __forceinline__ int wave_allmin_i32(int v) { typedef int T; R3D_BUTTERFLY(min) return v; }
__device__ __forceinline__ unsigned wave_allmin_u32(unsigned v) { typedef unsigned T; R3D_BUTTERFLY(min) return v; }
__device__ __forceinline__ int wave_allmax_i32(int v) { typedef int T; R3D_BUTTERFLY(max) return v; }

// ---------------------------------------------------------------------------------------------------------
// One SGM path step for one disparity vector (OpenCV accumulateCostsLeftTop / accumulateCostsRight):
//   L[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, minp+P2) - (minp+P2),   Lp[-1] = Lp[D] = SHRT_MAX
// P holds Lp on entry and L on exit; minp (wave-uniform) holds min_d Lp on entry and min_d L on exit.
// The packed adds cannot overflow inside the envelope checked on the host (C <= 16383, P2 <= 16383); the
// +P1 on the SHRT_MAX padding saturates (v_pk_add_i16 clamp), which never wins the min.
template <int NP>
__device__ __forceinline__ void sgm_step(int (&P)[NP], int &minp, const int (&C)[NP], int P1pk, int P2, bool lane_valid) {
    const int mp2 = pk_dup(minp + P2);
    const int up = wave_shr1(P[NP - 1], PADPK);  // lane-1's highest pair
    const int dn = wave_shl1(P[0], PADPK);       // lane+1's lowest pair
    int Q[NP];
    int m32 = 0x7fff;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int below = j == 0 ? up : P[j - 1];
        const int above = j == NP - 1 ? dn : P[j + 1];
        const int A = __builtin_amdgcn_alignbit(P[j], below, 16);  // (L[d-1]) for both halves
        const int B = __builtin_amdgcn_alignbit(above, P[j], 16);  // (L[d+1]) for both halves
        const int nb = pk_add_sat(pk_min(A, B), P1pk);
        const int m = pk_min(pk_min(P[j], mp2), nb);
        int q = pk_add(C[j], pk_sub(m, mp2));
        q = lane_valid ? q : PADPK;
        Q[j] = q;
        m32 = min(m32, min(lo16(q), hi16(q)));
    }
#pragma unroll
    for (int j = 0; j < NP; j++) P[j] = Q[j];
    minp = wave_allmin_i32(m32);
}

// ---------------------------------------------------------------------------------------------------------
// Generic lane mapping (v2 kernels): a disparity vector of DP = 2*NPL*LPC entries is spread over LPC adjacent lanes,
// NPL packed registers (2*NPL consecutive disparities) per lane, so one wave holds 64/LPC independent vectors.
// Cross-lane work (2 DPP shifts + log2(LPC) butterfly stages) is amortised over NPL registers.
template <int LPC>
__device__ __forceinline__ int grp_shr1(int v, int fill, bool first) {  // lane i <- lane i-1 inside its LPC-lane group
    if constexpr (LPC == 1) return fill;
    else if constexpr (LPC == 64) return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
    else if constexpr (LPC == 32) { int t = __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); return first ? fill : t; }
    else { int t = __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); return (LPC < 16 && first) ? fill : t; }
}
template <int LPC>
__device__ __forceinline__ int grp_shl1(int v, int fill, bool last) {   // lane i <- lane i+1 inside its group
    if constexpr (LPC == 1) return fill;
    else if constexpr (LPC == 64) return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false);
    else if constexpr (LPC == 32) { int t = __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); return last ? fill : t; }
    else { int t = __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xf, 0xf, false); return (LPC < 16 && last) ? fill : t; }
}
// all-reduce min inside each LPC-lane group; `old` = identity so that the DPP move folds into v_min_i32_dpp
template <int LPC>
__device__ __forceinline__ int grp_allmin(int v) {
    if constexpr (LPC >= 2) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0xB1, 0xf, 0xf, false));
    if constexpr (LPC >= 4) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x4E, 0xf, 0xf, false));
    if constexpr (LPC >= 8) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x141, 0xf, 0xf, false));
    if constexpr (LPC >= 16) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x140, 0xf, 0xf, false));
    if constexpr (LPC >= 32) { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = min((int)r[0], (int)r[1]); }
    if constexpr (LPC >= 64) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = min((int)r[0], (int)r[1]); }
    return v;
}
template <int LPC>
__device__ __forceinline__ int grp_allsum(int v) {
    if constexpr (LPC >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    if constexpr (LPC >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    if constexpr (LPC >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    if constexpr (LPC >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    if constexpr (LPC >= 32) { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = (int)r[0] + (int)r[1]; }
    if constexpr (LPC >= 64) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = (int)r[0] + (int)r[1]; }
    return v;
}

// same recurrence as sgm_step, generic mapping; minp is uniform inside a group.  Critical path per step:
// add P2 -> pk_min -> pk_min -> pk_add -> (pk_min tree) -> sdwa min -> log2(LPC) butterfly stages.
template <int NPL, int LPC, bool PADDED = true>
__device__ __forceinline__ void sgm_step_g(int (&P)[NPL], int &minp, const int (&C)[NPL], int P1pk, int P2, bool first, bool last,
                                           bool lane_valid) {
    const short m16 = (short)(minp + P2);
    const int mp2 = as_i((s16x2){m16, m16});      // splat: folded into op_sel by the compiler
    const int up = grp_shr1<LPC>(P[NPL - 1], PADPK, first);
    const int dn = grp_shl1<LPC>(P[0], PADPK, last);
    // Stage by stage across all NPL registers ("breadth first"): on gfx950 a packed (VOP3P) result needs one wait state
    // before a dependent VALU read, and the compiler fills it with s_nop instead of independent work when the source is
    // written register by register; in this order every dependent pair is NPL instructions apart.
    int Q[NPL], t1[NPL], t3[NPL], cm[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int below = j == 0 ? up : P[j - 1];
        const int above = j == NPL - 1 ? dn : P[j + 1];
        const int A = __builtin_amdgcn_alignbit(P[j], below, 16);
        const int B = __builtin_amdgcn_alignbit(above, P[j], 16);
        t1[j] = pk_min(A, B);
    }
#pragma unroll
    for (int j = 0; j < NPL; j++) cm[j] = pk_sub(C[j], mp2);  // independent of the min chain
#pragma unroll
    for (int j = 0; j < NPL; j++) t3[j] = pk_min(P[j], mp2);
#pragma unroll
    for (int j = 0; j < NPL; j++) t1[j] = pk_add_sat(t1[j], P1pk);
#pragma unroll
    for (int j = 0; j < NPL; j++) t3[j] = pk_min(t3[j], t1[j]);
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int q = pk_add(cm[j], t3[j]);
        Q[j] = (!PADDED || lane_valid) ? q : PADPK;
    }
    // balanced min tree (a linear chain would pay the wait state at every step)
    int mt[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) mt[j] = Q[j];
#pragma unroll
    for (int w = NPL / 2; w >= 1; w /= 2) {
#pragma unroll
        for (int j = 0; j < w; j++) mt[j] = pk_min(mt[j], mt[j + w]);
    }
    const int m = mt[0];
#pragma unroll
    for (int j = 0; j < NPL; j++) P[j] = Q[j];
    minp = grp_allmin<LPC>(min(lo16(m), hi16(m)));
}

// Two independent chains (A, B) of the wave-wide mapping (NPL = 1, LPC = 64) advanced in lockstep: their per-lane minima
// travel through ONE packed butterfly (lo16 = A, hi16 = B; v_pk_min_i16 cannot take a DPP modifier, so each of the
// four row stages is mov_dpp + pk_min), which both shortens the instruction stream and hard-wires the interleave that
// the scheduler does not find by itself.  minAB holds (min A | min B << 16).
__device__ __forceinline__ int pk_allmin64(int v) {
    // every lane has a source in these four permutations, so the DPP move needs no `old` value (no extra copy)
    v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
    v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
    v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
    v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));
    { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = pk_min((int)r[0], (int)r[1]); }
    { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = pk_min((int)r[0], (int)r[1]); }
    return v;
}
template <bool PADDED>
__device__ __forceinline__ void sgm_step_dual(int &PA, int &PB, int &minAB, int CA, int CB, int P1pk, int P2pk, bool lane_valid) {
    const s16x2 m = as_s(pk_add(minAB, P2pk));
    const int mp2A = as_i((s16x2){m.x, m.x}), mp2B = as_i((s16x2){m.y, m.y});
    const int upA = wave_shr1(PA, PADPK), upB = wave_shr1(PB, PADPK);
    const int dnA = wave_shl1(PA, PADPK), dnB = wave_shl1(PB, PADPK);
    const int nbA = pk_add_sat(pk_min(__builtin_amdgcn_alignbit(PA, upA, 16), __builtin_amdgcn_alignbit(dnA, PA, 16)), P1pk);
    const int nbB = pk_add_sat(pk_min(__builtin_amdgcn_alignbit(PB, upB, 16), __builtin_amdgcn_alignbit(dnB, PB, 16)), P1pk);
    const int cmA = pk_sub(CA, mp2A), cmB = pk_sub(CB, mp2B);
    int qA = pk_add(cmA, pk_min(pk_min(PA, mp2A), nbA));
    int qB = pk_add(cmB, pk_min(pk_min(PB, mp2B), nbB));
    if (PADDED) { qA = lane_valid ? qA : PADPK; qB = lane_valid ? qB : PADPK; }
    PA = qA; PB = qB;
    // (lo A | lo B << 16) vs (hi A | hi B << 16)
    const int lo = __builtin_amdgcn_perm(qB, qA, 0x05040100), hi = __builtin_amdgcn_perm(qB, qA, 0x07060302);
    minAB = pk_allmin64(pk_min(lo, hi));
}

// packed (two int16 lanes) all-reduce min inside each LPC-lane group
template <int LPC>
__device__ __forceinline__ int pk_grp_allmin(int v) {
    if constexpr (LPC >= 2) v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
    if constexpr (LPC >= 4) v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
    if constexpr (LPC >= 8) v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
    if constexpr (LPC >= 16) v = pk_min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));
    if constexpr (LPC >= 32) { auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); v = pk_min((int)r[0], (int)r[1]); }
    if constexpr (LPC >= 64) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); v = pk_min((int)r[0], (int)r[1]); }
    return v;
}
// two independent chains (A, B) of the generic mapping advanced in lockstep, sharing one packed butterfly
template <int NPL, int LPC, bool PADDED>
__device__ __forceinline__ void sgm_step_dual_g(int (&PA)[NPL], int (&PB)[NPL], int &minAB, const int (&CA)[NPL], const int (&CB)[NPL],
                                                int P1pk, int P2pk, bool first, bool last, bool lane_valid) {
    const s16x2 m = as_s(pk_add(minAB, P2pk));
    const int mp2A = as_i((s16x2){m.x, m.x}), mp2B = as_i((s16x2){m.y, m.y});
    const int upA = grp_shr1<LPC>(PA[NPL - 1], PADPK, first), upB = grp_shr1<LPC>(PB[NPL - 1], PADPK, first);
    const int dnA = grp_shl1<LPC>(PA[0], PADPK, last), dnB = grp_shl1<LPC>(PB[0], PADPK, last);
    int QA[NPL], QB[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int belowA = j == 0 ? upA : PA[j - 1], aboveA = j == NPL - 1 ? dnA : PA[j + 1];
        const int belowB = j == 0 ? upB : PB[j - 1], aboveB = j == NPL - 1 ? dnB : PB[j + 1];
        const int nbA = pk_add_sat(pk_min(__builtin_amdgcn_alignbit(PA[j], belowA, 16), __builtin_amdgcn_alignbit(aboveA, PA[j], 16)), P1pk);
        const int nbB = pk_add_sat(pk_min(__builtin_amdgcn_alignbit(PB[j], belowB, 16), __builtin_amdgcn_alignbit(aboveB, PB[j], 16)), P1pk);
        int qA = pk_add(pk_sub(CA[j], mp2A), pk_min(pk_min(PA[j], mp2A), nbA));
        int qB = pk_add(pk_sub(CB[j], mp2B), pk_min(pk_min(PB[j], mp2B), nbB));
        if (PADDED) { qA = lane_valid ? qA : PADPK; qB = lane_valid ? qB : PADPK; }
        QA[j] = qA; QB[j] = qB;
    }
    int mA = QA[0], mB = QB[0];
#pragma unroll
    for (int j = 1; j < NPL; j++) { mA = pk_min(mA, QA[j]); mB = pk_min(mB, QB[j]); }
#pragma unroll
    for (int j = 0; j < NPL; j++) { PA[j] = QA[j]; PB[j] = QB[j]; }
    const int lo = __builtin_amdgcn_perm(mB, mA, 0x05040100), hi = __builtin_amdgcn_perm(mB, mA, 0x07060302);
    minAB = pk_grp_allmin<LPC>(pk_min(lo, hi));
}

// trunc(n / d) for d > 0, |n|, d < 2^23: float reciprocal estimate + exact integer correction
__device__ __forceinline__ int trunc_div_small(int n, int d) {
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    int r = n - q * d;
    if (n >= 0) { if (r < 0) q--; else if (r >= d) q++; }
    else { if (r > 0) q++; else if (r <= -d) q--; }
    return q;
}
// smallest integer T with T * a >= thr  (a > 0): S * a < thr  <=>  S < T
__device__ __forceinline__ int ceil_div_small(int thr, int a, float inv_a) {
    // |thr| < 2^23 and a <= 100: the float estimate is within one of the answer; two branch-free corrections each way (a
    // `while` here becomes a divergent loop with exec-mask branches in the middle of the row step)
    int q = (int)floorf((float)thr * inv_a);
    q += (q * a < thr) ? 1 : 0;
    q += (q * a < thr) ? 1 : 0;
    q -= ((q - 1) * a >= thr) ? 1 : 0;
    q -= ((q - 1) * a >= thr) ? 1 : 0;
    return q;
}

// ---------------------------------------------------------------------------------------------------------
// k_prefilter: per pixel, both images: clipped x-Sobel (calcPixelCostBT's `tab` lookup), raw intensity, and the
// half-pixel min/max intervals of both.  QUIRK: columns 0 and w-1 of both channels read ftzero.
// gradient(x, y) = clamp(2*(I[y][x+1]-I[y][x-1]) + (I[y-1][x+1]-I[y-1][x-1]) + (I[y+1][x+1]-I[y+1][x-1]), -ft, ft) + ft with rows
// clamped to the image; intensity = I[y][x].
__device__ __forceinline__ void bt_interval(int vm, int v, int vp, bool has_m, bool has_p, int &lo, int &hi) {
    int l = has_m ? (v + vm) / 2 : v, r = has_p ? (v + vp) / 2 : v;
    lo = min(min(l, r), v);
    hi = max(max(l, r), v);
}

__global__ void __launch_bounds__(256) k_fill_s16(int16_t *__restrict__ p, size_t n, int16_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

// One workgroup per 256-pixel column segment and band of PF_ROWS rows: every source row is read once (260 columns incl.
// halo; a thread keeps the three rows of its column in registers and rolls them), the gradient / intensity pair of 258
// positions is computed from LDS, and each thread forms its pixel's two intervals from its neighbours' pairs.  All global
// loads use clamped coordinates (no branch around a load).
#define PF_ROWS 8
__global__ void __launch_bounds__(256) k_prefilter(const uint8_t *__restrict__ L, const uint8_t *__restrict__ R, int ld,
                                                   int W, int H, int ft, uint2 *__restrict__ recL, uint2 *__restrict__ recR) {
    __shared__ int raw[260];  // rows y-1 | y << 8 | y+1 << 16 at columns x0-2 .. x0+257
    __shared__ int gi[258];   // gradient | intensity << 8 at columns x0-1 .. x0+256
    const int t = threadIdx.x, x0 = blockIdx.x * 256, yb = blockIdx.y * PF_ROWS;
    const uint8_t *img = blockIdx.z == 0 ? L : R;
    uint2 *rec = blockIdx.z == 0 ? recL : recR;
    const int xc = min(max(x0 - 2 + t, 0), W - 1), xe = min(max(x0 + 254 + (t & 3), 0), W - 1);
    auto px = [&](int yy, int xx) { return (int)img[(size_t)min(max(yy, 0), H - 1) * ld + xx]; };
    int a = px(yb - 1, xc), r = px(yb, xc), ae = px(yb - 1, xe), re = px(yb, xe);
    int bn = px(yb + 1, xc), ben = px(yb + 1, xe);   // the row below, requested ONE iteration early (its round trip ran on the critical path of every row)
    const int x = x0 + t;
    const bool hm = x > 0, hp = x < W - 1;
    for (int y = yb; y < min(yb + PF_ROWS, H); y++) {
        const int b = bn, be = ben;
        bn = px(y + 2, xc); ben = px(y + 2, xe);
        raw[t] = a | (r << 8) | (b << 16);
        if (t < 4) raw[256 + t] = ae | (re << 8) | (be << 16);
        a = r; r = b; ae = re; re = be;
        __syncthreads();
        auto pair_at = [&](int j) {  // j: index into gi, column x0-1+j, raw index of that column = j+1
            const int xx = x0 - 1 + j;
            const int m = raw[j], c = raw[j + 1], p = raw[j + 2];
            const int s = (((p >> 8) & 255) - ((m >> 8) & 255)) * 2 + ((p & 255) - (m & 255)) + (((p >> 16) & 255) - ((m >> 16) & 255));
            int g = min(max(s, -ft), ft) + ft, i = (c >> 8) & 255;
            if (xx <= 0 || xx >= W - 1) { g = ft; i = ft; }
            gi[j] = g | (i << 8);
        };
        pair_at(t);
        if (t < 2) pair_at(256 + t);
        __syncthreads();
        if (x < W) {
            const int vm = gi[t], v = gi[t + 1], vp = gi[t + 2];
            int g0, g1, i0, i1;
            const int g = v & 255, i = v >> 8;
            bt_interval(vm & 255, g, vp & 255, hm, hp, g0, g1);
            bt_interval(vm >> 8, i, vp >> 8, hm, hp, i0, i1);
            rec[(size_t)y * W + x] = make_uint2((unsigned)g | (g0 << 8) | (g1 << 16) | (i << 24), (unsigned)i0 | (i1 << 8));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_cost: Birchfield-Tomasi pixel cost -> blockSize x blockSize box sum -> cost volume.
// Workgroup = 8 waves, one tile of TX cost columns, marching down a band of rows 8 at a time:
//   phase 1: wave w computes the horizontal box sums hsum(x, r) of image row r for the tile (sliding along x;
//            the last 2*SW2+1 pixel-cost vectors live in a per-wave LDS ring) into an LDS ring of rows;
//   phase 2: wave w sums the 2*SH2+1 ring rows around output row y and streams C(y, tile) to HBM
//            (TX*256*NP contiguous bytes per wave).
// Replicated borders: columns clamp to [0, w1) in COST coordinates, rows clamp to [clampTop, h-1].
__device__ __forceinline__ int bt_cost_pk(int U, int U0, int U1, int V, int V0, int V1) {
    int c0 = pk_umax(pk_usub_sat(U, V1), pk_usub_sat(V0, U));
    int c1 = pk_umax(pk_usub_sat(V, U1), pk_usub_sat(U0, V));
    return pk_umin(c0, c1);
}

constexpr int COST_NW = 8;

template <int NP>
__global__ void __launch_bounds__(COST_NW * 64) k_cost(const uint2 *__restrict__ recL, const uint2 *__restrict__ recR, SgmGeom g,
                                                       int *__restrict__ cvol, int *__restrict__ cspec, int TX, int BAND,
                                                       int nMain, int RING) {
    extern __shared__ int lds[];
    constexpr int NPW = NP * 64;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NR = COST_NW + 2 * g.SH2;
    int *hs = lds;                                          // [NR][TX][NPW]
    int *ring = lds + (size_t)NR * TX * NPW + w * RING * NPW;  // per wave [RING][NPW]
    const int xt0 = blockIdx.x * TX;
    const size_t rowWords = (size_t)g.W1 * NPW;
    int y0, y1, clampTop;
    int *obase;  // output row y lives at obase + (y - y0) * rowWords
    if ((int)blockIdx.y < nMain) {
        y0 = blockIdx.y * BAND;
        y1 = min(y0 + BAND, g.H);
        clampTop = 0;
        obase = cvol + (size_t)y0 * rowWords;
    } else {
        const int n = blockIdx.y - nMain + 1;
        const int ss = max(min(n * g.stripe_sz - g.overlap, g.H), 0);
        y0 = ss;
        y1 = min(ss + g.SH2, g.H);
        clampTop = ss;
        obase = cspec + (size_t)(n - 1) * g.SH2 * rowWords;
    }
    const int ntx = min(TX, g.W1 - xt0);
    int have = max(y0 - g.SH2, clampTop) - 1;
    for (int yb = y0; yb < y1; yb += COST_NW) {
        const int need = min(yb + COST_NW - 1 + g.SH2, g.H - 1);
        // ---- phase 1
        for (int r = have + 1 + w; r <= need; r += COST_NW) {
            const uint2 *lrow = recL + (size_t)r * g.W;
            const uint2 *rrow = recR + (size_t)r * g.W;
            int *hrow = hs + (size_t)(r % NR) * TX * NPW;
            int acc[NP];
#pragma unroll
            for (int j = 0; j < NP; j++) acc[j] = 0;
            const int kstart = xt0 - g.SW2, kend = xt0 + ntx - 1 + g.SW2;
            for (int k = kstart; k <= kend; ++k) {
                const int xc = min(max(k, 0), g.W1 - 1);
                const int x = xc + g.minX1;
                const uint2 lr = lrow[x];  // wave-uniform
                const int Ug = (lr.x & 255) * 0x10001, Ug0 = ((lr.x >> 8) & 255) * 0x10001, Ug1 = ((lr.x >> 16) & 255) * 0x10001;
                const int Ui = (lr.x >> 24) * 0x10001, Ui0 = (lr.y & 255) * 0x10001, Ui1 = ((lr.y >> 8) & 255) * 0x10001;
                const int rbase = x - g.minD - 2 * NP * lane;  // right column of this lane's lowest disparity
                const int slot = (k - kstart) & (RING - 1);
                const int oslot = (k - 2 * g.SW2 - kstart) & (RING - 1);
                const bool emit = k >= xt0 + g.SW2;
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    const int ra = min(max(rbase - 2 * j, 0), g.W - 1);      // d even  (low half)
                    const int rb = min(max(rbase - 2 * j - 1, 0), g.W - 1);  // d odd   (high half)
                    const uint2 A = rrow[ra], B = rrow[rb];
                    const int Vg = __builtin_amdgcn_perm(B.x, A.x, 0x0c040c00), Vg0 = __builtin_amdgcn_perm(B.x, A.x, 0x0c050c01);
                    const int Vg1 = __builtin_amdgcn_perm(B.x, A.x, 0x0c060c02), Vi = __builtin_amdgcn_perm(B.x, A.x, 0x0c070c03);
                    const int Vi0 = __builtin_amdgcn_perm(B.y, A.y, 0x0c040c00), Vi1 = __builtin_amdgcn_perm(B.y, A.y, 0x0c050c01);
                    const int cg = bt_cost_pk(Ug, Ug0, Ug1, Vg, Vg0, Vg1);
                    const int ci = bt_cost_pk(Ui, Ui0, Ui1, Vi, Vi0, Vi1);
                    const int pix = pk_add(cg, (ci >> 2) & 0x3fff3fff);  // gradient + (intensity >> 2), both halves
                    acc[j] = pk_add(acc[j], pix);
                    ring[slot * NPW + lane * NP + j] = pix;
                    if (emit) {
                        hrow[(k - g.SW2 - xt0) * NPW + lane * NP + j] = acc[j];
                        acc[j] = pk_sub(acc[j], ring[oslot * NPW + lane * NP + j]);
                    }
                }
            }
        }
        have = need;
        __syncthreads();
        // ---- phase 2
        const int yo = yb + w;
        if (yo < y1) {
            int *orow = obase + (size_t)(yo - y0) * rowWords + (size_t)xt0 * NPW;
            for (int xt = 0; xt < ntx; ++xt) {
                int s[NP];
#pragma unroll
                for (int j = 0; j < NP; j++) s[j] = 0;
                for (int dy = -g.SH2; dy <= g.SH2; ++dy) {
                    const int rr = min(max(yo + dy, clampTop), g.H - 1);
                    const int *h = hs + ((size_t)(rr % NR) * TX + xt) * NPW + lane * NP;
#pragma unroll
                    for (int j = 0; j < NP; j++) s[j] = pk_add(s[j], h[j]);
                }
#pragma unroll
                for (int j = 0; j < NP; j++) orow[xt * NPW + lane * NP + j] = s[j];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_cost2: same result as k_cost, restructured for the vector pipes (v1 was bound by the scalar ALU: ~35 SALU per
// column step, and every load was waited on immediately).
// Mapping: lane = (column g = lane / LPC, disparity chunk k = lane % LPC), 16 disparities (8 packed registers) per lane,
// CW = 64/LPC adjacent columns per wave, NWAVE waves per workgroup = one tile of TC = NWAVE*CW columns (2*SH2 of them
// halo) that marches DOWN a band of rows:
//   per row: (a) all threads stage the NEXT row's inputs into LDS: left records, and for the right image ready-made
//                packed pair words W_q[r] = q[r] | q[r-1] << 16 for the six BT quantities (no byte shuffles later);
//            (b) every lane evaluates the Birchfield-Tomasi cost of its 16 disparities (3 ds_read_b64 + 17 packed
//                VALU per register) and slides the VERTICAL box sum, whose 2*SH2+1 rows live in registers;
//            (c) vertical sums go to an LDS tile; one barrier; (d) each lane adds the 2*SH2+1 neighbouring columns
//                (ds_read_b128) and streams C to HBM, 2 KB contiguous per wave.
// Borders: columns clamp in cost coordinates, rows clamp to [clampTop, h-1] (replication, as the original).

// VCH (v3): the vertical path L_top is aggregated right here, on the freshly summed block cost that is still in
// registers: one workgroup column-tile per STRIPE marches from the stripe's first warm-up row to its last row, and
// writes C and L_top for the rows the stripe owns (the warm-up rows' special block costs never leave the chip).
template <int LPC, int SH2, bool TRACK, bool VCH, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64) k_cost2(const uint2 *__restrict__ recL, const uint2 *__restrict__ recR, SgmGeom g,
                                                            int *__restrict__ cvol, int *__restrict__ cspec, int BAND, int nMain,
                                                            int *__restrict__ maxc, int *__restrict__ ltvol, int tile0) {
    constexpr int NPL = 8, CW = 64 / LPC, TC = NWAVE * CW, TO = TC - 2 * SH2, DP = 16 * LPC, DPW = NPL * LPC;
    constexpr int R = 2 * SH2 + 1, NRR = TC + DP, NT = NWAVE * 64;
    // pair words: 6 dwords per right pixel, plus 8 dwords of padding after every 16 pixels: lanes of one column group
    // read records 16 apart (16*6 dwords = 32 mod 64 banks -> 4-way conflicts); with the pad the eight chunks land on
    // eight different multiples of 8 banks and the four column groups of a half-wave fill the gaps: conflict-free.
    constexpr int SWN = NRR * 6 + (NRR / 16 + 1) * 8;
    __shared__ int sW[2][SWN];            // right-image pair words of one row, double buffered
    __shared__ uint2 sL[2][TC];           // left records of one row
    __shared__ int sV[2][TC * DPW];       // vertical box sums of the tile, double buffered
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, k = lane % LPC, grp = lane / LPC;
    const int cl = w * CW + grp;                       // local column 0..TC-1
    const int t0 = (blockIdx.x + tile0) * TO;          // first OUTPUT cost column of the tile (tile0: column-slab launches)
    const int xc = min(max(t0 - SH2 + cl, 0), g.W1 - 1);   // cost column this lane evaluates (replicated at the borders)
    const int x = xc + g.minX1;                        // image column
    const int r_base = max(t0 - SH2, 0) + g.minX1 - g.minD - (DP - 1);
    const int ri0 = min(max(x - g.minD - 16 * k - r_base, 15), NRR - 1);   // pair-word index of this lane's lowest disparity
    const size_t rowWords = (size_t)g.W1 * DPW;
    int y0, y1, clampTop, out_start = 0;
    int *obase;
    if (VCH) {
        const int n = blockIdx.y;                       // stripe
        y0 = max(min(n * g.stripe_sz - g.overlap, g.H), 0);
        y1 = min((n + 1) * g.stripe_sz, g.H);
        clampTop = y0;
        out_start = min(n * g.stripe_sz, g.H);
        obase = cvol + (size_t)y0 * rowWords;
    } else if ((int)blockIdx.y < nMain) {
        y0 = blockIdx.y * BAND; y1 = min(y0 + BAND, g.H); clampTop = 0;
        obase = cvol + (size_t)y0 * rowWords;
    } else {
        const int n = blockIdx.y - nMain + 1;
        const int ss = max(min(n * g.stripe_sz - g.overlap, g.H), 0);
        y0 = ss; y1 = min(ss + SH2, g.H); clampTop = ss;
        obase = cspec + (size_t)(n - 1) * SH2 * rowWords;
    }
    if (y0 >= y1) return;
    auto crow = [&](int yy) { return min(max(yy, clampTop), g.H - 1); };
    // vertical path state (VCH)
    int LT[NPL], ltmin = 0;
    const bool lane_valid = 16 * k < g.D;
#pragma unroll
    for (int j = 0; j < NPL; j++) LT[j] = lane_valid ? 0 : PADPK;
    const int P1pk = pk_dup(g.P1);

    // (a) staging of an image row: the raw records are fetched one iteration EARLY into registers (fetch), and turned
    // into LDS pair words one iteration later (commit), so the global-load latency is never waited for inside a row
    static_assert(NRR <= NT, "one pair-word record per thread");
    uint2 pfL = make_uint2(0, 0), pfA = make_uint2(0, 0), pfB = make_uint2(0, 0);
    auto fetch = [&](int row) {
        const uint2 *lr = recL + (size_t)row * g.W, *rr = recR + (size_t)row * g.W;
        if (tid < TC) pfL = lr[min(max(t0 - SH2 + tid, 0), g.W1 - 1) + g.minX1];
        if (tid < NRR) {
            const int r = r_base + tid;
            pfA = rr[min(max(r, 0), g.W - 1)];
            pfB = rr[min(max(r - 1, 0), g.W - 1)];
        }
    };
    auto commit = [&](int b) {
        if (tid < TC) sL[b][tid] = pfL;
        if (tid < NRR) {
            const uint2 A = pfA, B = pfB;
            int *o = &sW[b][tid * 6 + (tid >> 4) * 8];
            o[0] = __builtin_amdgcn_perm(B.x, A.x, 0x0c040c00); o[1] = __builtin_amdgcn_perm(B.x, A.x, 0x0c050c01);
            o[2] = __builtin_amdgcn_perm(B.x, A.x, 0x0c060c02); o[3] = __builtin_amdgcn_perm(B.x, A.x, 0x0c070c03);
            o[4] = __builtin_amdgcn_perm(B.y, A.y, 0x0c040c00); o[5] = __builtin_amdgcn_perm(B.y, A.y, 0x0c050c01);
        }
    };
    // (b) BT pixel cost of this lane's 16 disparities from buffer b
    auto pixel_cost = [&](int b, int (&pix)[NPL]) {
        const uint2 lr = sL[b][cl];
        const int Ug = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c000c00), Ug0 = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c010c01);
        const int Ug1 = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c020c02), Ui = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c030c03);
        const int Ui0 = __builtin_amdgcn_perm(lr.y, lr.y, 0x0c000c00), Ui1 = __builtin_amdgcn_perm(lr.y, lr.y, 0x0c010c01);
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int rj = ri0 - 2 * j;
            // three ds_read_b64 (banks mod 64, 2 LDS cycles each, conflict-free with the padding above).  Volatile keeps the compiler
            // from pairing two of them into ds_read2_b64, which is banked mod 32 in 16-lane groups: chunks k and k+4 then collide
            // (2-way) and the pair costs 16 LDS cycles instead of 4 -- the 31 % bank conflicts of the round-2/3 counters.
            typedef int v2i __attribute__((ext_vector_type(2)));
            typedef const volatile __attribute__((address_space(3))) v2i lds_v2i;   // (a plain volatile pointer would read through flat_load)
            lds_v2i *p = (lds_v2i *)&sW[b][rj * 6 + (rj >> 4) * 8];
            const v2i a = p[0], bq = p[1], c = p[2];       // (Vg, Vg0) (Vg1, Vi) (Vi0, Vi1)
            const int cg = bt_cost_pk(Ug, Ug0, Ug1, a.x, a.y, bq.x);
            const int ci = bt_cost_pk(Ui, Ui0, Ui1, bq.y, c.x, c.y);
            pix[j] = pk_add_nc(cg, (ci >> 2) & 0x3fff3fff);
        }
    };

    // The vertical window (R rows of pixel costs, each <= 189 so stored as bytes: 4 registers per row) lives in
    // registers and is rotated by plain moves, which keeps the loop rolled and the register count predictable.
    int ring[R][NPL / 2], vs[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) vs[j] = 0;
#pragma unroll
    for (int q = 0; q < R; q++)
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) ring[q][j] = 0;
    const bool is_out = cl >= SH2 && cl < TC - SH2 && (t0 - SH2 + cl) < g.W1;
    int *optr = obase + (size_t)(t0 - SH2 + cl) * DPW + k * NPL;
    const int tile_x0 = t0 - SH2;
    int cmax = 0;   // TRACK: running maximum of every emitted block cost (exact-arithmetic envelope check on the host)
    fetch(crow(y0 - SH2));
    commit(0);
    fetch(crow(y0 - SH2 + 1));
    commit(1);
    fetch(crow(y0 - SH2 + 2));
    __syncthreads();
    // iteration t: row e = y0 - SH2 + t enters the window (inputs in buffer t&1, staged TWO iterations earlier);
    // from t = 2*SH2 on the window is full and output row y = e - SH2 is produced.  One barrier per iteration.
    // The staging of row t+2 sits right AFTER the barrier (buffer t&1 is free from there on), not before it: its wait for the
    // prefetched records is an s_waitcnt vmcnt(0) (the count cannot be known across the conditional stores), which also waits for
    // this wave's own stores of C -- issued a whole iteration earlier here, most of an iteration earlier in the old order, where
    // waves 0-2 stalled on them (~1.8 us of write latency against a 2.1 us iteration) and the other five at the barrier behind them:
    // 0.68 -> 0.59 ms with the loads removed (R3D_EXP_COST=2, wrong results), the same as with the stores removed (=1).
    const int niter = (y1 - y0) + 2 * SH2;
#pragma unroll 1
    for (int t = 0; t < niter; t++) {
        const int b = t & 1;
        int pn[NPL];
        pixel_cost(b, pn);
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) {
            const int old = ring[0][j];
            vs[2 * j] = pk_add_nc(pk_sub_nb(vs[2 * j], __builtin_amdgcn_perm(old, old, 0x0c010c00)), pn[2 * j]);
            vs[2 * j + 1] = pk_add_nc(pk_sub_nb(vs[2 * j + 1], __builtin_amdgcn_perm(old, old, 0x0c030c02)), pn[2 * j + 1]);
        }
#pragma unroll
        for (int q = 0; q + 1 < R; q++)
#pragma unroll
            for (int j = 0; j < NPL / 2; j++) ring[q][j] = ring[q + 1][j];
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) ring[R - 1][j] = __builtin_amdgcn_perm(pn[2 * j + 1], pn[2 * j], 0x06040200);
        const bool outp = t >= 2 * SH2;
        if (outp) {
            // tile layout: column-major, inside a column the two 16-byte halves of all lanes are stored as two planes
            // whose order alternates with the column parity: conflict-free for the b128 write and the b128 reads below
            *(int4 *)&sV[b][cl * DPW + (DPW / 2) * (cl & 1) + 4 * k] = make_int4(vs[0], vs[1], vs[2], vs[3]);
            *(int4 *)&sV[b][cl * DPW + (DPW / 2) * ((cl & 1) ^ 1) + 4 * k] = make_int4(vs[4], vs[5], vs[6], vs[7]);
        }
        __syncthreads();
#if R3D_COST_STAGE == 1
        commit(b);
        fetch(crow(y0 - SH2 + t + 3));
#endif
        const bool do_out = outp && is_out;
        int c[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) c[j] = 0;
        if (do_out) {
#pragma unroll
            for (int i = -SH2; i <= SH2; i++) {
                const int col = min(max(tile_x0 + cl + i, 0), g.W1 - 1) - tile_x0;
                const int4 v0 = *(const int4 *)&sV[b][col * DPW + (DPW / 2) * (col & 1) + 4 * k];
                const int4 v1 = *(const int4 *)&sV[b][col * DPW + (DPW / 2) * ((col & 1) ^ 1) + 4 * k];
                c[0] = pk_add_nc(c[0], v0.x); c[1] = pk_add_nc(c[1], v0.y); c[2] = pk_add_nc(c[2], v0.z); c[3] = pk_add_nc(c[3], v0.w);
                c[4] = pk_add_nc(c[4], v1.x); c[5] = pk_add_nc(c[5], v1.y); c[6] = pk_add_nc(c[6], v1.z); c[7] = pk_add_nc(c[7], v1.w);
            }
        }
        // between the box sum and its stores: the staging's vmcnt(0) then waits for stores that are exactly one iteration old
#if R3D_COST_STAGE == 2
        commit(b);                                      // row t+2 (fetched during iteration t-1) into the buffer this iteration just read
#if R3D_EXP_COST != 2
        fetch(crow(y0 - SH2 + t + 3));                  // row t+3, consumed by the next iteration's commit
#endif
#endif
        if (do_out) {
            if (VCH) {
                sgm_step_g<NPL, LPC, true>(LT, ltmin, c, P1pk, g.P2, k == 0, k == LPC - 1, lane_valid);
                if (y0 + t - 2 * SH2 >= out_start) {
                    const size_t ro = (size_t)(t - 2 * SH2) * rowWords;
                    int *o = optr + ro, *l = ltvol + (optr - cvol) + ro;
                    *(int4 *)o = make_int4(c[0], c[1], c[2], c[3]);
                    *(int4 *)(o + 4) = make_int4(c[4], c[5], c[6], c[7]);
                    *(int4 *)l = make_int4(LT[0], LT[1], LT[2], LT[3]);
                    *(int4 *)(l + 4) = make_int4(LT[4], LT[5], LT[6], LT[7]);
                }
            } else {
                int *o = optr + (size_t)(t - 2 * SH2) * rowWords;
#if R3D_EXP_COST == 1
                if ((c[0] ^ c[1] ^ c[2] ^ c[3] ^ c[4] ^ c[5] ^ c[6] ^ c[7]) == 0x12345678) *(int4 *)o = make_int4(c[0], c[1], c[2], c[3]);   // timing experiment: no stores
#else
                *(int4 *)o = make_int4(c[0], c[1], c[2], c[3]);
                *(int4 *)(o + 4) = make_int4(c[4], c[5], c[6], c[7]);
#endif
            }
            if (TRACK && 16 * k < g.D) {
#pragma unroll
                for (int j = 0; j < NPL; j++) cmax = pk_umax(cmax, c[j]);
            }
        }
    }
    if (TRACK) {
        const int m = wave_allmax_i32(max(cmax & 0xffff, (int)((unsigned)cmax >> 16)));
        if (lane == 0) atomicMax(maxc, m);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_hscan: one wave per image row: forward scan writes L_left, backward scan adds L_right in place.
template <int NP, int U>
__device__ __forceinline__ void load_run(int (&buf)[U][NP], const int *__restrict__ row, int x0, int dir, int W1, int NPW) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int x = min(max(x0 + dir * u, 0), W1 - 1);
#pragma unroll
        for (int j = 0; j < NP; j++) buf[u][j] = row[(size_t)x * NPW + j];
    }
}

template <int NP>
__global__ void __launch_bounds__(64) k_hscan(const int *__restrict__ cvol, int *__restrict__ hvol, SgmGeom g) {
    constexpr int NPW = NP * 64, U = 16;
    const int lane = threadIdx.x, y = blockIdx.x;
    const int *crow = cvol + (size_t)y * g.W1 * NPW + lane * NP;
    int *hrow = hvol + (size_t)y * g.W1 * NPW + lane * NP;
    const bool valid = 2 * NP * lane < g.D;
    const int P1pk = pk_dup(g.P1), W1 = g.W1;
    int P[NP], minp = 0;
    int cA[U][NP], cB[U][NP], lA[U][NP], lB[U][NP];
#pragma unroll
    for (int j = 0; j < NP; j++) P[j] = valid ? 0 : PADPK;

#define HS_FWD(buf, xb)                                                    \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                        \
        const int x = (xb) + u;                                            \
        if (x < W1) {                                                      \
            sgm_step<NP>(P, minp, buf[u], P1pk, g.P2, valid);              \
            _Pragma("unroll") for (int j = 0; j < NP; j++) hrow[(size_t)x * NPW + j] = P[j]; \
        }                                                                  \
    }
    load_run<NP, U>(cA, crow, 0, 1, W1, NPW);
    for (int x0 = 0; x0 < W1; x0 += 2 * U) {
        load_run<NP, U>(cB, crow, x0 + U, 1, W1, NPW);
        HS_FWD(cA, x0)
        load_run<NP, U>(cA, crow, x0 + 2 * U, 1, W1, NPW);
        HS_FWD(cB, x0 + U)
    }
#undef HS_FWD
    // backward: the L_left values written above are read back by the same lanes (same wave => program order)
    minp = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) P[j] = valid ? 0 : PADPK;
#define HS_BWD(cb, lb, xb)                                                 \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                        \
        const int x = (xb)-u;                                              \
        if (x >= 0) {                                                      \
            sgm_step<NP>(P, minp, cb[u], P1pk, g.P2, valid);               \
            _Pragma("unroll") for (int j = 0; j < NP; j++) hrow[(size_t)x * NPW + j] = pk_add(lb[u][j], P[j]); \
        }                                                                  \
    }
    load_run<NP, U>(cA, crow, W1 - 1, -1, W1, NPW);
    load_run<NP, U>(lA, hrow, W1 - 1, -1, W1, NPW);
    for (int x0 = W1 - 1; x0 >= 0; x0 -= 2 * U) {
        load_run<NP, U>(cB, crow, x0 - U, -1, W1, NPW);
        load_run<NP, U>(lB, hrow, x0 - U, -1, W1, NPW);
        HS_BWD(cA, lA, x0)
        load_run<NP, U>(cA, crow, x0 - 2 * U, -1, W1, NPW);
        load_run<NP, U>(lA, hrow, x0 - 2 * U, -1, W1, NPW);
        HS_BWD(cB, lB, x0 - U)
    }
#undef HS_BWD
}

// ---------------------------------------------------------------------------------------------------------
// k_vscan: one wave per (stripe, CPW adjacent cost columns): marches down the stripe's rows keeping the CPW
// L_top vectors in registers (CPW independent dependency chains => ILP), adds L_left+L_right, and does the
// winner-take-all, uniqueness test and sub-pixel interpolation of every output row.
template <int NP>
__device__ __forceinline__ int read_s(const int (&S)[NP], int d) {  // d wave-uniform
    const int l = __builtin_amdgcn_readfirstlane(d / (2 * NP));
    int v;
    if (NP == 1) v = __builtin_amdgcn_readlane(S[0], l);
    else {
        const int v0 = __builtin_amdgcn_readlane(S[0], l), v1 = __builtin_amdgcn_readlane(S[NP - 1], l);
        v = ((d >> 1) & 1) ? v1 : v0;
    }
    return (d & 1) ? hi16(v) : lo16(v);
}

template <int NP, int CPW>
__global__ void __launch_bounds__(64) k_vscan(const int *__restrict__ cvol, const int *__restrict__ cspec,
                                              const int *__restrict__ hvol, SgmGeom g, int16_t *__restrict__ raw,
                                              int16_t *__restrict__ mins) {
    constexpr int NPW = NP * 64;
    const int lane = threadIdx.x, n = blockIdx.y, xc0 = blockIdx.x * CPW;
    const size_t rowWords = (size_t)g.W1 * NPW;
    const int src_start = max(min(n * g.stripe_sz - g.overlap, g.H), 0);
    const int src_end = min((n + 1) * g.stripe_sz, g.H);
    const int out_start = min(n * g.stripe_sz, g.H);
    if (src_start >= src_end) return;  // stripe lies below the image (tiny h)
    const bool valid = 2 * NP * lane < g.D;
    const int P1pk = pk_dup(g.P1);
    int P[CPW][NP], minp[CPW], col[CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        minp[c] = 0;
        col[c] = min(xc0 + c, g.W1 - 1) * NPW + lane * NP;
#pragma unroll
        for (int j = 0; j < NP; j++) P[c][j] = valid ? 0 : PADPK;
    }
    int cc[CPW][NP], hh[CPW][NP], cn[CPW][NP], hn[CPW][NP];
    auto crow_of = [&](int y) -> const int * {
        return (n > 0 && y < src_start + g.SH2) ? cspec + ((size_t)(n - 1) * g.SH2 + (y - src_start)) * rowWords
                                                 : cvol + (size_t)y * rowWords;
    };
    auto load_row = [&](int y, int (&cb)[CPW][NP], int (&hb)[CPW][NP]) {
        const int yy = min(y, src_end - 1);
        const int *cr = crow_of(yy);
        const int *hr = hvol + (size_t)yy * rowWords;
        const bool wantH = yy >= out_start;
#pragma unroll
        for (int c = 0; c < CPW; c++)
#pragma unroll
            for (int j = 0; j < NP; j++) {
                cb[c][j] = cr[col[c] + j];
                hb[c][j] = wantH ? hr[col[c] + j] : 0;
            }
    };
    auto process = [&](int y, int (&cb)[CPW][NP], int (&hb)[CPW][NP]) {
#pragma unroll
        for (int c = 0; c < CPW; c++) sgm_step<NP>(P[c], minp[c], cb[c], P1pk, g.P2, valid);
        if (y < out_start) return;
        int odisp = g.invalid, omin = 0x7fff;
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            int S[NP];
            unsigned key = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < NP; j++) {
                S[j] = pk_add_sat(hb[c][j], P[c][j]);
                const int d0 = 2 * NP * lane + 2 * j;
                const unsigned k0 = ((unsigned)(lo16(S[j]) + 32768) << 8) | (unsigned)d0;
                const unsigned k1 = ((unsigned)(hi16(S[j]) + 32768) << 8) | (unsigned)(d0 + 1);
                key = min(key, min(k0, k1));
            }
            key = valid ? key : 0xffffffffu;
            key = wave_allmin_u32(key);
            const int best = __builtin_amdgcn_readfirstlane((int)(key & 255u));
            const int minS = __builtin_amdgcn_readfirstlane((int)(key >> 8) - 32768);
            bool bad = false;
            if (g.uniq > 0) {
                bool lb = false;
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    const int d0 = 2 * NP * lane + 2 * j;
                    lb |= (lo16(S[j]) * (100 - g.uniq) < minS * 100) && (abs(d0 - best) > 1);
                    lb |= (hi16(S[j]) * (100 - g.uniq) < minS * 100) && (abs(d0 + 1 - best) > 1);
                }
                bad = __any(lb && valid);
            }
            int dsp = g.invalid;
            if (!bad) {
                if (0 < best && best < g.D - 1) {
                    const int sm = read_s<NP>(S, best - 1), sp = read_s<NP>(S, best + 1);
                    const int den = max(sm + sp - 2 * minS, 1);
                    dsp = best * 16 + ((sm - sp) * 16 + den) / (den * 2);
                } else
                    dsp = best * 16;
                dsp += g.minD * 16;
            }
            if (lane == c && xc0 + c < g.W1) { odisp = dsp; omin = minS; }
        }
        if (lane < CPW && xc0 + lane < g.W1) {
            const size_t o = (size_t)y * g.W + g.minX1 + xc0 + lane;
            raw[o] = (int16_t)odisp;
            mins[o] = (int16_t)omin;
        }
    };
    load_row(src_start, cc, hh);
    for (int y = src_start; y < src_end; y += 2) {
        load_row(y + 1, cn, hn);
        process(y, cc, hh);
        load_row(y + 2, cc, hh);
        if (y + 1 < src_end) process(y + 1, cn, hn);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_hscan2: one wave per image row, NO spill of L_left to HBM.
//   phase 1: forward chain over the row; the state entering every K-column segment is check-pointed (8 B/lane);
//   phase 2: segments right-to-left: the segment's C values are loaded ONCE into registers, the forward chain is
//            recomputed from the checkpoint (L_left of the segment stays in registers), then the backward chain
//            runs over the same registers and streams L_left + L_right to HBM.
// HBM traffic per row: C read twice, sum written once, + 2 * 8 B * 64 * W1/K of checkpoints (6 % at K = 32).
// A tail of W1 % K columns uses the v1 scheme (L_left parked in the output row).
// PHASE 3: both phases in one launch (as described above).  PHASE 1 / 2: the two phases as separate launches, phase 1 over the
// segments [seg0, seg1) only, so that it can follow the cost kernel slab by slab (cost of slab j+1 overlaps the forward chain
// over slab j): the state entering segment seg0 is read from the checkpoint the previous launch left, the state entering seg1
// is left for the next one; the launch with seg1 == nfull also runs the tail columns.  Phase 2 only needs the checkpoints.
template <int NPL, int LPC, int K, bool PADDED, int PHASE = 3>
__global__ void __launch_bounds__(64) k_hscan2(const int *__restrict__ cvol, int *__restrict__ hvol, int *__restrict__ ckpt, SgmGeom g,
                                               int seg0, int seg1) {
    // words per column, rows per wave; CKS: words of one checkpoint = the DP/2 packed words of L + its minimum, padded to 16 B.
    // Checkpoints are addressed by IMAGE ROW, not by wave and lane, so that the two phases may use different lane mappings
    // (R3D_HSCAN_SPLIT: forward sweep with 2 rows per wave, backward sweep with 4)
    constexpr int DPW = NPL * LPC, RPW = 64 / LPC, CKS = DPW + 4;
    const int lane = threadIdx.x, k = lane % LPC;
    const int yraw = blockIdx.x * RPW + lane / LPC;
    const bool row_ok = yraw < g.H;
    const int y = min(yraw, g.H - 1);
    const int *crow = cvol + (size_t)y * g.W1 * DPW + k * NPL;
    int *hrow = hvol + (size_t)y * g.W1 * DPW + k * NPL;
    const int W1 = g.W1, nfull = W1 / K, P1pk = pk_dup(g.P1), P2pk = pk_dup(g.P2);
    int *ckrow = ckpt + (size_t)yraw * (nfull + 1) * CKS, *ck = ckrow + k * NPL;
    const bool valid = 2 * NPL * k < g.D, first = k == 0, last = k == LPC - 1;
    int P[NPL], minp = 0;
    // four rotating cost buffers: a segment is requested two rounds (2*K steps) before its first use and is never
    // touched in between, so the loads stay in flight across whole rounds
    int c0[K][NPL], c1[K][NPL], c2[K][NPL], c3[K][NPL], llA[K][NPL], llB[K][NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) P[j] = valid ? 0 : PADPK;
    // phase-1 range launches never read past their own slab (the cost kernel may still be writing the next one)
    const int seg_hi = PHASE == 1 ? min(seg1, nfull) : nfull;
    auto load_seg = [&](int (&buf)[K][NPL], int sidx) {
        const int sc = min(max(sidx, 0), max(seg_hi - 1, 0));
        const int *p = crow + (size_t)sc * K * DPW;
#pragma unroll
        for (int u = 0; u < K; u++)
#pragma unroll
            for (int j = 0; j < NPL; j++) buf[u][j] = p[(size_t)u * DPW + j];
    };
    auto save_ck = [&](int sidx) {
#pragma unroll
        for (int j = 0; j < NPL; j++) ck[(size_t)sidx * CKS + j] = P[j];
        if (first) ckrow[(size_t)sidx * CKS + DPW] = minp;
    };
    auto load_ck = [&](int sidx) {
#pragma unroll
        for (int j = 0; j < NPL; j++) P[j] = ck[(size_t)sidx * CKS + j];
        minp = ckrow[(size_t)sidx * CKS + DPW];
    };
    auto store_sum = [&](int *dst, const int (&a)[NPL], const int (&b)[NPL]) {
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < NPL; j++) dst[j] = pk_add(a[j], b[j]);
        }
    };
    // ---- phase 1: forward chain, checkpoint the state entering every segment
    auto fwd_round = [&](int (&cur)[K][NPL], int (&pre)[K][NPL], int sidx) {   // pre <- segment sidx+3
        load_seg(pre, sidx + 3);
        save_ck(sidx);
#pragma unroll
        for (int u = 0; u < K; u++) sgm_step_g<NPL, LPC, PADDED>(P, minp, cur[u], P1pk, g.P2, first, last, valid);
    };
    const int sb = PHASE == 1 ? max(seg0, 0) : 0;
    if (PHASE & 1) {
    if (PHASE == 1 && sb > 0) load_ck(sb);
    if (seg_hi > sb) {
        load_seg(c0, sb); load_seg(c1, sb + 1); load_seg(c2, sb + 2);
#pragma unroll 1
        for (int s0 = sb; s0 < seg_hi; s0 += 4) {
            fwd_round(c0, c3, s0);
            if (s0 + 1 < seg_hi) fwd_round(c1, c0, s0 + 1);
            if (s0 + 2 < seg_hi) fwd_round(c2, c1, s0 + 2);
            if (s0 + 3 < seg_hi) fwd_round(c3, c2, s0 + 3);
        }
    }
    if (PHASE == 1 && seg_hi < nfull) save_ck(seg_hi);     // state entering the next launch's first segment
    // tail columns [nfull*K, W1): forward values parked in the output row (rows beyond the image park nothing: they
    // recompute nothing useful either, their results are never stored)
    if (PHASE == 3 || seg_hi == nfull)
    for (int x = nfull * K; x < W1; x++) {
        int c[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) c[j] = crow[(size_t)x * DPW + j];
        sgm_step_g<NPL, LPC, PADDED>(P, minp, c, P1pk, g.P2, first, last, valid);
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < NPL; j++) hrow[(size_t)x * DPW + j] = P[j];
        }
    }
    }
    if (!(PHASE & 2)) return;
    // ---- phase 2: backward chain of segment s in lockstep with the recomputed forward chain of segment s-1
    int R[NPL], minr = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) R[j] = valid ? 0 : PADPK;
    for (int x = W1 - 1; x >= nfull * K; x--) {
        int c[NPL], l[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) { c[j] = crow[(size_t)x * DPW + j]; l[j] = hrow[(size_t)x * DPW + j]; }
        sgm_step_g<NPL, LPC, PADDED>(R, minr, c, P1pk, g.P2, first, last, valid);
        store_sum(hrow + (size_t)x * DPW, l, R);
    }
    // round(s): A = costs of segment s (backward), B = costs of segment s-1 (forward), pre <- segment s-3
    auto bwd_round = [&](int (&A)[K][NPL], int (&B)[K][NPL], int (&pre)[K][NPL], int sidx) {
        load_seg(pre, sidx - 3);
        int *hp = hrow + (size_t)sidx * K * DPW;
        if (sidx > 0) {
            load_ck(sidx - 1);
            int mAB = (minr & 0xffff) | (minp << 16);
#pragma unroll
            for (int u = 0; u < K; u++) {
                sgm_step_dual_g<NPL, LPC, PADDED>(R, P, mAB, A[K - 1 - u], B[u], P1pk, P2pk, first, last, valid);
                store_sum(hp + (size_t)(K - 1 - u) * DPW, llA[K - 1 - u], R);
#pragma unroll
                for (int j = 0; j < NPL; j++) llB[u][j] = P[j];
            }
            minr = lo16(mAB);
            minp = hi16(mAB);
        } else {
#pragma unroll
            for (int u = 0; u < K; u++) {
                sgm_step_g<NPL, LPC, PADDED>(R, minr, A[K - 1 - u], P1pk, g.P2, first, last, valid);
                store_sum(hp + (size_t)(K - 1 - u) * DPW, llA[K - 1 - u], R);
            }
        }
#pragma unroll
        for (int u = 0; u < K; u++)
#pragma unroll
            for (int j = 0; j < NPL; j++) llA[u][j] = llB[u][j];
    };
    if (nfull > 0) {
        load_seg(c0, nfull - 1); load_seg(c1, nfull - 2); load_seg(c2, nfull - 3);
        load_ck(nfull - 1);
#pragma unroll
        for (int u = 0; u < K; u++) {
            sgm_step_g<NPL, LPC, PADDED>(P, minp, c0[u], P1pk, g.P2, first, last, valid);
#pragma unroll
            for (int j = 0; j < NPL; j++) llA[u][j] = P[j];
        }
#pragma unroll 1
        for (int s = nfull - 1; s >= 0; s -= 4) {
            bwd_round(c0, c1, c3, s);
            if (s - 1 >= 0) bwd_round(c1, c2, c0, s - 1);
            if (s - 2 >= 0) bwd_round(c2, c3, c1, s - 2);
            if (s - 3 >= 0) bwd_round(c3, c0, c2, s - 3);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_cost_fwd (R3D_SGM_IMPL=v5): block cost AND forward chain in one kernel -- the forward sweep no longer reads C from HBM
// (2 GB per 8 MP map), and the cost arithmetic runs underneath the chain instead of in front of it.
// The chain runs along image rows, so this kernel marches ALONG rows (k_cost2 marches down them): a workgroup owns RB = 4 output
// rows for all W1 cost columns.  Wave 0 is the chain wave (the lane mapping of k_hscan2<4, 16, ...>: 16 lanes x 4 packed registers
// per row, 4 rows); waves 1..3 are producers: producer p evaluates, for every third column, the Birchfield-Tomasi pixel cost of
// the RB + 2*SH2 input rows (lane = disparity pair, so a column is one wave-wide vector per row; right-image records straight from
// global memory: a wave reads one contiguous 1 KB window per row) and the VERTICAL box sums V of the RB output rows, which go to a
// ring of columns in LDS.  The chain wave turns V into C by a sliding horizontal sum (C(x) = C(x-1) + V'(x+SW2) - V'(x-SW2-1),
// exact in int16 arithmetic), writes C to HBM for the backward phase (k_hscan2<PHASE 2>), runs the L_left step, and saves the
// checkpoints that phase needs.  One workgroup barrier per KR = 12 columns; producers run one round ahead.
// V'(xv) = V(clamp(xv, 0, W1-1)): the replicated borders of the original's box filter in cost coordinates; rows clamp to the image.
template <int SH2, bool PADDED>
__global__ void __launch_bounds__(256) k_cost_fwd(const uint2 *__restrict__ recL, const uint2 *__restrict__ recR, SgmGeom g,
                                                  int *__restrict__ cvol, int *__restrict__ hvol, int *__restrict__ ckpt) {
    constexpr int NPL = 4, LPC = 16, DPW = NPL * LPC, RB = 4, RIN = RB + 2 * SH2, R = 2 * SH2 + 1, KR = 12, NPROD = 3, K = 16, CKS = DPW + 4;
    constexpr int CPP = KR / NPROD;                       // columns a producer evaluates per round
    constexpr int SLOTS = 2 * KR + 2 * SH2 + 1;          // V columns alive at once: [x0 - SH2 - 1, x0 + 2 KR + SH2)
    constexpr int RING = 256;                             // right-image records per input row kept in LDS (a window of 128 + 2 KR + SH2 is live)
    constexpr int LSLOTS = 2 * KR + 2 * SH2 + 2;          // left-image columns kept in LDS (the prologue's KR + 2 SH2 + 1 plus one round being filed)
    // LDS per workgroup at SH2 = 2: 29 KB + 16.1 KB + 5.6 KB = 50.7 KB: three workgroups per CU (612 workgroups at C2 = 2.4 per CU)
    constexpr int NITEM = 2 * RIN * KR, NST = (NITEM + NPROD * 64 - 1) / (NPROD * 64);   // staged records per round; per producer thread
    static_assert(KR % NPROD == 0 && KR + 2 * SH2 + 1 + KR <= LSLOTS, "round geometry");
    __shared__ int sV[SLOTS][RB][DPW];
    // records of the right image, per input row: record i sits at index (i & 255) + 1; index 0 mirrors index 256, so that the pair
    // (i - 1, i) a lane needs is always 16 contiguous bytes
    __shared__ uint2 sR[RIN][RING + 2];
    // left image: per input row and column the six Birchfield-Tomasi quantities, each already splatted into both halves of a word
    // (read by every lane of a producer at the same address: a broadcast)
    __shared__ int sL[RIN][LSLOTS][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y0 = blockIdx.x * RB, W1 = g.W1;
    const int nr = (W1 + KR - 1) / KR;                    // rounds of KR chain columns
    auto slot_of = [&](int xv) { int v = (xv + SH2 + 1) % SLOTS; return v < 0 ? v + SLOTS : v; };
    auto lslot_of = [&](int xv) { int v = (xv + SH2 + 1) % LSLOTS; return v < 0 ? v + LSLOTS : v; };
    auto row_of = [&](int q) { return min(max(y0 - SH2 + q, 0), g.H - 1); };
    // highest right-image column the production of round t reads (t = -1: the prologue)
    const int rofs = g.minX1 - g.minD;
    auto rec_hi = [&](int t) { return min(KR * (t + 2) + SH2 - 1, W1 - 1) + rofs; };
    auto put_rec = [&](int q, int i, uint2 v) {
        sR[q][(i & (RING - 1)) + 1] = v;
        if ((i & (RING - 1)) == RING - 1) sR[q][0] = v;
    };
    auto get_left = [&](int q, int xv) { return recL[(size_t)row_of(q) * g.W + min(max(xv, 0), W1 - 1) + g.minX1]; };
    auto put_left = [&](int q, int xv, uint2 lr) {
        int2 *o = (int2 *)&sL[q][lslot_of(xv)][0];
        o[0] = make_int2((int)(lr.x & 255u) * 0x10001, (int)((lr.x >> 8) & 255u) * 0x10001);
        o[1] = make_int2((int)((lr.x >> 16) & 255u) * 0x10001, (int)(lr.x >> 24) * 0x10001);
        o[2] = make_int2((int)(lr.y & 255u) * 0x10001, (int)((lr.y >> 8) & 255u) * 0x10001);
    };
    // all 256 threads: what the prologue production reads (right-image window, left columns [-SH2 - 1, KR + SH2))
    {
        const int lo = rofs - (g.DP - 1) - 1, hi = rec_hi(-1), cnt = hi - lo + 1;
        for (int u = threadIdx.x; u < RIN * cnt; u += 256) {
            const int q = u / cnt, i = lo + u % cnt;
            put_rec(q, i, recR[(size_t)row_of(q) * g.W + min(max(i, 0), g.W - 1)]);
        }
        constexpr int LC = KR + 2 * SH2 + 1;
        for (int u = threadIdx.x; u < RIN * LC; u += 256) {
            const int q = u / LC, xv = -SH2 - 1 + u % LC;
            put_left(q, xv, get_left(q, xv));
        }
    }
    __syncthreads();
    if (wave > 0) {
        // ---------------------------------------------------------------- producers
        const int p = wave - 1, u0 = threadIdx.x - 64;
        // evaluation of one column, in two halves so that the LDS reads of the next column are in flight during the arithmetic of this one
        struct Col { uint2 A[RIN], B[RIN]; int xv; };   // (the left-image words are the same in every lane: read where they are used)
        auto load_col = [&](int xv, Col &c) {
            const int xc = min(max(xv, 0), W1 - 1), xi = xc + g.minX1;
            const int r = xi - g.minD - 2 * lane;        // right column of the even disparity of this lane's pair
            const int ri = r & (RING - 1);               // the pair (r - 1, r) = sR[.][ri], sR[.][ri + 1]
            c.xv = xv;
#pragma unroll
            for (int q = 0; q < RIN; q++) { c.B[q] = sR[q][ri]; c.A[q] = sR[q][ri + 1]; }
        };
        auto compute_col = [&](const Col &c) {
            int pix[RIN];
            const int ls = lslot_of(c.xv);
#pragma unroll
            for (int q = 0; q < RIN; q++) {
                const uint2 A = c.A[q], B = c.B[q];
                const int2 *lp = (const int2 *)&sL[q][ls][0];
                const int2 l0 = lp[0], l1 = lp[1], l2 = lp[2];           // (Ug, Ug0) (Ug1, Ui) (Ui0, Ui1): one address for the whole wave
                const int Vg = __builtin_amdgcn_perm(B.x, A.x, 0x0c040c00), Vg0 = __builtin_amdgcn_perm(B.x, A.x, 0x0c050c01);
                const int Vg1 = __builtin_amdgcn_perm(B.x, A.x, 0x0c060c02), Vi = __builtin_amdgcn_perm(B.x, A.x, 0x0c070c03);
                const int Vi0 = __builtin_amdgcn_perm(B.y, A.y, 0x0c040c00), Vi1 = __builtin_amdgcn_perm(B.y, A.y, 0x0c050c01);
                const int cg = bt_cost_pk(l0.x, l0.y, l1.x, Vg, Vg0, Vg1);
                const int ci = bt_cost_pk(l1.y, l2.x, l2.y, Vi, Vi0, Vi1);
                pix[q] = pk_add(cg, (ci >> 2) & 0x3fff3fff);
            }
            int v = 0;
#pragma unroll
            for (int q = 0; q < R; q++) v = pk_add(v, pix[q]);
            int *dst = &sV[slot_of(c.xv)][0][lane];
            dst[0] = v;
#pragma unroll
            for (int rr = 1; rr < RB; rr++) {
                v = pk_add(pk_sub(v, pix[rr - 1]), pix[rr - 1 + R]);
                dst[rr * DPW] = v;
            }
        };
        // a round's new records (right image: the KR columns the window advances by; left image: the KR columns the NEXT round
        // evaluates) are fetched at its start and filed at its end: a whole round hides the loads
        uint2 sv[NST];
        auto stage_fetch = [&](int t) {                  // t = -1: during the prologue
#pragma unroll
            for (int e = 0; e < NST; e++) {
                const int u = u0 + e * NPROD * 64;
                sv[e] = make_uint2(0, 0);
                if (u < RIN * KR) {
                    const int q = u / KR, i = rec_hi(t) + 1 + u % KR;
                    if (i <= rec_hi(t + 1)) sv[e] = recR[(size_t)row_of(q) * g.W + min(max(i, 0), g.W - 1)];
                } else if (u < NITEM) {
                    const int v2 = u - RIN * KR;
                    sv[e] = get_left(v2 / KR, KR * (t + 2) + SH2 + v2 % KR);
                }
            }
        };
        auto stage_file = [&](int t) {
#pragma unroll
            for (int e = 0; e < NST; e++) {
                const int u = u0 + e * NPROD * 64;
                if (u < RIN * KR) {
                    const int q = u / KR, i = rec_hi(t) + 1 + u % KR;
                    if (i <= rec_hi(t + 1)) put_rec(q, i, sv[e]);
                } else if (u < NITEM) {
                    const int v2 = u - RIN * KR;
                    put_left(v2 / KR, KR * (t + 2) + SH2 + v2 % KR, sv[e]);
                }
            }
        };
        stage_fetch(-1);
        // prologue: everything round 0 reads: xv in [-SH2 - 1, KR + SH2) (the slot at -SH2 - 1 is only ever subtracted at x = 0,
        // where the chain does not slide; it is produced anyway so that no slot is read before it was written)
        {
            Col c;
            for (int xv = -SH2 - 1 + p; xv < KR + SH2; xv += NPROD) { load_col(xv, c); compute_col(c); }
        }
        stage_file(-1);
        __syncthreads();
        for (int t = 0; t < nr; t++) {
            const int x0 = KR * (t + 1) + SH2;           // what round t + 1 needs beyond what round t had
            stage_fetch(t);
            if (t + 1 < nr) {
                Col ca, cb;
                load_col(x0 + p, ca);
#pragma unroll
                for (int c = 0; c < CPP; c += 2) {
                    if (c + 1 < CPP) load_col(x0 + p + (c + 1) * NPROD, cb);
                    compute_col(ca);
                    if (c + 2 < CPP) load_col(x0 + p + (c + 2) * NPROD, ca);
                    if (c + 1 < CPP) compute_col(cb);
                }
            }
            stage_file(t);
            __syncthreads();
        }
        return;
    }
    // -------------------------------------------------------------------- chain wave
    const int k = lane % LPC, rl = lane / LPC;
    const int yraw = y0 + rl;
    const bool row_ok = yraw < g.H;
    const int y = min(yraw, g.H - 1);
    int *crow = cvol + (size_t)y * W1 * DPW + k * NPL;
    int *hrow = hvol + (size_t)y * W1 * DPW + k * NPL;
    const int nfull = W1 / K, P1pk = pk_dup(g.P1);
    int *ckrow = ckpt + (size_t)yraw * (nfull + 1) * CKS, *ck = ckrow + k * NPL;
    const bool valid = 2 * NPL * k < g.D, first = k == 0, last = k == LPC - 1;
    int P[NPL], C[NPL], minp = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) { P[j] = valid ? 0 : PADPK; C[j] = 0; }
    auto ldv = [&](int xv) { return *(const int4 *)&sV[slot_of(xv)][rl][k * NPL]; };
    __syncthreads();                                      // the producers' prologue (the staging above was barrier one)
    for (int t = 0; t < nr; t++) {
        const int xb = KR * t;
        // the round's V columns [xb - SH2 - 1, xb + KR + SH2) into registers first: the chain below then never waits for LDS
        int4 v[KR + 2 * SH2 + 1];
#pragma unroll
        for (int i = 0; i < KR + 2 * SH2 + 1; i++) v[i] = ldv(xb - SH2 - 1 + i);
#pragma unroll
        for (int c = 0; c < KR; c++) {
            const int x = xb + c;
            if (x < W1) {
                if (x == 0) {
#pragma unroll
                    for (int i = 1; i <= 2 * SH2 + 1; i++) {   // V'(-SH2) .. V'(SH2)
                        C[0] = pk_add(C[0], v[i].x); C[1] = pk_add(C[1], v[i].y); C[2] = pk_add(C[2], v[i].z); C[3] = pk_add(C[3], v[i].w);
                    }
                } else {
                    const int4 a = v[c + 2 * SH2 + 1], b = v[c];   // V'(x + SH2), V'(x - SH2 - 1)
                    C[0] = pk_sub(pk_add(C[0], a.x), b.x); C[1] = pk_sub(pk_add(C[1], a.y), b.y);
                    C[2] = pk_sub(pk_add(C[2], a.z), b.z); C[3] = pk_sub(pk_add(C[3], a.w), b.w);
                }
                if (row_ok) *(int4 *)(crow + (size_t)x * DPW) = make_int4(C[0], C[1], C[2], C[3]);
                if ((x & (K - 1)) == 0 && x / K < nfull) {   // the state entering segment x / K (k_hscan2's checkpoint layout)
                    const int sidx = x / K;
                    *(int4 *)(ck + (size_t)sidx * CKS) = make_int4(P[0], P[1], P[2], P[3]);
                    if (first) ckrow[(size_t)sidx * CKS + DPW] = minp;
                }
                sgm_step_g<NPL, LPC, PADDED>(P, minp, C, P1pk, g.P2, first, last, valid);
                if (x >= nfull * K && row_ok) *(int4 *)(hrow + (size_t)x * DPW) = make_int4(P[0], P[1], P[2], P[3]);   // tail: L_left parked
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_hscan_fwd: phase 1 of k_hscan2 (forward chain + checkpoints) as a LOW-REGISTER kernel of its own, for the overlap with the
// cost kernel.  k_hscan2 keeps 256 + 128 registers of cost segments in flight and owns the whole register file of its SIMD, so
// none of its waves can start while cost waves occupy that SIMD (measured: cost and a k_hscan2<PHASE 1> launch on two streams
// simply ran one after the other).  Here the rotating buffers hold K = 4 columns (64 registers), the kernel is compiled for
// four waves per SIMD (<= 128 registers) and fits beside the two cost workgroups a CU holds (4 x 96 registers per SIMD).
// Segments [seg0, seg1) are counted in checkpoint intervals of KCK columns (the K of the backward-phase kernel that reads the
// checkpoints); the state entering seg0 comes from the previous launch, the state entering seg1 is left for the next one; the
// launch that reaches the last interval also runs the tail columns (parked in the output row, as k_hscan2 does).
template <int NPL, int LPC, int K, int KCK, bool PADDED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_hscan_fwd(const int *__restrict__ cvol, int *__restrict__ hvol, int *__restrict__ ckpt, SgmGeom g, int seg0, int seg1) {
    static_assert(KCK % K == 0, "checkpoint interval is a whole number of buffer segments");
    // (Raising this wave's issue priority with s_setprio 3 was tried: the cost slabs then took 313 us instead of 200-295 and
    // the forward slabs stayed at 200-430 us: the slowdown is in the memory system, not in the issue arbiter.)
    constexpr int DPW = NPL * LPC, RPW = 64 / LPC, CKS = DPW + 4, RATIO = KCK / K;   // checkpoint layout: see k_hscan2
    const int lane = threadIdx.x, k = lane % LPC;
    const int yraw = blockIdx.x * RPW + lane / LPC;
    const bool row_ok = yraw < g.H;
    const int y = min(yraw, g.H - 1);
    const int *crow = cvol + (size_t)y * g.W1 * DPW + k * NPL;
    int *hrow = hvol + (size_t)y * g.W1 * DPW + k * NPL;
    const int W1 = g.W1, nck = W1 / KCK, P1pk = pk_dup(g.P1);
    int *ckrow = ckpt + (size_t)yraw * (nck + 1) * CKS, *ck = ckrow + k * NPL;
    const bool valid = 2 * NPL * k < g.D, first = k == 0, last = k == LPC - 1;
    const int ck_lo = max(seg0, 0), ck_hi = min(seg1, nck);
    const int sb = ck_lo * RATIO, se = ck_hi * RATIO;          // sub-segments of K columns
    int P[NPL], minp = 0;
    if (ck_lo > 0) {
#pragma unroll
        for (int j = 0; j < NPL; j++) P[j] = ck[(size_t)ck_lo * CKS + j];
        minp = ckrow[(size_t)ck_lo * CKS + DPW];
    } else {
#pragma unroll
        for (int j = 0; j < NPL; j++) P[j] = valid ? 0 : PADPK;
    }
    int c0[K][NPL], c1[K][NPL], c2[K][NPL], c3[K][NPL];
    auto load_seg = [&](int (&buf)[K][NPL], int sidx) {        // never reads past this launch's slab
        const int sc = min(max(sidx, 0), max(se - 1, 0));
        const int *p = crow + (size_t)sc * K * DPW;
#pragma unroll
        for (int u = 0; u < K; u++)
#pragma unroll
            for (int j = 0; j < NPL; j++) buf[u][j] = p[(size_t)u * DPW + j];
    };
    auto round = [&](int (&cur)[K][NPL], int (&pre)[K][NPL], int sidx) {
        load_seg(pre, sidx + 3);
        if (sidx % RATIO == 0) {
#pragma unroll
            for (int j = 0; j < NPL; j++) ck[(size_t)(sidx / RATIO) * CKS + j] = P[j];
            if (first) ckrow[(size_t)(sidx / RATIO) * CKS + DPW] = minp;
        }
#pragma unroll
        for (int u = 0; u < K; u++) sgm_step_g<NPL, LPC, PADDED>(P, minp, cur[u], P1pk, g.P2, first, last, valid);
    };
    if (se > sb) {
        load_seg(c0, sb); load_seg(c1, sb + 1); load_seg(c2, sb + 2);
#pragma unroll 1
        for (int s0 = sb; s0 < se; s0 += 4) {
            round(c0, c3, s0);
            if (s0 + 1 < se) round(c1, c0, s0 + 1);
            if (s0 + 2 < se) round(c2, c1, s0 + 2);
            if (s0 + 3 < se) round(c3, c2, s0 + 3);
        }
    }
    if (ck_hi < nck) {                                          // state entering the next launch's first interval
#pragma unroll
        for (int j = 0; j < NPL; j++) ck[(size_t)ck_hi * CKS + j] = P[j];
        if (first) ckrow[(size_t)ck_hi * CKS + DPW] = minp;
        return;
    }
    for (int x = nck * KCK; x < W1; x++) {                      // tail columns: forward values parked in the output row
        int c[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) c[j] = crow[(size_t)x * DPW + j];
        sgm_step_g<NPL, LPC, PADDED>(P, minp, c, P1pk, g.P2, first, last, valid);
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < NPL; j++) hrow[(size_t)x * DPW + j] = P[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Winner-take-all + uniqueness + sub-pixel for one disparity vector in the generic mapping (group-uniform results).
// sS: per-wave LDS image (64 * NPL ints) used to fetch the winner's two neighbours with a per-group address.
template <int NPL, int LPC>
__device__ __forceinline__ void wta_eval(const int (&S)[NPL], int lane, int k, bool valid, const SgmGeom &g, float inv_a, int *sS,
                                         int &dsp_out, int &minS_out) {
    int key = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int d0 = 2 * NPL * k + 2 * j;
        const int k0 = (int)((unsigned)S[j] << 16) | d0;
        const int k1 = (S[j] & (int)0xffff0000) | (d0 + 1);
        key = min(key, min(k0, k1));
    }
    if (!valid) key = 0x7fffffff;
    key = grp_allmin<LPC>(key);
    const int best = key & 0xffff, minS = key >> 16;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPL; j++) sS[lane * NPL + j] = S[j];
    __syncthreads();
    const int gbase = (lane - k) * NPL;
    const int dm = max(best - 1, 0), dp = min(best + 1, g.D - 1);
    const int wm = sS[gbase + (dm >> 1)], wp = sS[gbase + (dp >> 1)];
    const int sm = (dm & 1) ? hi16(wm) : lo16(wm), sp = (dp & 1) ? hi16(wp) : lo16(wp);
    bool bad = false;
    if (g.uniq > 0) {
        const int T = ceil_div_small(minS * 100, 100 - g.uniq, inv_a);
        int cnt;
        if (T > 32767) cnt = valid ? 2 * NPL : 0;
        else {
            const int Tpk = pk_dup(max(T, -32768));
            int acc = 0;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                const int diff = as_i(__builtin_elementwise_sub_sat(as_s(S[j]), as_s(Tpk)));
                acc = pk_sub(acc, as_i(as_s(diff) >> (s16x2){15, 15}));
            }
            cnt = valid ? lo16(acc) + hi16(acc) : 0;
        }
        cnt = grp_allsum<LPC>(cnt);
        int win = (minS < T) ? 1 : 0;
        if (best > 0 && sm < T) win++;
        if (best < g.D - 1 && sp < T) win++;
        bad = cnt > win;
    }
    int dsp = g.invalid;
    if (!bad) {
        dsp = best * 16;
        if (0 < best && best < g.D - 1) {
            const int den = max(sm + sp - 2 * minS, 1);
            dsp += trunc_div_small((sm - sp) * 16 + den, den * 2);
        }
        dsp += g.minD * 16;
    }
    dsp_out = dsp;
    minS_out = minS;
}

// ---------------------------------------------------------------------------------------------------------
// k_hscan3 (v3): k_hscan2 whose backward phase finishes the pixel: S = (L_left + L_right) + L_top, winner-take-all,
// uniqueness and sub-pixel happen in the same step, so only the disparity and its cost leave the kernel -- the
// L_left + L_right volume is never written.  L_top comes from k_cost2<VCH>; its loads are off the dependency chain.
template <int NPL, int LPC, int K, bool PADDED>
__global__ void __launch_bounds__(64) k_hscan3(const int *__restrict__ cvol, const int *__restrict__ ltvol, int *__restrict__ tail,
                                               int *__restrict__ ckpt, SgmGeom g, float inv_a, int16_t *__restrict__ raw,
                                               int16_t *__restrict__ mins) {
    constexpr int DPW = NPL * LPC, RPW = 64 / LPC, CKW = (NPL + 1) * 64;
    __shared__ int sS[64 * NPL];
    const int lane = threadIdx.x, k = lane % LPC;
    const int yraw = blockIdx.x * RPW + lane / LPC;
    const bool row_ok = yraw < g.H;
    const int y = min(yraw, g.H - 1);
    const int *crow = cvol + (size_t)y * g.W1 * DPW + k * NPL;
    const int *lrow = ltvol + (size_t)y * g.W1 * DPW + k * NPL;
    int *trow = tail + (size_t)y * K * DPW + k * NPL;                 // parking space of the tail columns' L_left
    const int W1 = g.W1, nfull = W1 / K, P1pk = pk_dup(g.P1), P2pk = pk_dup(g.P2);
    int *ck = ckpt + (size_t)blockIdx.x * (nfull + 1) * CKW + lane * (NPL + 1);
    const bool valid = 2 * NPL * k < g.D, first = k == 0, last = k == LPC - 1;
    int P[NPL], minp = 0;
    int c0[K][NPL], c1[K][NPL], c2[K][NPL], c3[K][NPL], llA[K][NPL], llB[K][NPL], ltA[K][NPL], ltB[K][NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) P[j] = valid ? 0 : PADPK;
    auto load_seg = [&](int (&buf)[K][NPL], const int *row, int sidx) {
        const int sc = min(max(sidx, 0), max(nfull - 1, 0));
        const int *p = row + (size_t)sc * K * DPW;
#pragma unroll
        for (int u = 0; u < K; u++)
#pragma unroll
            for (int j = 0; j < NPL; j++) buf[u][j] = p[(size_t)u * DPW + j];
    };
    auto save_ck = [&](int sidx) {
#pragma unroll
        for (int j = 0; j < NPL; j++) ck[(size_t)sidx * CKW + j] = P[j];
        ck[(size_t)sidx * CKW + NPL] = minp;
    };
    auto load_ck = [&](int sidx) {
#pragma unroll
        for (int j = 0; j < NPL; j++) P[j] = ck[(size_t)sidx * CKW + j];
        minp = ck[(size_t)sidx * CKW + NPL];
    };
    // finishes cost column xcol: S, WTA, stores
    auto finish = [&](int xcol, const int (&ll)[NPL], const int (&lr)[NPL], const int (&lt)[NPL]) {
        int S[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) S[j] = pk_add_sat(pk_add(ll[j], lr[j]), lt[j]);
        int dsp, mS;
        wta_eval<NPL, LPC>(S, lane, k, valid, g, inv_a, sS, dsp, mS);
        if (first && row_ok) {
            const size_t o = (size_t)y * g.W + g.minX1 + xcol;
            raw[o] = (int16_t)dsp;
            mins[o] = (int16_t)mS;
        }
    };
    // ---- phase 1: forward chain, checkpoint the state entering every segment
    auto fwd_round = [&](int (&cur)[K][NPL], int (&pre)[K][NPL], int sidx) {
        load_seg(pre, crow, sidx + 3);
        save_ck(sidx);
#pragma unroll
        for (int u = 0; u < K; u++) sgm_step_g<NPL, LPC, PADDED>(P, minp, cur[u], P1pk, g.P2, first, last, valid);
    };
    if (nfull > 0) {
        load_seg(c0, crow, 0); load_seg(c1, crow, 1); load_seg(c2, crow, 2);
#pragma unroll 1
        for (int s0 = 0; s0 < nfull; s0 += 4) {
            fwd_round(c0, c3, s0);
            if (s0 + 1 < nfull) fwd_round(c1, c0, s0 + 1);
            if (s0 + 2 < nfull) fwd_round(c2, c1, s0 + 2);
            if (s0 + 3 < nfull) fwd_round(c3, c2, s0 + 3);
        }
    }
    for (int x = nfull * K; x < W1; x++) {
        int c[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) c[j] = crow[(size_t)x * DPW + j];
        sgm_step_g<NPL, LPC, PADDED>(P, minp, c, P1pk, g.P2, first, last, valid);
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < NPL; j++) trow[(size_t)(x - nfull * K) * DPW + j] = P[j];
        }
    }
    // ---- phase 2
    int R[NPL], minr = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) R[j] = valid ? 0 : PADPK;
    for (int x = W1 - 1; x >= nfull * K; x--) {
        int c[NPL], l[NPL], t[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            c[j] = crow[(size_t)x * DPW + j];
            l[j] = row_ok ? trow[(size_t)(x - nfull * K) * DPW + j] : 0;
            t[j] = lrow[(size_t)x * DPW + j];
        }
        sgm_step_g<NPL, LPC, PADDED>(R, minr, c, P1pk, g.P2, first, last, valid);
        finish(x, l, R, t);
    }
    // round(s): A = costs of segment s (backward), B = costs of segment s-1 (forward), pre <- segment s-3;
    //           TA = L_top of segment s, TB <- L_top of segment s-1 (one round of lead, off the chain)
    auto bwd_round = [&](int (&A)[K][NPL], int (&B)[K][NPL], int (&pre)[K][NPL], int (&TA)[K][NPL], int (&TB)[K][NPL], int sidx) {
        load_seg(pre, crow, sidx - 3);
        load_seg(TB, lrow, sidx - 1);
        if (sidx > 0) {
            load_ck(sidx - 1);
            int mAB = (minr & 0xffff) | (minp << 16);
#pragma unroll
            for (int u = 0; u < K; u++) {
                sgm_step_dual_g<NPL, LPC, PADDED>(R, P, mAB, A[K - 1 - u], B[u], P1pk, P2pk, first, last, valid);
                finish(sidx * K + K - 1 - u, llA[K - 1 - u], R, TA[K - 1 - u]);
#pragma unroll
                for (int j = 0; j < NPL; j++) llB[u][j] = P[j];
            }
            minr = lo16(mAB);
            minp = hi16(mAB);
        } else {
#pragma unroll
            for (int u = 0; u < K; u++) {
                sgm_step_g<NPL, LPC, PADDED>(R, minr, A[K - 1 - u], P1pk, g.P2, first, last, valid);
                finish(sidx * K + K - 1 - u, llA[K - 1 - u], R, TA[K - 1 - u]);
            }
        }
#pragma unroll
        for (int u = 0; u < K; u++)
#pragma unroll
            for (int j = 0; j < NPL; j++) llA[u][j] = llB[u][j];
    };
    if (nfull > 0) {
        load_seg(c0, crow, nfull - 1); load_seg(c1, crow, nfull - 2); load_seg(c2, crow, nfull - 3);
        load_seg(ltA, lrow, nfull - 1);
        load_ck(nfull - 1);
#pragma unroll
        for (int u = 0; u < K; u++) {
            sgm_step_g<NPL, LPC, PADDED>(P, minp, c0[u], P1pk, g.P2, first, last, valid);
#pragma unroll
            for (int j = 0; j < NPL; j++) llA[u][j] = P[j];
        }
#pragma unroll 1
        for (int s = nfull - 1; s >= 0; s -= 4) {
            bwd_round(c0, c1, c3, ltA, ltB, s);
            if (s - 1 >= 0) bwd_round(c1, c2, c0, ltB, ltA, s - 1);
            if (s - 2 >= 0) bwd_round(c2, c3, c1, ltA, ltB, s - 2);
            if (s - 3 >= 0) bwd_round(c3, c0, c2, ltB, ltA, s - 3);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_vscan2: vertical path + winner-take-all with 16 disparities per lane (NPL = 8): LPC = DP/16 lanes per column,
// CPW = 64/LPC adjacent columns per wave, each column an independent chain inside its lane group.  Everything after
// the path step -- argmin (32-bit keys cost<<16|d, v_min3 tree + group butterfly), uniqueness (packed compare against
// the per-column threshold, counted), sub-pixel (the owner lane picks the winner's two neighbours out of its registers,
// a 3-stage butterfly shares them) -- is per-lane VALU work: no scalar unit traffic, no LDS, no barrier.
template <int NPL, int LPC>
__global__ void __launch_bounds__(64) k_vscan2(const int *__restrict__ cvol, const int *__restrict__ cspec,
                                               const int *__restrict__ hvol, SgmGeom g, float inv_a, int16_t *__restrict__ raw,
                                               int16_t *__restrict__ mins, int col0) {
    constexpr int CPW = 64 / LPC, DPW = NPL * LPC;  // columns per wave, words per column
    static_assert(NPL == 4 || NPL == 8 || NPL == 16, "one, two or four 16-byte loads per lane");
    const int lane = threadIdx.x, k = lane % LPC, grp = lane / LPC, n = blockIdx.y;
    const int xc = col0 + blockIdx.x * CPW + grp;   // col0: first cost column of this launch (the balanced split below)
    const bool col_ok = xc < g.W1;
    const size_t rowWords = (size_t)g.W1 * DPW;
    const int src_start = max(min(n * g.stripe_sz - g.overlap, g.H), 0);
    const int src_end = min((n + 1) * g.stripe_sz, g.H);
    const int out_start = min(n * g.stripe_sz, g.H);
    if (src_start >= src_end) return;
    const bool valid = 2 * NPL * k < g.D, first = k == 0, last = k == LPC - 1;
    const int P1pk = pk_dup(g.P1);
    const size_t off = (size_t)min(xc, g.W1 - 1) * DPW + k * NPL;
    const int a = 100 - g.uniq;
    int P[NPL], minp = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) P[j] = valid ? 0 : PADPK;
    int cc[NPL], hh[NPL], cn[NPL], hn[NPL];
    auto load_row = [&](int y, int (&cb)[NPL], int (&hb)[NPL]) {
        const int yy = min(y, src_end - 1);
        const int *cr = ((n > 0 && yy < src_start + g.SH2) ? cspec + ((size_t)(n - 1) * g.SH2 + (yy - src_start)) * rowWords
                                                             : cvol + (size_t)yy * rowWords) + off;
#pragma unroll
        for (int q = 0; q < NPL / 4; q++) {
            const int4 c0 = *(const int4 *)(cr + 4 * q);
            cb[4 * q] = c0.x; cb[4 * q + 1] = c0.y; cb[4 * q + 2] = c0.z; cb[4 * q + 3] = c0.w;
        }
        if (yy >= out_start) {
            const int *hr = hvol + (size_t)yy * rowWords + off;
#pragma unroll
            for (int q = 0; q < NPL / 4; q++) {
                const int4 h0 = *(const int4 *)(hr + 4 * q);
                hb[4 * q] = h0.x; hb[4 * q + 1] = h0.y; hb[4 * q + 2] = h0.z; hb[4 * q + 3] = h0.w;
            }
        }
    };
    auto process = [&](int y, int (&cb)[NPL], int (&hb)[NPL]) {
        sgm_step_g<NPL, LPC>(P, minp, cb, P1pk, g.P2, first, last, valid);
        if (y < out_start) return;
        int S[NPL];
        int key = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            S[j] = pk_add_sat(hb[j], P[j]);
            const int d0 = 2 * NPL * k + 2 * j;
            const int k0 = (int)((unsigned)S[j] << 16) | d0;              // (cost of d0) << 16 | d0, signed order
            const int k1 = (S[j] & (int)0xffff0000) | (d0 + 1);
            key = min(key, min(k0, k1));
        }
        if (!valid) key = 0x7fffffff;
        key = grp_allmin<LPC>(key);
        const int best = key & 0xffff, minS = key >> 16;
        // the winner's neighbours S[best-1], S[best+1]: the lane that owns the disparity picks it out of its registers
        // with a select tree (the index is group-uniform, no LDS round trip, no barrier) and a 3-stage butterfly
        // hands it to the whole group
        auto fetch_s = [&](int d) -> int {
            const int j = (d >> 1) % NPL;
            int v;
            if constexpr (NPL == 16) {
                const bool b0 = j & 1, b1 = j & 2, b2 = j & 4;
                const int t0 = b0 ? S[1] : S[0], t1 = b0 ? S[3] : S[2], t2 = b0 ? S[5] : S[4], t3 = b0 ? S[7] : S[6];
                const int t4 = b0 ? S[9] : S[8], t5 = b0 ? S[11] : S[10], t6 = b0 ? S[13] : S[12], t7 = b0 ? S[15] : S[14];
                const int u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2, u2 = b1 ? t5 : t4, u3 = b1 ? t7 : t6;
                const int w0 = b2 ? u1 : u0, w1 = b2 ? u3 : u2;
                v = (j & 8) ? w1 : w0;
            } else if constexpr (NPL == 8) {
                const int t0 = (j & 1) ? S[1] : S[0], t1 = (j & 1) ? S[3] : S[2], t2 = (j & 1) ? S[5] : S[4], t3 = (j & 1) ? S[7] : S[6];
                const int u0 = (j & 2) ? t1 : t0, u1 = (j & 2) ? t3 : t2;
                v = (j & 4) ? u1 : u0;
            } else {
                const int t0 = (j & 1) ? S[1] : S[0], t1 = (j & 1) ? S[3] : S[2];
                v = (j & 2) ? t1 : t0;
            }
            const int val = (d & 1) ? hi16(v) : lo16(v);
            return grp_allmin<LPC>(((d / (2 * NPL)) == k) ? val : 0x7fffffff);
        };
        const int dm = max(best - 1, 0), dp = min(best + 1, g.D - 1);
        const int sm = fetch_s(dm), sp = fetch_s(dp);
        bool bad = false;
        if (g.uniq > 0) {
            // S*a < minS*100  <=>  S < T ; count the disparities below T, subtract those inside [best-1, best+1]
            const int T = ceil_div_small(minS * 100, a, inv_a);
            int cnt;
            if (T > 32767) cnt = valid ? 2 * NPL : 0;
            else {
                const int Tpk = pk_dup(max(T, -32768));
                int acc = 0;
#pragma unroll
                for (int j = 0; j < NPL; j++) {
                    const int diff = as_i(__builtin_elementwise_sub_sat(as_s(S[j]), as_s(Tpk)));   // < 0  <=>  S < T
                    acc = pk_sub(acc, as_i(as_s(diff) >> (s16x2){15, 15}));                        // += 1 per negative half
                }
                cnt = valid ? lo16(acc) + hi16(acc) : 0;
            }
            cnt = grp_allsum<LPC>(cnt);
            int win = (minS < T) ? 1 : 0;
            if (best > 0 && sm < T) win++;
            if (best < g.D - 1 && sp < T) win++;
            bad = cnt > win;
        }
        int dsp = g.invalid;
        if (!bad) {
            dsp = best * 16;
            if (0 < best && best < g.D - 1) {
                const int den = max(sm + sp - 2 * minS, 1);
                dsp += trunc_div_small((sm - sp) * 16 + den, den * 2);
            }
            dsp += g.minD * 16;
        }
        if (first && col_ok) {
            const size_t o = (size_t)y * g.W + g.minX1 + xc;
            raw[o] = (int16_t)dsp;
            mins[o] = (int16_t)minS;
        }
    };
    // four row buffers in rotation: a row is requested three rows (~3 x 900 cycles) before it is consumed
    int c2[NPL], h2[NPL], c3[NPL], h3[NPL];
    load_row(src_start, cc, hh);
    load_row(src_start + 1, cn, hn);
    load_row(src_start + 2, c2, h2);
#pragma unroll 1
    for (int y = src_start; y < src_end; y += 4) {
        load_row(y + 3, c3, h3);
        process(y, cc, hh);
        load_row(y + 4, cc, hh);
        if (y + 1 < src_end) process(y + 1, cn, hn);
        load_row(y + 5, cn, hn);
        if (y + 2 < src_end) process(y + 2, c2, h2);
        load_row(y + 6, c2, h2);
        if (y + 3 < src_end) process(y + 3, c3, h3);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Winner-take-all + uniqueness + sub-pixel of one disparity vector held in registers in the generic mapping (the arithmetic
// of k_vscan2's row step as a function; group-uniform results; no LDS, no barrier).
template <int NPL, int LPC>
__device__ __forceinline__ void wta_regs(const int (&S)[NPL], int k, bool valid, const SgmGeom &g, int a, float inv_a, int &dsp_out,
                                         int &minS_out) {
    int key = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int d0 = 2 * NPL * k + 2 * j;
        const int k0 = (int)((unsigned)S[j] << 16) | d0;
        const int k1 = (S[j] & (int)0xffff0000) | (d0 + 1);
        key = min(key, min(k0, k1));
    }
    if (!valid) key = 0x7fffffff;
    key = grp_allmin<LPC>(key);
    const int best = key & 0xffff, minS = key >> 16;
    auto fetch_s = [&](int d) -> int {
        const int j = (d >> 1) % NPL;
        int v;
        if constexpr (NPL == 16) {
            const bool b0 = j & 1, b1 = j & 2, b2 = j & 4;
            const int t0 = b0 ? S[1] : S[0], t1 = b0 ? S[3] : S[2], t2 = b0 ? S[5] : S[4], t3 = b0 ? S[7] : S[6];
            const int t4 = b0 ? S[9] : S[8], t5 = b0 ? S[11] : S[10], t6 = b0 ? S[13] : S[12], t7 = b0 ? S[15] : S[14];
            const int u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2, u2 = b1 ? t5 : t4, u3 = b1 ? t7 : t6;
            const int w0 = b2 ? u1 : u0, w1 = b2 ? u3 : u2;
            v = (j & 8) ? w1 : w0;
        } else if constexpr (NPL == 8) {
            const int t0 = (j & 1) ? S[1] : S[0], t1 = (j & 1) ? S[3] : S[2], t2 = (j & 1) ? S[5] : S[4], t3 = (j & 1) ? S[7] : S[6];
            const int u0 = (j & 2) ? t1 : t0, u1 = (j & 2) ? t3 : t2;
            v = (j & 4) ? u1 : u0;
        } else {
            const int t0 = (j & 1) ? S[1] : S[0], t1 = (j & 1) ? S[3] : S[2];
            v = (j & 2) ? t1 : t0;
        }
        const int val = (d & 1) ? hi16(v) : lo16(v);
        return grp_allmin<LPC>(((d / (2 * NPL)) == k) ? val : 0x7fffffff);
    };
    const int dm = max(best - 1, 0), dp = min(best + 1, g.D - 1);
    const int sm = fetch_s(dm), sp = fetch_s(dp);
    bool bad = false;
    if (g.uniq > 0) {
        const int T = ceil_div_small(minS * 100, a, inv_a);
        int cnt;
        if (T > 32767) cnt = valid ? 2 * NPL : 0;
        else {
            const int Tpk = pk_dup(max(T, -32768));
            int acc = 0;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                const int diff = as_i(__builtin_elementwise_sub_sat(as_s(S[j]), as_s(Tpk)));
                acc = pk_sub(acc, as_i(as_s(diff) >> (s16x2){15, 15}));
            }
            cnt = valid ? lo16(acc) + hi16(acc) : 0;
        }
        cnt = grp_allsum<LPC>(cnt);
        int win = (minS < T) ? 1 : 0;
        if (best > 0 && sm < T) win++;
        if (best < g.D - 1 && sp < T) win++;
        bad = cnt > win;
    }
    int dsp = g.invalid;
    if (!bad) {
        dsp = best * 16;
        if (0 < best && best < g.D - 1) {
            const int den = max(sm + sp - 2 * minS, 1);
            dsp += trunc_div_small((sm - sp) * 16 + den, den * 2);
        }
        dsp += g.minD * 16;
    }
    dsp_out = dsp;
    minS_out = minS;
}

// ---------------------------------------------------------------------------------------------------------
// k_vscan3 (R3D_SGM_IMPL=v4): the vertical pass RECOMPUTES the block cost instead of reading it.  Same result as
// k_vscan2; HBM traffic: the two record images + the L_left + L_right volume (read once) -- the 2 GB read of C is gone,
// and so are the special stripe-top rows (cspec): the stripe's own march produces them.
// Structure = k_cost2's producer loop (one workgroup per STRIPE x column tile marching down the stripe from its first
// warm-up row; pixel cost -> vertical window in registers -> LDS tile -> horizontal box sum), whose freshly summed C
// vector -- still in registers, in the 16-disparities-per-lane mapping -- feeds the L_top recurrence, S = (L_l + L_r) + L_t,
// winner-take-all, uniqueness and sub-pixel of the same lane group.  The L_l + L_r rows are requested two rows ahead
// into two register buffers (the loop is unrolled by two, so buffer and LDS slot are static).
template <int LPC, int SH2, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64) k_vscan3(const uint2 *__restrict__ recL, const uint2 *__restrict__ recR, SgmGeom g,
                                                       const int *__restrict__ hvol, float inv_a, int16_t *__restrict__ raw,
                                                       int16_t *__restrict__ mins) {
    constexpr int NPL = 8, CW = 64 / LPC, TC = NWAVE * CW, TO = TC - 2 * SH2, DP = 16 * LPC, DPW = NPL * LPC;
    constexpr int R = 2 * SH2 + 1, NRR = TC + DP, NT = NWAVE * 64;
    constexpr int SWN = NRR * 6 + (NRR / 16 + 1) * 8;   // pair-word layout: see k_cost2
    __shared__ int sW[2][SWN];
    __shared__ uint2 sL[2][TC];
    __shared__ int sV[2][TC * DPW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, k = lane % LPC, grp = lane / LPC;
    const int cl = w * CW + grp;
    const int t0 = blockIdx.x * TO;
    const int xc = min(max(t0 - SH2 + cl, 0), g.W1 - 1);
    const int x = xc + g.minX1;
    const int r_base = max(t0 - SH2, 0) + g.minX1 - g.minD - (DP - 1);
    const int ri0 = min(max(x - g.minD - 16 * k - r_base, 15), NRR - 1);
    const size_t rowWords = (size_t)g.W1 * DPW;
    const int n = blockIdx.y;
    const int y0 = max(min(n * g.stripe_sz - g.overlap, g.H), 0), y1 = min((n + 1) * g.stripe_sz, g.H);
    const int out_start = min(n * g.stripe_sz, g.H);
    if (y0 >= y1) return;
    auto crow = [&](int yy) { return min(max(yy, y0), g.H - 1); };
    const bool lane_valid = 16 * k < g.D, first = k == 0, last = k == LPC - 1;
    const int P1pk = pk_dup(g.P1), a = 100 - g.uniq;
    int LT[NPL], ltmin = 0;
#pragma unroll
    for (int j = 0; j < NPL; j++) LT[j] = lane_valid ? 0 : PADPK;

    static_assert(NRR <= NT, "one pair-word record per thread");
    uint2 pfL = make_uint2(0, 0), pfA = make_uint2(0, 0), pfB = make_uint2(0, 0);
    auto fetch = [&](int row) {
        const uint2 *lr = recL + (size_t)row * g.W, *rr = recR + (size_t)row * g.W;
        if (tid < TC) pfL = lr[min(max(t0 - SH2 + tid, 0), g.W1 - 1) + g.minX1];
        if (tid < NRR) {
            const int r = r_base + tid;
            pfA = rr[min(max(r, 0), g.W - 1)];
            pfB = rr[min(max(r - 1, 0), g.W - 1)];
        }
    };
    auto commit = [&](int b) {
        if (tid < TC) sL[b][tid] = pfL;
        if (tid < NRR) {
            const uint2 A = pfA, B = pfB;
            int *o = &sW[b][tid * 6 + (tid >> 4) * 8];
            o[0] = __builtin_amdgcn_perm(B.x, A.x, 0x0c040c00); o[1] = __builtin_amdgcn_perm(B.x, A.x, 0x0c050c01);
            o[2] = __builtin_amdgcn_perm(B.x, A.x, 0x0c060c02); o[3] = __builtin_amdgcn_perm(B.x, A.x, 0x0c070c03);
            o[4] = __builtin_amdgcn_perm(B.y, A.y, 0x0c040c00); o[5] = __builtin_amdgcn_perm(B.y, A.y, 0x0c050c01);
        }
    };
    auto pixel_cost = [&](int b, int (&pix)[NPL]) {
        const uint2 lr = sL[b][cl];
        const int Ug = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c000c00), Ug0 = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c010c01);
        const int Ug1 = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c020c02), Ui = __builtin_amdgcn_perm(lr.x, lr.x, 0x0c030c03);
        const int Ui0 = __builtin_amdgcn_perm(lr.y, lr.y, 0x0c000c00), Ui1 = __builtin_amdgcn_perm(lr.y, lr.y, 0x0c010c01);
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int rj = ri0 - 2 * j;
            typedef int v2i __attribute__((ext_vector_type(2)));           // three ds_read_b64, never ds_read2_b64: see k_cost2
            typedef const volatile __attribute__((address_space(3))) v2i lds_v2i;
            lds_v2i *p = (lds_v2i *)&sW[b][rj * 6 + (rj >> 4) * 8];
            const v2i aa = p[0], bq = p[1], c = p[2];
            const int cg = bt_cost_pk(Ug, Ug0, Ug1, aa.x, aa.y, bq.x);
            const int ci = bt_cost_pk(Ui, Ui0, Ui1, bq.y, c.x, c.y);
            pix[j] = pk_add(cg, (ci >> 2) & 0x3fff3fff);
        }
    };
    int ring[R][NPL / 2], vs[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) vs[j] = 0;
#pragma unroll
    for (int q = 0; q < R; q++)
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) ring[q][j] = 0;
    const int ocol = t0 - SH2 + cl;                        // cost column of this lane group
    const bool is_out = cl >= SH2 && cl < TC - SH2 && ocol < g.W1;
    const int tile_x0 = t0 - SH2;
    const int *hptr = hvol + (size_t)min(max(ocol, 0), g.W1 - 1) * DPW + k * NPL;
    // L_left + L_right of output row y (only rows the stripe owns are ever read)
    auto hload = [&](int y, int (&hb)[NPL]) {
        if (y >= out_start && y < y1) {
            const int4 h0 = *(const int4 *)(hptr + (size_t)y * rowWords), h1 = *(const int4 *)(hptr + (size_t)y * rowWords + 4);
            hb[0] = h0.x; hb[1] = h0.y; hb[2] = h0.z; hb[3] = h0.w; hb[4] = h1.x; hb[5] = h1.y; hb[6] = h1.z; hb[7] = h1.w;
        }
    };
    int hA[NPL], hB[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) hA[j] = hB[j] = 0;
    // iteration t: image row e = y0 - SH2 + t enters the vertical window (inputs in LDS buffer b = t & 1, staged one
    // iteration earlier); from t = 2*SH2 on the window is full and output row y = y0 + t - 2*SH2 is produced
    auto iter = [&](int t, int b, int (&hb)[NPL]) {
        int pn[NPL];
        pixel_cost(b, pn);
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) {
            const int old = ring[0][j];
            vs[2 * j] = pk_add(pk_sub(vs[2 * j], __builtin_amdgcn_perm(old, old, 0x0c010c00)), pn[2 * j]);
            vs[2 * j + 1] = pk_add(pk_sub(vs[2 * j + 1], __builtin_amdgcn_perm(old, old, 0x0c030c02)), pn[2 * j + 1]);
        }
#pragma unroll
        for (int q = 0; q + 1 < R; q++)
#pragma unroll
            for (int j = 0; j < NPL / 2; j++) ring[q][j] = ring[q + 1][j];
#pragma unroll
        for (int j = 0; j < NPL / 2; j++) ring[R - 1][j] = __builtin_amdgcn_perm(pn[2 * j + 1], pn[2 * j], 0x06040200);
        const bool outp = t >= 2 * SH2;
        if (outp) {
            *(int4 *)&sV[b][cl * DPW + (DPW / 2) * (cl & 1) + 4 * k] = make_int4(vs[0], vs[1], vs[2], vs[3]);
            *(int4 *)&sV[b][cl * DPW + (DPW / 2) * ((cl & 1) ^ 1) + 4 * k] = make_int4(vs[4], vs[5], vs[6], vs[7]);
        }
        commit(b ^ 1);
        fetch(crow(y0 - SH2 + t + 2));
        __syncthreads();
        if (outp && is_out) {
            int c[NPL];
#pragma unroll
            for (int j = 0; j < NPL; j++) c[j] = 0;
#pragma unroll
            for (int i = -SH2; i <= SH2; i++) {
                const int col = min(max(tile_x0 + cl + i, 0), g.W1 - 1) - tile_x0;
                const int4 v0 = *(const int4 *)&sV[b][col * DPW + (DPW / 2) * (col & 1) + 4 * k];
                const int4 v1 = *(const int4 *)&sV[b][col * DPW + (DPW / 2) * ((col & 1) ^ 1) + 4 * k];
                c[0] = pk_add(c[0], v0.x); c[1] = pk_add(c[1], v0.y); c[2] = pk_add(c[2], v0.z); c[3] = pk_add(c[3], v0.w);
                c[4] = pk_add(c[4], v1.x); c[5] = pk_add(c[5], v1.y); c[6] = pk_add(c[6], v1.z); c[7] = pk_add(c[7], v1.w);
            }
            sgm_step_g<NPL, LPC, true>(LT, ltmin, c, P1pk, g.P2, first, last, lane_valid);
            const int y = y0 + t - 2 * SH2;
            if (y >= out_start) {
                int S[NPL];
#pragma unroll
                for (int j = 0; j < NPL; j++) S[j] = pk_add_sat(hb[j], LT[j]);
                hload(y + 2, hb);
                int dsp, mS;
                wta_regs<NPL, LPC>(S, k, lane_valid, g, a, inv_a, dsp, mS);
                if (first) {
                    const size_t o = (size_t)y * g.W + g.minX1 + ocol;
                    raw[o] = (int16_t)dsp;
                    mins[o] = (int16_t)mS;
                }
            } else
                hload(y + 2, hb);
        }
    };
    fetch(crow(y0 - SH2));
    commit(0);
    fetch(crow(y0 - SH2 + 1));
    if (is_out) { hload(y0, hA); hload(y0 + 1, hB); }
    __syncthreads();
    const int niter = (y1 - y0) + 2 * SH2;
#pragma unroll 1
    for (int t = 0; t < niter; t += 2) {
        iter(t, 0, hA);
        if (t + 1 < niter) iter(t + 1, 1, hB);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_lrcheck: per row: rebuild OpenCV's disp2 / disp2cost scatter (lowest cost wins, among equal costs the
// LARGEST x, because the original sweeps x right-to-left with a strict '>') with one LDS atomicMin on the key
// (cost+32768)<<16 | (w-1-x), then apply the two-sided disp12MaxDiff test.  Output covers all w columns.
__global__ void __launch_bounds__(256) k_lrcheck(const int16_t *__restrict__ raw, const int16_t *__restrict__ mins, SgmGeom g,
                                                 int16_t *__restrict__ out) {
    extern __shared__ unsigned keys[];
    const int y = blockIdx.x, W = g.W;
    const int16_t *r = raw + (size_t)y * W, *m = mins + (size_t)y * W;
    for (int x = threadIdx.x; x < W; x += 256) keys[x] = 0xffffffffu;
    __syncthreads();
    // four strides of loads in flight per thread (a rolled loop waits for each 2-byte load before the next: 13 dependent round
    // trips per phase at C2's width were most of this kernel's 41 us)
    for (int x0 = g.minX1 + (int)threadIdx.x; x0 < g.maxX1; x0 += 4 * 256) {
        int d1v[4], mv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = min(x0 + 256 * u, g.maxX1 - 1);
            d1v[u] = r[x]; mv[u] = m[x];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = x0 + 256 * u, d1 = d1v[u];
            if (x >= g.maxX1 || d1 == g.invalid) continue;
            const int d = (d1 + 7) >> 4;  // best + minD (sub-pixel offset lies in [-7, 8])
            const int x2 = x - d;
            atomicMin(&keys[x2], ((unsigned)(mv[u] + 32768) << 16) | (unsigned)(W - 1 - x));
        }
    }
    __syncthreads();
    int d1n[4];
#pragma unroll
    for (int u = 0; u < 4; u++) d1n[u] = r[min((int)threadIdx.x + 256 * u, W - 1)];
    for (int xb = (int)threadIdx.x; xb < W; xb += 4 * 256) {
        int d1c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { d1c[u] = d1n[u]; d1n[u] = r[min(xb + 4 * 256 + 256 * u, W - 1)]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
        const int x = xb + 256 * u;
        if (x >= W) continue;
        int d1 = g.invalid;
        if (x >= g.minX1 && x < g.maxX1) {
            d1 = d1c[u];
            if (d1 != g.invalid) {
                const int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
                const int _x = x - _d, x_ = x - d_;
                // QUIRK: an unset disp2 entry holds the SCALED invalid marker (minD-1)*16, which passes the ">= minD"
                // test whenever minD >= 2 and then counts as a disagreeing match
                bool f1 = false, f2 = false;
                if (0 <= _x && _x < W) {
                    const int d2 = keys[_x] != 0xffffffffu ? (W - 1 - (int)(keys[_x] & 0xffffu)) - _x : g.invalid;
                    f1 = d2 >= g.minD && abs(d2 - _d) > g.d12;
                }
                if (0 <= x_ && x_ < W) {
                    const int d2 = keys[x_] != 0xffffffffu ? (W - 1 - (int)(keys[x_] & 0xffffu)) - x_ : g.invalid;
                    f2 = d2 >= g.minD && abs(d2 - d_) > g.d12;
                }
                if (f1 && f2) d1 = g.invalid;
            }
        }
        out[(size_t)y * W + x] = (int16_t)d1;
        }
    }
}

// QUIRK_SMALL_IMAGE_STRIPES (oracle/sgbm3way.c has the derivation): on images so small that a stripe's warm-up start is clamped to
// row 0 (stripe_sz < overlap: H <= 12 at blockSize 5) the original assembles that stripe's rows from the wrong rows of its private
// buffer: out[n * stripe_sz + j] = row overlap + j of a run that started at row 0, and rows the run never wrote are uninitialised
// memory there (the invalid marker here).  `run0` is the LR-checked map of ONE run over the whole image from row 0.
__global__ void __launch_bounds__(256) k_tiny_assemble(int16_t *__restrict__ lrd, const int16_t *__restrict__ run0, SgmGeom g) {
    const int x = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (x >= g.W) return;
    const int n = i / g.stripe_sz;
    if (n < 1 || n * g.stripe_sz - g.overlap >= 0) return;
    const int r = g.overlap + (i - n * g.stripe_sz), src_end = min((n + 1) * g.stripe_sz, g.H);
    lrd[(size_t)i * g.W + x] = r < src_end ? run0[(size_t)r * g.W + x] : (int16_t)g.invalid;
}

// k_median3: medianBlur(disp, 3) on int16 with replicated borders
__device__ __forceinline__ void cswap(int &a, int &b) { int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__global__ void __launch_bounds__(256) k_median3(const int16_t *__restrict__ src, int16_t *__restrict__ dst, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int16_t *r0 = src + (size_t)max(y - 1, 0) * W, *r1 = src + (size_t)y * W, *r2 = src + (size_t)min(y + 1, H - 1) * W;
    const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    int p0 = r0[xl], p1 = r0[x], p2 = r0[xr], p3 = r1[xl], p4 = r1[x], p5 = r1[xr], p6 = r2[xl], p7 = r2[x], p8 = r2[xr];
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1); cswap(p3, p4); cswap(p6, p7);
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
    cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7); cswap(p4, p2); cswap(p6, p4);
    cswap(p4, p2);
    dst[(size_t)y * W + x] = (int16_t)p4;
}

// ---------------------------------------------------------------------------------------------------------
// filterSpeckles(disp, newVal, maxSpeckleSize, maxDiff) (called by StereoSGBM.compute iff speckleWindowSize > 0, the
// depth4.py / depth_test.py parameter family): 4-connected components of pixels != newVal whose neighbouring values
// differ by <= maxDiff; components of at most maxSpeckleSize pixels are set to newVal.  The original flood-fills in
// raster order; component membership does not depend on the order, so a lock-free union-find gives the same image.
__device__ __forceinline__ int uf_find(int *L, int i) {
    int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) { i = p; p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return i;
}
__device__ __forceinline__ void uf_union(int *L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }   // link the larger root to the smaller one
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}
__global__ void __launch_bounds__(256) k_spk_init(const int16_t *__restrict__ img, int n, int newVal, int *__restrict__ L, int *__restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    L[i] = img[i] != newVal ? i : -1;
    cnt[i] = 0;
}
__global__ void __launch_bounds__(256) k_spk_merge(const int16_t *__restrict__ img, int W, int H, int newVal, int maxDiff, int *__restrict__ L) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = y * W + x, v = img[i];
    if (v == newVal) return;
    if (x + 1 < W) { const int u = img[i + 1]; if (u != newVal && abs(v - u) <= maxDiff) uf_union(L, i, i + 1); }
    if (y + 1 < H) { const int u = img[i + W]; if (u != newVal && abs(v - u) <= maxDiff) uf_union(L, i, i + W); }
}
__global__ void __launch_bounds__(256) k_spk_count(int n, int *__restrict__ L, int *__restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || L[i] < 0) return;
    const int r = uf_find(L, i);
    L[i] = r;                       // flatten (every thread only shortens its own entry)
    atomicAdd(&cnt[r], 1);
}
__global__ void __launch_bounds__(256) k_spk_apply(int16_t *__restrict__ img, int n, int newVal, int maxSize, const int *__restrict__ L,
                                                   const int *__restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    if (r >= 0 && cnt[r] <= maxSize) img[i] = (int16_t)newVal;
}

// ---------------------------------------------------------------------------------------------------------
// k_selftest: checks the cross-lane building blocks against their definition.
__global__ void __launch_bounds__(64) k_selftest(int *out) {
    const int lane = threadIdx.x;
    int bad = 0;
    const int v = (lane * 37 + 11) % 101 - 50;
    if (wave_shr1(v, -777) != (lane == 0 ? -777 : ((lane - 1) * 37 + 11) % 101 - 50)) bad |= 1;
    if (wave_shl1(v, -777) != (lane == 63 ? -777 : ((lane + 1) * 37 + 11) % 101 - 50)) bad |= 2;
    int mn = 1 << 30, mx = -(1 << 30);
    for (int l = 0; l < 64; l++) { int t = (l * 37 + 11) % 101 - 50; mn = min(mn, t); mx = max(mx, t); }
    if (wave_allmin_i32(v) != mn) bad |= 4;
    if (wave_allmax_i32(v) != mx) bad |= 8;
    {
        int first = 0;  // first lane holding the minimum: the WTA key relies on "smallest index wins among ties"
        for (int l = 63; l >= 0; l--) if ((l * 37 + 11) % 101 - 50 == mn) first = l;
        if (wave_allmin_u32((unsigned)(v + 100) << 8 | (unsigned)lane) != (((unsigned)(mn + 100) << 8) | (unsigned)first)) bad |= 16;
    }
    // packed helpers
    if (pk_add_sat(0x7fff7fff, pk_dup(600)) != 0x7fff7fff) bad |= 32;
    if (pk_min(0x00050003, (int)0xfffe0004) != (int)0xfffe0003) bad |= 64;
    if (pk_usub_sat(0x00050003, 0x00070001) != 0x00000002) bad |= 128;
    if (__builtin_amdgcn_alignbit(0x11112222, 0x33334444, 16) != 0x22223333) bad |= 256;
    if (__builtin_amdgcn_perm(0xa3a2a1a0u, 0xb3b2b1b0u, 0x0c050c01) != 0x00a100b1u) bad |= 512;
    // one sgm_step against the scalar definition, D = 128
    {
        int P[1] = {(((lane * 7) % 13) & 0xffff) | ((((lane * 5) % 11)) << 16)};
        int minp = 0;
        int allmin = wave_allmin_i32(min(lo16(P[0]), hi16(P[0])));
        minp = allmin;
        const int C[1] = {((lane + 3) & 0xffff) | ((2 * lane + 1) << 16)};
        const int P1 = 3, P2 = 9;
        // scalar reference for this lane's two disparities
        auto Lp = [&](int d) -> int { if (d < 0 || d > 127) return 32767; int l = d >> 1; return (d & 1) ? (l * 5) % 11 : (l * 7) % 13; };
        int want[2];
        for (int h = 0; h < 2; h++) {
            int d = 2 * lane + h;
            int m = min(min(Lp(d), minp + P2), min(Lp(d - 1), Lp(d + 1)) + P1);
            int c = h ? 2 * lane + 1 : lane + 3;
            want[h] = c + m - (minp + P2);
        }
        sgm_step<1>(P, minp, C, pk_dup(P1), P2, true);
        if (lo16(P[0]) != want[0] || hi16(P[0]) != want[1]) bad |= 1024;
        int wm = wave_allmin_i32(min(want[0], want[1]));
        if (minp != wm) bad |= 2048;
    }
    // dual-chain step == two single steps
    {
        int PA[1] = {(((lane * 7) % 13) & 0xffff) | ((((lane * 5) % 11)) << 16)}, PB[1] = {(((lane * 3) % 17) & 0xffff) | ((((lane * 11) % 7)) << 16)};
        int mA = wave_allmin_i32(min(lo16(PA[0]), hi16(PA[0]))), mB = wave_allmin_i32(min(lo16(PB[0]), hi16(PB[0])));
        const int CA[1] = {((lane + 3) & 0xffff) | ((2 * lane + 1) << 16)}, CB[1] = {((5 * lane + 2) & 0xffff) | ((lane + 9) << 16)};
        int dA = PA[0], dB = PB[0], mAB = (mA & 0xffff) | (mB << 16);
        for (int it = 0; it < 3; it++) {
            sgm_step_g<1, 64, false>(PA, mA, CA, pk_dup(3), 9, lane == 0, lane == 63, true);
            sgm_step_g<1, 64, false>(PB, mB, CB, pk_dup(3), 9, lane == 0, lane == 63, true);
            sgm_step_dual<false>(dA, dB, mAB, CA[0], CB[0], pk_dup(3), pk_dup(9), true);
            if (dA != PA[0] || dB != PB[0] || lo16(mAB) != mA || hi16(mAB) != mB) bad |= 1 << 23;
        }
    }
    // generic dual step (2 registers x 32 lanes) == two generic single steps
    {
        const int kk = lane % 32;
        int PA[2] = {(((lane * 7) % 13) & 0xffff) | ((((lane * 5) % 11)) << 16), ((lane * 3) % 19) | (((lane + 4) % 23) << 16)};
        int PB[2] = {(((lane * 3) % 17) & 0xffff) | ((((lane * 11) % 7)) << 16), ((lane * 9) % 29) | (((lane * 2 + 1) % 31) << 16)};
        int mA = grp_allmin<32>(min(min(lo16(PA[0]), hi16(PA[0])), min(lo16(PA[1]), hi16(PA[1]))));
        int mB = grp_allmin<32>(min(min(lo16(PB[0]), hi16(PB[0])), min(lo16(PB[1]), hi16(PB[1]))));
        const int CA[2] = {((lane + 3) & 0xffff) | ((2 * lane + 1) << 16), (lane % 7) | ((lane % 5) << 16)};
        const int CB[2] = {((5 * lane + 2) & 0xffff) | ((lane + 9) << 16), (lane % 3) | ((lane % 11) << 16)};
        int dA[2] = {PA[0], PA[1]}, dB[2] = {PB[0], PB[1]}, mAB = (mA & 0xffff) | (mB << 16);
        for (int it = 0; it < 3; it++) {
            sgm_step_g<2, 32, false>(PA, mA, CA, pk_dup(3), 9, kk == 0, kk == 31, true);
            sgm_step_g<2, 32, false>(PB, mB, CB, pk_dup(3), 9, kk == 0, kk == 31, true);
            sgm_step_dual_g<2, 32, false>(dA, dB, mAB, CA, CB, pk_dup(3), pk_dup(9), kk == 0, kk == 31, true);
            if (dA[0] != PA[0] || dA[1] != PA[1] || dB[0] != PB[0] || dB[1] != PB[1] || lo16(mAB) != mA || hi16(mAB) != mB) bad |= 1 << 24;
        }
        // and the 2x32 single step against the scalar definition
        int Q[2] = {(((lane * 7) % 13) & 0xffff) | ((((lane * 5) % 11)) << 16), ((lane * 3) % 19) | (((lane + 4) % 23) << 16)};
        auto Lp = [&](int grp, int d) -> int {
            if (d < 0 || d > 127) return 32767;
            const int l = grp * 32 + d / 4, r = (d / 2) % 2, hh = d % 2;
            const int v0 = (((l * 7) % 13) & 0xffff) | ((((l * 5) % 11)) << 16), v1 = ((l * 3) % 19) | (((l + 4) % 23) << 16);
            const int v = r ? v1 : v0;
            return hh ? hi16(v) : lo16(v);
        };
        int mq = grp_allmin<32>(min(min(lo16(Q[0]), hi16(Q[0])), min(lo16(Q[1]), hi16(Q[1]))));
        const int mq0 = mq;
        sgm_step_g<2, 32, false>(Q, mq, CA, pk_dup(3), 9, kk == 0, kk == 31, true);
        for (int r = 0; r < 2; r++)
            for (int hh = 0; hh < 2; hh++) {
                const int d = 4 * kk + 2 * r + hh, grp = lane / 32;
                const int mm = min(min(Lp(grp, d), mq0 + 9), min(Lp(grp, d - 1), Lp(grp, d + 1)) + 3);
                const int c = hh ? hi16(CA[r]) : lo16(CA[r]);
                const int got = hh ? hi16(Q[r]) : lo16(Q[r]);
                if (got != c + mm - (mq0 + 9)) bad |= 1 << 25;
            }
    }
    // generic group helpers (v2 kernels), LPC = 8 and 16
    {
        auto val = [](int l) { return (l * 37 + 11) % 101 - 50; };
        const int k8 = lane % 8, k16 = lane % 16;
        if (grp_shr1<8>(v, -777, k8 == 0) != (k8 == 0 ? -777 : val(lane - 1))) bad |= 1 << 12;
        if (grp_shl1<8>(v, -777, k8 == 7) != (k8 == 7 ? -777 : val(lane + 1))) bad |= 1 << 13;
        if (grp_shr1<16>(v, -777, k16 == 0) != (k16 == 0 ? -777 : val(lane - 1))) bad |= 1 << 14;
        if (grp_shl1<16>(v, -777, k16 == 15) != (k16 == 15 ? -777 : val(lane + 1))) bad |= 1 << 15;
        int m8 = 1 << 30, s8 = 0, m16 = 1 << 30, s16 = 0;
        for (int l = lane - k8; l < lane - k8 + 8; l++) { m8 = min(m8, val(l)); s8 += val(l); }
        for (int l = lane - k16; l < lane - k16 + 16; l++) { m16 = min(m16, val(l)); s16 += val(l); }
        if (grp_allmin<8>(v) != m8) bad |= 1 << 16;
        if (grp_allsum<8>(v) != s8) bad |= 1 << 17;
        if (grp_allmin<16>(v) != m16) bad |= 1 << 18;
        if (grp_allsum<16>(v) != s16) bad |= 1 << 19;
        if (grp_allmin<64>(v) != mn) bad |= 1 << 20;
        // exact small divisions
        const int nn = (lane - 31) * 4099 + 7, dd = 2 * (lane * 37 + 1);
        if (trunc_div_small(nn, dd) != nn / dd) bad |= 1 << 21;
        const int thr = (lane - 20) * 997 * 100, aa = 85;
        int T = ceil_div_small(thr, aa, 1.0f / 85.0f);
        if (!(T * aa >= thr && (T - 1) * aa < thr)) bad |= 1 << 22;
    }
    atomicOr(out, bad);
}

// ---------------------------------------------------------------------------------------------------------
// k_streambench (diagnostic, not on the product path): every wave streams through its own `row_bytes`-long row,
// MODE 0: 4 B/lane requests (256 B per wave-instruction, the access shape of k_hscan), MODE 1: 16 B/lane (1 KB);
// `write` adds a store of the same shape to a second buffer.  Used to price access shapes.
template <int MODE, int NTL, int NTS>
__global__ void __launch_bounds__(64) k_streambench(const int *__restrict__ in, int *__restrict__ out, size_t row_words, int write, int delay) {
    const int lane = threadIdx.x;
    const int *r = in + (size_t)blockIdx.x * row_words;
    int *o = out + (size_t)blockIdx.x * row_words;
    int acc = 0;
    if (MODE == 0) {
        for (size_t x = 0; x + 64 * 16 <= row_words; x += 64 * 16) {
            int v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = NTL ? __builtin_nontemporal_load(&r[x + u * 64 + lane]) : r[x + u * 64 + lane];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc += v[u];
                for (int d = 0; d < delay; d++) acc = __builtin_amdgcn_update_dpp(acc, acc, 0xB1, 0xf, 0xf, false) + 1;
                if (write) {
                    if (NTS) __builtin_nontemporal_store(acc, &o[x + u * 64 + lane]);
                    else o[x + u * 64 + lane] = acc;
                }
            }
        }
    } else {
        typedef int v4i __attribute__((ext_vector_type(4)));
        for (size_t x = 0; x + 256 * 4 <= row_words; x += 256 * 4) {
            v4i v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const v4i *q = (const v4i *)&r[x + u * 256 + lane * 4];
                v[u] = NTL ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc += v[u].x + v[u].y + v[u].z + v[u].w;
                for (int d = 0; d < 4 * delay; d++) acc = __builtin_amdgcn_update_dpp(acc, acc, 0xB1, 0xf, 0xf, false) + 1;
                if (write) {
                    const v4i w4 = {acc, acc, acc, acc};
                    v4i *q = (v4i *)&o[x + u * 256 + lane * 4];
                    if (NTS) __builtin_nontemporal_store(w4, q);
                    else *q = w4;
                }
            }
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

int derive_geom(r3d_ctx *ctx, const r3d_sgbm_params *p, int w, int h, SgmGeom &g) {
    if (!p) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: params is NULL");
    if (p->mode != R3D_SGBM_MODE_3WAY)
        return r3d_fail(ctx, R3D_E_UNSUPPORTED, "sgbm: only mode=STEREO_SGBM_MODE_SGBM_3WAY (2) is implemented, got %d", p->mode);
    if (p->numDisparities <= 0 || p->numDisparities % 16 != 0)
        return r3d_fail(ctx, R3D_E_BADARG, "sgbm: numDisparities must be a positive multiple of 16, got %d", p->numDisparities);
    if (p->numDisparities > 256) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "sgbm: numDisparities > 256 not supported (got %d)", p->numDisparities);
    if (p->blockSize < 1 || p->blockSize % 2 == 0 || p->blockSize > 11)
        return r3d_fail(ctx, R3D_E_BADARG, "sgbm: blockSize must be odd in [1, 11], got %d", p->blockSize);
    if (w <= 0 || h <= 0 || w > 65536) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: bad image size %dx%d", w, h);
    g.W = w; g.H = h;
    g.minD = p->minDisparity; g.D = p->numDisparities;
    g.NP = g.D <= 128 ? 1 : 2;
    g.DP = g.D <= 32 ? 32 : g.D <= 64 ? 64 : g.D <= 128 ? 128 : 256;
    const int maxD = g.minD + g.D;
    g.minX1 = maxD > 0 ? maxD : 0;
    g.maxX1 = w + (g.minD < 0 ? g.minD : 0);
    g.W1 = g.maxX1 - g.minX1;
    // W1 <= 0 (the disparity range leaves no column to match) is not an error: the caller fills the map with the invalid marker
    g.SW2 = g.SH2 = p->blockSize / 2;
    g.P1 = p->P1 > 0 ? p->P1 : 2;
    g.P2 = p->P2 > 0 ? p->P2 : 5;
    if (g.P2 < g.P1 + 1) g.P2 = g.P1 + 1;
    g.uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    if (g.uniq >= 100) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "sgbm: uniquenessRatio >= 100 not supported (got %d)", g.uniq);
    g.d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g.ftzero = (p->preFilterCap > 15 ? p->preFilterCap : 15) | 1;
    g.stripe_sz = (h + 3) / 4;
    g.overlap = (p->blockSize / 2 + 1) + (g.stripe_sz + 9) / 10;
    g.invalid = (g.minD - 1) * 16;
    // exact-int16 envelope (DESIGN.md "arithmetic envelope"): no packed add may wrap
    // static half: the block cost must fit int16 at all and P2 <= 16383; when the worst-case block cost exceeds 16383
    // the cost kernel tracks the actual maximum and the call fails loudly only if THIS image pair leaves the envelope
    const long cmax = (long)p->blockSize * p->blockSize * (2L * g.ftzero + 63);
    if (cmax > 32767 || g.P2 > 16383 || g.ftzero > 127)
        return r3d_fail(ctx, R3D_E_UNSUPPORTED,
                        "sgbm: blockSize=%d preFilterCap=%d P2=%d leave the exact int16 envelope (max block cost %ld > 32767 or P2 > 16383)",
                        p->blockSize, p->preFilterCap, g.P2, cmax);
    if ((long)g.minD * 16 - 16 < -32768 || ((long)maxD) * 16 > 32767)
        return r3d_fail(ctx, R3D_E_BADARG, "sgbm: disparity range [%d, %d) does not fit the x16 int16 output", g.minD, maxD);
    return R3D_OK;
}

// col_lo / col_hi: cost columns of a column slab (whole tiles: [ceil(col_lo / TO), ceil(col_hi / TO)) of the tile grid, so
// consecutive slabs partition the tiles); col_hi < 0 = the whole width
template <int LPC, int SH2, bool TRACK, bool VCH, int NWAVE>
int launch_cost2_n(r3d_ctx *ctx, r3d_sgm_ws &ws, const SgmGeom &g, hipStream_t st, int col_lo, int col_hi, bool spec_only = false) {
    constexpr int CW = 64 / LPC, TC = NWAVE * CW, TO = TC - 2 * SH2;
    if (TO <= 0) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "sgbm: tile too small for this block size");
    const int all_tiles = (g.W1 + TO - 1) / TO;
    const int tile_lo = col_hi < 0 ? 0 : std::min((col_lo + TO - 1) / TO, all_tiles);
    const int tile_hi = col_hi < 0 ? all_tiles : std::min((col_hi + TO - 1) / TO, all_tiles);
    const int tiles = tile_hi - tile_lo;
    if (tiles <= 0) return R3D_OK;
    // size the row bands so that one round of workgroups fills the chip (each band pays 2*SH2 extra rows of pixel cost)
    // occupancy x CU count, queried once per instantiation AND device (slab launches call this several times per map; contexts of
    // different devices / threads may race here: the slot is written once with a complete value, readers see 0 or that value)
    static std::atomic<int> slots_of[R3D_MAX_DEVICES];
    const int dev = ctx->device >= 0 && ctx->device < R3D_MAX_DEVICES ? ctx->device : 0;
    int slots = slots_of[dev].load(std::memory_order_relaxed);
    if (slots == 0) {
        int v = 1, c = 256;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, (const void *)k_cost2<LPC, SH2, TRACK, VCH, NWAVE>, NWAVE * 64, 0);
        (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, ctx->device);
        slots = (v < 1 ? 1 : v) * (c < 1 ? 256 : c);
        slots_of[dev].store(slots, std::memory_order_relaxed);
    }
    int nb = slots / tiles;
    if (nb < 1) nb = 1;
    int BAND = (g.H + nb - 1) / nb;
    if (BAND < 16) BAND = 16;
    int nMain = (g.H + BAND - 1) / BAND;
    const int nSpec = SH2 > 0 ? 3 : 0;
    if (spec_only) {      // only the stripe-top rows (cspec): the main volume comes from k_cost_fwd
        if (nSpec == 0) return R3D_OK;
        nMain = 0;
    }
    int *maxc = (int *)ws.flags.p + 8;
    if (TRACK) R3D_HIP(ctx, hipMemsetAsync(maxc, 0, 4, st));
    k_cost2<LPC, SH2, TRACK, VCH, NWAVE><<<dim3(tiles, VCH ? 4 : nMain + nSpec), NWAVE * 64, 0, st>>>(
        (const uint2 *)ws.rec_l.p, (const uint2 *)ws.rec_r.p, g, (int *)ws.cost.p, (int *)ws.cspec.p, BAND, nMain, maxc, (int *)ws.ltop.p, tile_lo);
    R3D_HIP(ctx, hipGetLastError());
    if (TRACK) {
        // data-dependent half of the exact-int16 envelope: only reached when the static bound cannot prove it
        int m = 0;
        R3D_HIP(ctx, hipMemcpyAsync(&m, maxc, 4, hipMemcpyDeviceToHost, st));
        R3D_HIP(ctx, hipStreamSynchronize(st));
        if (m > 16383)
            return r3d_fail(ctx, R3D_E_UNSUPPORTED, "sgbm: block cost reaches %d > 16383 on this image pair; outside the exact int16 envelope", m);
    }
    return R3D_OK;
}
template <int LPC, int SH2, bool TRACK, bool VCH>
int launch_cost2_t(r3d_ctx *ctx, r3d_sgm_ws &ws, const SgmGeom &g, hipStream_t st, int col_lo, int col_hi, bool spec_only = false) {
    // 8 waves = 64-column tiles (2*SH2 halo columns); R3D_COST_NWAVE=4 selects 32-column tiles for A/B measurements
    static const bool four = [] { const char *e = getenv("R3D_COST_NWAVE"); return e && !strcmp(e, "4"); }();
    if constexpr (LPC == 8 && 4 * (64 / LPC) > 2 * SH2) {
        if (four) return launch_cost2_n<LPC, SH2, TRACK, VCH, 4>(ctx, ws, g, st, col_lo, col_hi, spec_only);
    }
    return launch_cost2_n<LPC, SH2, TRACK, VCH, 8>(ctx, ws, g, st, col_lo, col_hi, spec_only);
}
template <int LPC, bool VCH>
int launch_cost2_l(r3d_ctx *ctx, r3d_sgm_ws &ws, const SgmGeom &g, hipStream_t st, int col_lo, int col_hi, bool spec_only = false) {
    const bool track = (long)(2 * g.SH2 + 1) * (2 * g.SH2 + 1) * (2L * g.ftzero + 63) > 16383;
    switch (g.SH2) {
        case 0: return launch_cost2_t<LPC, 0, false, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 1: return launch_cost2_t<LPC, 1, false, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 2: return launch_cost2_t<LPC, 2, false, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 3: return track ? launch_cost2_t<LPC, 3, true, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only) : launch_cost2_t<LPC, 3, false, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 4: return track ? launch_cost2_t<LPC, 4, true, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only) : launch_cost2_t<LPC, 4, false, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        default: return launch_cost2_t<LPC, 5, true, VCH>(ctx, ws, g, st, col_lo, col_hi, spec_only);
    }
}
int launch_cost2(r3d_ctx *ctx, r3d_sgm_ws &ws, const SgmGeom &g, hipStream_t st, bool vch, int col_lo = 0, int col_hi = -1, bool spec_only = false) {
    if (vch) return g.NP == 1 ? launch_cost2_l<8, true>(ctx, ws, g, st, 0, -1) : launch_cost2_l<16, true>(ctx, ws, g, st, 0, -1);
    switch (g.DP) {  // LPC = DP / 16 lanes per column
        case 32: return launch_cost2_l<2, false>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 64: return launch_cost2_l<4, false>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        case 128: return launch_cost2_l<8, false>(ctx, ws, g, st, col_lo, col_hi, spec_only);
        default: return launch_cost2_l<16, false>(ctx, ws, g, st, col_lo, col_hi, spec_only);
    }
}
// k_cost_fwd (v5) for the DP = 128 layout: block cost along rows + forward chain + checkpoints
int launch_cost_fwd(hipStream_t st, const r3d_sgm_ws &ws, const SgmGeom &g, int *cost, int *hsum, int *ckpt) {
    const uint2 *rl = (const uint2 *)ws.rec_l.p, *rr = (const uint2 *)ws.rec_r.p;
    const int nwg = (g.H + 3) / 4;
    const bool padded = g.D != g.DP;
#define R3D_CF(S)                                                                                           \
    case S:                                                                                                 \
        if (padded) k_cost_fwd<S, true><<<nwg, 256, 0, st>>>(rl, rr, g, cost, hsum, ckpt);                    \
        else k_cost_fwd<S, false><<<nwg, 256, 0, st>>>(rl, rr, g, cost, hsum, ckpt);                          \
        break;
    switch (g.SH2) {
        R3D_CF(0) R3D_CF(1) R3D_CF(2) R3D_CF(3) R3D_CF(4) R3D_CF(5)
        default: return -1;
    }
#undef R3D_CF
    return (int)hipGetLastError();
}


// launcher of k_vscan2.  Variants: 16 columns per wave (NPL = 16, LPC = 4: 784 waves at C2, no
// SIMD carries two; needs whole 32-disparity lanes) is the default where it applies, 8 columns (NPL = 8, LPC = 8) otherwise;
// R3D_VSCAN_COLS = 4 | 8 | 16 forces one for A/B runs (4 columns: 3136 finer-grained waves, measured 1.15 ms against 0.89 ms for 8:
// the extra cross-lane stages cost more than the better SIMD balance returns; 16 vs 8: 0.921 vs 0.956 ms interleaved).
// Balanced split (R3D_VSCAN_SPLIT=1; DEFAULT OFF: measured slower).  The kernel is one chain per wave, limited by what a CU's
// memory pipeline keeps in flight, so it lasts as long as its most loaded CU: 784 single-wave workgroups at C2 are three per CU
// on 240 CUs and FOUR on 16, and those 16 set the time (tools/gpu_balance_probe.py: 117 us per million cells with 784 waves, 108
// with exactly 768 at W = 3200).  The split cuts the launch into a main part whose wave count is a multiple of 256 and a tail of
// the remaining columns in the 8-column mapping (half the load per wave) running concurrently on the lane's second stream.
// MEASURED (interleaved on one box): 0.960 ms with the split against 0.822-0.828 ms without -- a wave of the 8-column mapping
// needs longer per row (its extra cross-lane stage), and the join waits for it; the imbalance costs less than that.
int launch_vscan2(r3d_ctx *ctx, r3d_sgm_ws &ws, hipStream_t st, const SgmGeom &g, float inv_a, const int *cost, const int *cspec, const int *hsum,
                  int16_t *raw, int16_t *mins) {
    static const int force = [] { const char *e = getenv("R3D_VSCAN_COLS"); return e ? atoi(e) : 0; }();
    static const bool split_on = [] { const char *e = getenv("R3D_VSCAN_SPLIT"); return e && !strcmp(e, "1"); }();
    if (g.DP == 32) {
        k_vscan2<4, 4><<<dim3((g.W1 + 15) / 16, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
    } else if (g.DP == 64) {
        k_vscan2<8, 4><<<dim3((g.W1 + 15) / 16, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
    } else if (g.DP == 128) {
        const bool ok16 = g.D % 32 == 0;
        if ((force == 16 || force == 0) && ok16) {
            const int groups = (g.W1 + 15) / 16, main_groups = (groups / 64) * 64;
            if (split_on && main_groups >= 64 && main_groups < groups) {
                if (!ws.aux) R3D_HIP(ctx, hipStreamCreateWithFlags(&ws.aux, hipStreamNonBlocking));
                if (!ws.vs_fork) {
                    R3D_HIP(ctx, hipEventCreateWithFlags(&ws.vs_fork, hipEventDisableTiming));
                    R3D_HIP(ctx, hipEventCreateWithFlags(&ws.vs_join, hipEventDisableTiming));
                }
                const int col0 = main_groups * 16, tail_cols = g.W1 - col0;
                R3D_HIP(ctx, hipEventRecord(ws.vs_fork, st));
                R3D_HIP(ctx, hipStreamWaitEvent(ws.aux, ws.vs_fork, 0));
                k_vscan2<8, 8><<<dim3((tail_cols + 7) / 8, 4), 64, 0, ws.aux>>>(cost, cspec, hsum, g, inv_a, raw, mins, col0);
                R3D_HIP(ctx, hipEventRecord(ws.vs_join, ws.aux));
                SgmGeom gm = g;
                k_vscan2<16, 4><<<dim3(main_groups, 4), 64, 0, st>>>(cost, cspec, hsum, gm, inv_a, raw, mins, 0);
                R3D_HIP(ctx, hipStreamWaitEvent(st, ws.vs_join, 0));
            } else {
                k_vscan2<16, 4><<<dim3(groups, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
            }
        } else if (force == 4) k_vscan2<4, 16><<<dim3((g.W1 + 3) / 4, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
        else k_vscan2<8, 8><<<dim3((g.W1 + 7) / 8, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
    } else {
        k_vscan2<8, 16><<<dim3((g.W1 + 3) / 4, 4), 64, 0, st>>>(cost, cspec, hsum, g, inv_a, raw, mins, 0);
    }
    return (int)hipGetLastError();
}

// launcher of k_vscan3 (the vertical pass that recomputes C): LPC = DP / 16 lanes per column as in k_cost2, one workgroup per
// stripe and 64 / 32-column tile (2 * SH2 halo columns)
template <int LPC>
int launch_vscan3_l(hipStream_t st, const r3d_sgm_ws &ws, const SgmGeom &g, float inv_a, const int *hsum, int16_t *raw, int16_t *mins) {
    // (4-wave workgroups of 32 columns capped at 128 registers, so that two fit a SIMD beside another kernel's waves, spill 29
    // registers and measured 3.04 ms against 1.66 ms: profiles/r04_ab_vscan3.log)
    constexpr int NWAVE = 8, CW = 64 / LPC, TC = NWAVE * CW;
    const uint2 *rl = (const uint2 *)ws.rec_l.p, *rr = (const uint2 *)ws.rec_r.p;
#define R3D_VS3(S)                                                                                                          \
    case S: {                                                                                                               \
        constexpr int TO = TC - 2 * S;                                                                                      \
        if constexpr (TO > 0)                                                                                               \
            k_vscan3<LPC, S, NWAVE><<<dim3((g.W1 + TO - 1) / TO, 4), NWAVE * 64, 0, st>>>(rl, rr, g, hsum, inv_a, raw, mins); \
        else return -1;                                                                                                     \
    } break;
    switch (g.SH2) {
        R3D_VS3(0) R3D_VS3(1) R3D_VS3(2) R3D_VS3(3) R3D_VS3(4) R3D_VS3(5)
        default: return -1;
    }
#undef R3D_VS3
    return (int)hipGetLastError();
}
int launch_vscan3(hipStream_t st, const r3d_sgm_ws &ws, const SgmGeom &g, float inv_a, const int *hsum, int16_t *raw, int16_t *mins) {
    switch (g.DP) {
        case 32: return launch_vscan3_l<2>(st, ws, g, inv_a, hsum, raw, mins);
        case 64: return launch_vscan3_l<4>(st, ws, g, inv_a, hsum, raw, mins);
        case 128: return launch_vscan3_l<8>(st, ws, g, inv_a, hsum, raw, mins);
        default: return launch_vscan3_l<16>(st, ws, g, inv_a, hsum, raw, mins);
    }
}

}  // namespace

int r3d_streambench_run(r3d_ctx *ctx, int mode, int rows, size_t row_bytes, int write, int delay, int reps, float *ms) {
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const size_t row_words = row_bytes / 4, bytes = (size_t)rows * row_bytes;
    int rc;
    if ((rc = r3d_reserve(ctx, ctx->ws[0].cost, bytes)) || (rc = r3d_reserve(ctx, ctx->ws[0].hsum, bytes))) return rc;
    hipEvent_t a, b;
    R3D_HIP(ctx, hipEventCreate(&a));
    R3D_HIP(ctx, hipEventCreate(&b));
    for (int i = 0; i < reps + 1; i++) {
        if (i == 1) R3D_HIP(ctx, hipEventRecord(a, ctx->stream));
        const int *in = (const int *)ctx->ws[0].cost.p;
        int *out = (int *)ctx->ws[0].hsum.p;
        // mode bit 0: request shape, bit 1: non-temporal loads, bit 2: non-temporal stores
        switch (mode & 7) {
            case 0: k_streambench<0, 0, 0><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 1: k_streambench<1, 0, 0><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 2: k_streambench<0, 1, 0><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 3: k_streambench<1, 1, 0><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 4: k_streambench<0, 0, 1><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 5: k_streambench<1, 0, 1><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            case 6: k_streambench<0, 1, 1><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
            default: k_streambench<1, 1, 1><<<rows, 64, 0, ctx->stream>>>(in, out, row_words, write, delay); break;
        }
    }
    R3D_HIP(ctx, hipEventRecord(b, ctx->stream));
    R3D_HIP(ctx, hipEventSynchronize(b));
    R3D_HIP(ctx, hipEventElapsedTime(ms, a, b));
    *ms /= reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return R3D_OK;
}

int r3d_speckle_run(r3d_ctx *ctx, r3d_sgm_ws &ws, hipStream_t st, int16_t *d_img, int w, int h, int newVal, int maxSize, int maxDiff) {
    const size_t n = (size_t)w * h;
    if (n > 0x7fffffff) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "filterSpeckles: image too large");
    int rc;
    if ((rc = r3d_reserve(ctx, ws.spk_l, n * 4)) || (rc = r3d_reserve(ctx, ws.spk_c, n * 4))) return rc;
    int *L = (int *)ws.spk_l.p, *C = (int *)ws.spk_c.p;
    const int nb = (int)((n + 255) / 256);
    k_spk_init<<<nb, 256, 0, st>>>(d_img, (int)n, newVal, L, C);
    k_spk_merge<<<dim3((w + 255) / 256, h), 256, 0, st>>>(d_img, w, h, newVal, maxDiff, L);
    k_spk_count<<<nb, 256, 0, st>>>((int)n, L, C);
    k_spk_apply<<<nb, 256, 0, st>>>(d_img, (int)n, newVal, maxSize, L, C);
    R3D_HIP(ctx, hipGetLastError());
    return R3D_OK;
}

int r3d_selftest_run(r3d_ctx *ctx) {
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = r3d_reserve(ctx, ctx->ws[0].flags, 256)) return rc;
    R3D_HIP(ctx, hipMemsetAsync(ctx->ws[0].flags.p, 0, 4, ctx->stream));
    k_selftest<<<1, 64, 0, ctx->stream>>>((int *)ctx->ws[0].flags.p);
    R3D_HIP(ctx, hipGetLastError());
    int bad = -1;
    R3D_HIP(ctx, hipMemcpyAsync(&bad, ctx->ws[0].flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != 0) return r3d_fail(ctx, R3D_E_HIP, "selftest: cross-lane primitive mismatch, mask=0x%x", bad);
    return R3D_OK;
}

// pass 0: the whole call.  pass 1 (only from pass 0, tiny images): ONE run over the whole image from row 0 (a single stripe), up to
// and including the LR check, its map left in ws.lrd2 -- the rows QUIRK_SMALL_IMAGE_STRIPES hands out in place of a clamped
// stripe's own.
static int sgm_run_impl(r3d_ctx *ctx, int lane, hipStream_t st, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right,
                        int w, int h, int stride, int16_t *d_disp, int pass);
int r3d_sgm_run(r3d_ctx *ctx, int lane, hipStream_t st, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right,
                int w, int h, int stride, int16_t *d_disp) {
    return sgm_run_impl(ctx, lane, st, p, d_left, d_right, w, h, stride, d_disp, 0);
}
static int sgm_run_impl(r3d_ctx *ctx, int lane, hipStream_t st, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right,
                        int w, int h, int stride, int16_t *d_disp, int pass) {
    r3d_sgm_ws &ws = ctx->ws[lane];
    if (ctx->poisoned) return r3d_fail(ctx, R3D_E_HIP, "context poisoned by an earlier timed-out call: destroy it");
    SgmGeom g;
    if (int rc = derive_geom(ctx, p, w, h, g)) return rc;
    if (!d_left || !d_right || !d_disp) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: null image pointer");
    if (stride < w) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: stride %d < width %d", stride, w);
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    if (g.W1 <= 0) {  // minX1 >= maxX1: like the original, an all-invalid map, no error
        const size_t np = (size_t)w * h;
        k_fill_s16<<<(unsigned)std::min<size_t>((np + 255) / 256, 4096), 256, 0, st>>>(d_disp, np, (int16_t)((g.minD - 1) * 16));
        R3D_HIP(ctx, hipGetLastError());
        ctx->last_w = w; ctx->last_h = h; ctx->last_w1 = 0; ctx->last_dp = 0;
        return R3D_OK;
    }
    // R3D_SGM_IMPL (read below as well): the v1 and v3 kernel generations only know 128 / 256 slots per column
    {
        const char *e = getenv("R3D_SGM_IMPL");
        if (e && (!strcmp(e, "v1") || !strcmp(e, "v3"))) g.DP = g.NP * 128;
    }
    const int NPW = g.DP / 2;   // 32-bit words (disparity pairs) per cost-volume column
    const size_t npix = (size_t)w * h;
    const size_t rowBytes = (size_t)g.W1 * NPW * 4;
    const size_t volBytes = rowBytes * h;
    int rc;
    if ((rc = r3d_reserve(ctx, ws.rec_l, npix * 8))) return rc;
    if ((rc = r3d_reserve(ctx, ws.rec_r, npix * 8))) return rc;
    if ((rc = r3d_reserve(ctx, ws.cost, volBytes))) return rc;
    if ((rc = r3d_reserve(ctx, ws.cspec, rowBytes * 3 * (g.SH2 > 0 ? g.SH2 : 1)))) return rc;
    if ((rc = r3d_reserve(ctx, ws.raw, npix * 2))) return rc;
    if ((rc = r3d_reserve(ctx, ws.mins, npix * 2))) return rc;
    if ((rc = r3d_reserve(ctx, ws.lrd, npix * 2))) return rc;
    // R3D_QUIRK_SMALL_IMAGE_STRIPES=0 places every row of a tiny image at its own position (the rounds 1-3 behaviour)
    static const bool tiny_quirk = [] { const char *e = getenv("R3D_QUIRK_SMALL_IMAGE_STRIPES"); return !(e && !strcmp(e, "0")); }();
    const bool tiny = pass == 0 && tiny_quirk && g.stripe_sz < h && g.stripe_sz - g.overlap < 0;   // stripe 1 exists and its start is clamped
    if (pass == 1) g.stripe_sz = h;                    // one stripe: rows [0, h) from row 0 (the other three own no row)
    if ((tiny || pass == 1) && (rc = r3d_reserve(ctx, ws.lrd2, npix * 2))) return rc;
    ctx->last_w = w; ctx->last_h = h; ctx->last_w1 = g.W1; ctx->last_dp = NPW * 2;
    if ((rc = r3d_reserve(ctx, ws.flags, 256))) return rc;
    r3d_prof_begin(ctx, ws);

    r3d_prof_mark(ctx, ws, st, "prefilter");
    k_prefilter<<<dim3((w + 255) / 256, (h + PF_ROWS - 1) / PF_ROWS, 2), 256, 0, st>>>(d_left, d_right, stride, w, h, g.ftzero, (uint2 *)ws.rec_l.p, (uint2 *)ws.rec_r.p);
    R3D_HIP(ctx, hipGetLastError());

    // implementation generations kept side by side for A/B measurements: R3D_SGM_IMPL = v1 | v2 (default) | v3.
    // v3 (L_top fused into the cost kernel, WTA fused into hscan) moves 2.7 GB less but measured 5.0 ms against 3.45 ms
    // for v2 on C2, so it is not the default.
    static const int impl = [] { const char *e = getenv("R3D_SGM_IMPL"); return !e ? 2 : !strcmp(e, "v1") ? 1 : !strcmp(e, "v3") ? 3 : !strcmp(e, "v4") ? 4 : !strcmp(e, "v5") ? 5 : 2; }();
    const bool use_v1 = impl == 1;
    ctx->last_impl = impl;
    const float inv_a = 1.0f / (float)(100 - g.uniq);
    if (impl == 3) {
        if ((rc = r3d_reserve(ctx, ws.ltop, volBytes))) return rc;
        r3d_prof_mark(ctx, ws, st, "cost_vpath");
        if ((rc = launch_cost2(ctx, ws, g, st, true))) return rc;
        r3d_prof_mark(ctx, ws, st, "hscan_wta");
        constexpr int K1 = 12, K2 = 6;
        const int K = g.NP == 1 ? K1 : K2;
        const bool padded = g.D != 128 * g.NP;
        const int nwaves = (h + 1) / 2;
        if ((rc = r3d_reserve(ctx, ws.ckpt, (size_t)nwaves * (g.W1 / K + 1) * (2 * g.NP + 1) * 64 * 4))) return rc;
        if ((rc = r3d_reserve(ctx, ws.hsum, (size_t)h * K * NPW * 4 + 4096))) return rc;   // tail parking only
        const int *cp = (const int *)ws.cost.p, *lp = (const int *)ws.ltop.p;
        int *tp = (int *)ws.hsum.p, *kp = (int *)ws.ckpt.p;
        int16_t *rp = (int16_t *)ws.raw.p, *mp = (int16_t *)ws.mins.p;
        if (g.NP == 1) {
            if (padded) k_hscan3<2, 32, K1, true><<<nwaves, 64, 0, st>>>(cp, lp, tp, kp, g, inv_a, rp, mp);
            else k_hscan3<2, 32, K1, false><<<nwaves, 64, 0, st>>>(cp, lp, tp, kp, g, inv_a, rp, mp);
        } else {
            if (padded) k_hscan3<4, 32, K2, true><<<nwaves, 64, 0, st>>>(cp, lp, tp, kp, g, inv_a, rp, mp);
            else k_hscan3<4, 32, K2, false><<<nwaves, 64, 0, st>>>(cp, lp, tp, kp, g, inv_a, rp, mp);
        }
        R3D_HIP(ctx, hipGetLastError());
    } else {
    if ((rc = r3d_reserve(ctx, ws.hsum, volBytes))) return rc;
    if (use_v1) {
        r3d_prof_mark(ctx, ws, st, "cost");
        const int TX = 16, BAND = 64;
        int RING = 8;
        while (RING < 2 * g.SW2 + 2) RING *= 2;
        const int NR = COST_NW + 2 * g.SH2;
        const size_t lds = ((size_t)NR * TX + (size_t)COST_NW * RING) * NPW * 4;
        const int nMain = (h + BAND - 1) / BAND;
        const int nSpec = g.SH2 > 0 ? 3 : 0;
        dim3 grid((g.W1 + TX - 1) / TX, nMain + nSpec);
        if (g.NP == 1) {
            R3D_HIP(ctx, hipFuncSetAttribute((const void *)k_cost<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_cost<1><<<grid, COST_NW * 64, lds, st>>>((const uint2 *)ws.rec_l.p, (const uint2 *)ws.rec_r.p, g, (int *)ws.cost.p, (int *)ws.cspec.p, TX, BAND, nMain, RING);
        } else {
            R3D_HIP(ctx, hipFuncSetAttribute((const void *)k_cost<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_cost<2><<<grid, COST_NW * 64, lds, st>>>((const uint2 *)ws.rec_l.p, (const uint2 *)ws.rec_r.p, g, (int *)ws.cost.p, (int *)ws.cspec.p, TX, BAND, nMain, RING);
        }
        R3D_HIP(ctx, hipGetLastError());
    }
    // Column-slab overlap of the cost kernel with the forward phase of the horizontal scan (R3D_SGM_OVERLAP = number of slabs;
    // DEFAULT OFF: measured slower, see below).  The forward chain at column x only needs C of columns <= x, so the
    // cost kernel runs slab by slab (left to right) on a second stream while k_hscan2<PHASE 1> follows one slab behind on the
    // map's own stream; the backward phase (PHASE 2), which needs every checkpoint, and everything after it stay as they were.
    // A kernel of this kind lasts as long as ONE of its waves' chains, whatever the number of waves, so splitting by ROWS
    // (or running the vertical scan behind the backward sweep) shortens nothing; only the cost kernel, whose work is not a
    // chain, can hide behind a chain kernel.
    // MEASURED (round 2, profiles/r02_overlap_trace.txt, interleaved A/B on one box): the two kernels do run concurrently, but
    // each then takes 1.5-3x its time alone (cost slab 176 -> 200-310 us, forward slab 120-140 -> 175-455 us), whether the
    // forward launches own their SIMD (k_hscan2<PHASE 1>, 280 registers) or share it (k_hscan_fwd, 97 registers) and whatever
    // the chain wave's issue priority: a write stream and a latency-bound read chain interleaved in HBM cost each other more
    // than running back to back.  320-334 maps/s sequential vs 305-331 overlapped, so the sequential order stays the default.
    static const int n_slabs = [] { const char *e = getenv("R3D_SGM_OVERLAP"); const int v = e ? atoi(e) : 0; return v < 2 ? 0 : (v > R3D_SGM_SLABS ? R3D_SGM_SLABS : v); }();
    static const bool rows2_env = [] { const char *e = getenv("R3D_HSCAN_ROWS"); return e && !strcmp(e, "2"); }();
    constexpr int KOV = 16;
    // R3D_SGM_FWD=wide: the forward launches use k_hscan2<PHASE 1> (whole register file: cannot share a SIMD with cost waves)
    static const bool lowreg = [] { const char *e = getenv("R3D_SGM_FWD"); return !(e && !strcmp(e, "wide")); }();
    const bool track = (long)(2 * g.SH2 + 1) * (2 * g.SH2 + 1) * (2L * g.ftzero + 63) > 16383;
    const bool overlapped = !use_v1 && n_slabs >= 2 && g.DP == 128 && !rows2_env && !track && g.W1 / KOV >= 8 * n_slabs;
    // v5: cost + forward chain fused (k_cost_fwd), then the backward phase of k_hscan2; D <= 128 layouts with 128 slots only, and only
    // where the static envelope bound holds (no TRACK pass); otherwise the v2 kernels below
    const bool fused_fwd = impl == 5 && g.DP == 128 && !track && g.W1 / 16 >= 1;
    if (fused_fwd) {
        constexpr int KF = 16;
        const int nfull = g.W1 / KF, nwaves = (h + 3) / 4;
        const bool padded = g.D != g.DP;
        if ((rc = r3d_reserve(ctx, ws.ckpt, (size_t)(h + 8) * (nfull + 1) * (NPW + 4) * 4))) return rc;
        const int *cp = (const int *)ws.cost.p;
        int *hp = (int *)ws.hsum.p, *kp = (int *)ws.ckpt.p;
        r3d_prof_mark(ctx, ws, st, "cost_fwd");
        if ((rc = launch_cost2(ctx, ws, g, st, false, 0, -1, true))) return rc;        // stripe-top rows (cspec) for the vertical pass
        if (int e = launch_cost_fwd(st, ws, g, (int *)ws.cost.p, hp, kp))
            return e < 0 ? r3d_fail(ctx, R3D_E_UNSUPPORTED, "k_cost_fwd: no instantiation for this block size")
                         : r3d_fail(ctx, R3D_E_HIP, "k_cost_fwd launch failed: %s", hipGetErrorString((hipError_t)e));
        r3d_prof_mark(ctx, ws, st, "hscan_bwd");
        if (padded) k_hscan2<4, 16, KF, true, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
        else k_hscan2<4, 16, KF, false, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
    } else
    if (overlapped) {
        if (!ws.aux) R3D_HIP(ctx, hipStreamCreateWithFlags(&ws.aux, hipStreamNonBlocking));
        if (!ws.slab_ev[0])
            for (int j = 0; j <= R3D_SGM_SLABS; j++) R3D_HIP(ctx, hipEventCreateWithFlags(&ws.slab_ev[j], hipEventDisableTiming));
        const int nfull = g.W1 / KOV, nwaves = (h + 3) / 4;
        const bool padded = g.D != g.DP;
        if ((rc = r3d_reserve(ctx, ws.ckpt, (size_t)(h + 8) * (nfull + 1) * (NPW + 4) * 4))) return rc;
        const int *cp = (const int *)ws.cost.p;
        int *hp = (int *)ws.hsum.p, *kp = (int *)ws.ckpt.p;
        r3d_prof_mark(ctx, ws, st, "cost+hscan_fwd");
        R3D_HIP(ctx, hipEventRecord(ws.slab_ev[R3D_SGM_SLABS], st));            // fork: prefilter done (and the previous map's readers of C)
        R3D_HIP(ctx, hipStreamWaitEvent(ws.aux, ws.slab_ev[R3D_SGM_SLABS], 0));
        int seg_lo = 0;
        for (int j = 0; j < n_slabs; j++) {
            const int seg_hi = j == n_slabs - 1 ? nfull : (int)((long)nfull * (j + 1) / n_slabs);
            const int col_lo = seg_lo * KOV, col_hi = j == n_slabs - 1 ? g.W1 : seg_hi * KOV;
            if ((rc = launch_cost2(ctx, ws, g, ws.aux, false, col_lo, col_hi))) return rc;
            R3D_HIP(ctx, hipEventRecord(ws.slab_ev[j], ws.aux));
            seg_lo = seg_hi;
        }
        seg_lo = 0;
        for (int j = 0; j < n_slabs; j++) {
            const int seg_hi = j == n_slabs - 1 ? nfull : (int)((long)nfull * (j + 1) / n_slabs);
            R3D_HIP(ctx, hipStreamWaitEvent(st, ws.slab_ev[j], 0));
            if (lowreg) {
                if (padded) k_hscan_fwd<4, 16, 4, KOV, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, seg_lo, seg_hi);
                else k_hscan_fwd<4, 16, 4, KOV, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, seg_lo, seg_hi);
            } else {
                if (padded) k_hscan2<4, 16, KOV, true, 1><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, seg_lo, seg_hi);
                else k_hscan2<4, 16, KOV, false, 1><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, seg_lo, seg_hi);
            }
            seg_lo = seg_hi;
        }
        R3D_HIP(ctx, hipGetLastError());
        r3d_prof_mark(ctx, ws, st, "hscan_bwd");
        if (padded) k_hscan2<4, 16, KOV, true, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
        else k_hscan2<4, 16, KOV, false, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
    } else {
    if (!use_v1) {
        r3d_prof_mark(ctx, ws, st, "cost");
        if ((rc = launch_cost2(ctx, ws, g, st, false))) return rc;
    }
    r3d_prof_mark(ctx, ws, st, "hscan");
    if (use_v1) {
        if (g.NP == 1) k_hscan<1><<<h, 64, 0, st>>>((const int *)ws.cost.p, (int *)ws.hsum.p, g);
        else k_hscan<2><<<h, 64, 0, st>>>((const int *)ws.cost.p, (int *)ws.hsum.p, g);
    } else {
        // D <= 128: 4 registers x 16 lanes per row = 4 rows per wave: 612 waves <= 1024 SIMDs, so no SIMD carries two
        // waves (with 2 rows per wave 1224 waves left 200 SIMDs with double work: makespan 2x the mean);
        // R3D_HSCAN_ROWS=2 selects the 2-rows-per-wave instantiation for A/B.  D <= 256: 4 x 32 (2 rows per wave).
        static const bool rows2 = [] { const char *e = getenv("R3D_HSCAN_ROWS"); return e && !strcmp(e, "2"); }();
        constexpr int K1 = 12, K1b = 16, K2 = 6;
        const bool padded = g.D != g.DP;
        // DP = 64 / 32: still 16 lanes per row and 4 rows per wave (612 waves at C2 height), with 2 / 1 registers per lane and
        // proportionally longer segments so that a segment stays 128 registers of loads in flight
        constexpr int K64 = 16, K32 = 32;
        const bool small = g.DP < 128;
        const bool four = (g.DP == 128 && !rows2) || small;
        const int rpw = four ? 4 : 2, npl = g.DP == 32 ? 1 : g.DP == 64 ? 2 : four ? 4 : 2 * g.NP;
        // R3D_HSCAN_K=8: 8-column segments for the D <= 128 layout (half the registers of the default 16: two waves fit a SIMD); A/B
        static const bool k8_env = [] { const char *e = getenv("R3D_HSCAN_K"); return e && !strcmp(e, "8"); }();
        const bool k8 = k8_env && g.DP == 128 && four;
        const int K = g.DP == 32 ? K32 : g.DP == 64 ? K64 : four ? (k8 ? 8 : K1b) : (g.NP == 1 ? K1 : K2);
        const int nwaves = (h + rpw - 1) / rpw;
        if ((rc = r3d_reserve(ctx, ws.ckpt, (size_t)(h + 8) * (g.W1 / K + 1) * (NPW + 4) * 4))) return rc;   // by image row (k_hscan2)
        const int *cp = (const int *)ws.cost.p;
        int *hp = (int *)ws.hsum.p, *kp = (int *)ws.ckpt.p;
        // R3D_HSCAN_SPLIT=1: the two phases as two launches with different lane mappings: the forward sweep (read-only, bound by
        // the issue rate of ONE wave per SIMD at 4 rows per wave) with 2 rows per wave = 1224 waves, two per SIMD where they
        // meet, so that one wave's dependency stalls are the other's issue slots; the backward sweep (bandwidth-bound) as it was
        static const bool split = [] { const char *e = getenv("R3D_HSCAN_SPLIT"); return e && !strcmp(e, "1"); }();
        (void)npl;
        if (split && g.DP == 128 && four && g.W1 / K1b >= 1) {
            const int nfull = g.W1 / K1b, nw2 = (h + 1) / 2;
            if (padded) {
                k_hscan2<2, 32, K1b, true, 1><<<nw2, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
                k_hscan2<4, 16, K1b, true, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
            } else {
                k_hscan2<2, 32, K1b, false, 1><<<nw2, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
                k_hscan2<4, 16, K1b, false, 2><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, nfull);
            }
        } else
        if (g.DP == 32) {
            if (padded) k_hscan2<1, 16, K32, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<1, 16, K32, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        } else if (g.DP == 64) {
            if (padded) k_hscan2<2, 16, K64, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<2, 16, K64, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        } else if (four && k8) {
            if (padded) k_hscan2<4, 16, 8, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<4, 16, 8, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        } else if (four) {
            if (padded) k_hscan2<4, 16, K1b, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<4, 16, K1b, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        } else if (g.NP == 1) {
            if (padded) k_hscan2<2, 32, K1, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<2, 32, K1, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        } else {
            if (padded) k_hscan2<4, 32, K2, true><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
            else k_hscan2<4, 32, K2, false><<<nwaves, 64, 0, st>>>(cp, hp, kp, g, 0, 0);
        }
    }
    }  // !overlapped
    R3D_HIP(ctx, hipGetLastError());

    r3d_prof_mark(ctx, ws, st, "vscan_wta");
    // raw must read INVALID wherever the scan does not write (columns outside [minX1, maxX1))
    {
        constexpr int CPW = 8;
        dim3 grid((g.W1 + CPW - 1) / CPW, 4);
        if (impl == 4) {
            // v4: the vertical pass recomputes C from the record images (k_vscan3); cost / cspec are not read again
            if (int e = launch_vscan3(st, ws, g, inv_a, (const int *)ws.hsum.p, (int16_t *)ws.raw.p, (int16_t *)ws.mins.p))
                return e < 0 ? r3d_fail(ctx, R3D_E_UNSUPPORTED, "k_vscan3: no instantiation for this block size")
                             : r3d_fail(ctx, R3D_E_HIP, "k_vscan3 launch failed: %s", hipGetErrorString((hipError_t)e));
        } else if (!use_v1) {
            // mapping per disparity-slot layout: see launch_vscan2
            if (int e = launch_vscan2(ctx, ws, st, g, inv_a, (const int *)ws.cost.p, (const int *)ws.cspec.p, (const int *)ws.hsum.p,
                                      (int16_t *)ws.raw.p, (int16_t *)ws.mins.p))
                return e < 0 ? e : r3d_fail(ctx, R3D_E_HIP, "k_vscan2 launch failed: %s", hipGetErrorString((hipError_t)e));
        } else if (g.NP == 1) k_vscan<1, CPW><<<grid, 64, 0, st>>>((const int *)ws.cost.p, (const int *)ws.cspec.p, (const int *)ws.hsum.p, g, (int16_t *)ws.raw.p, (int16_t *)ws.mins.p);
        else k_vscan<2, CPW><<<grid, 64, 0, st>>>((const int *)ws.cost.p, (const int *)ws.cspec.p, (const int *)ws.hsum.p, g, (int16_t *)ws.raw.p, (int16_t *)ws.mins.p);
        R3D_HIP(ctx, hipGetLastError());
    }
    }  // impl 1 / 2
    r3d_prof_mark(ctx, ws, st, "lrcheck");
    k_lrcheck<<<h, 256, (size_t)w * 4, st>>>((const int16_t *)ws.raw.p, (const int16_t *)ws.mins.p, g, (int16_t *)(pass == 1 ? ws.lrd2.p : ws.lrd.p));
    R3D_HIP(ctx, hipGetLastError());
    if (pass == 1) return R3D_OK;
    if (tiny) {
        // second run (workspaces are reused: this pass's LR-checked map is complete in ws.lrd, stream-ordered), then the assembly
        const bool prof = ctx->profiling;
        ctx->profiling = false;
        rc = sgm_run_impl(ctx, lane, st, p, d_left, d_right, w, h, stride, d_disp, 1);
        ctx->profiling = prof;
        if (rc) return rc;
        k_tiny_assemble<<<dim3((w + 255) / 256, h), 256, 0, st>>>((int16_t *)ws.lrd.p, (const int16_t *)ws.lrd2.p, g);
        R3D_HIP(ctx, hipGetLastError());
    }
    r3d_prof_mark(ctx, ws, st, "median3");
    k_median3<<<dim3((w + 255) / 256, h), 256, 0, st>>>((const int16_t *)ws.lrd.p, d_disp, w, h);
    R3D_HIP(ctx, hipGetLastError());
    if (p->speckleWindowSize > 0) {
        r3d_prof_mark(ctx, ws, st, "speckles");
        if ((rc = r3d_speckle_run(ctx, ws, st, d_disp, w, h, g.invalid, p->speckleWindowSize, 16 * p->speckleRange))) return rc;
    }
    r3d_prof_end(ctx, ws, st);
    return R3D_OK;
}
