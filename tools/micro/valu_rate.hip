// Issue rate of the vector instructions the SGM and registration kernels are made of, per SIMD, at 1 / 2 / 4 waves per SIMD (gfx950); inline
// assembly, 16 independent registers, so nothing is folded away.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/bin/valu_rate && tools/micro/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP>
__global__ void __launch_bounds__(64) k(int *out, int iters, int seed) {
    int a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
    int b = seed | 1;
    double p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = (OP == 20 || (OP >= 31 && OP <= 35) || OP == 45) ? 1.0 + seed * (i + 1) + threadIdx.x : 0.0;
    double pb = 1.0 + 1e-9 * seed;
    const unsigned long long m = 0x5555555555555555ull * (unsigned)seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define ONE(i)                                                                                                          \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    else if (OP == 1) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
    else if (OP == 2) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
    else if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                       \
    else if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));                              \
    else if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));           \
    else if (OP == 6) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                       \
    else if (OP == 7) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));                              \
    else if (OP == 8) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
    else if (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                       \
    else if (OP == 10) asm volatile("v_pk_lshrrev_b16 %0, 2, %0" : "+v"(a[i]));                                         \
    else if (OP == 11) asm volatile("v_min_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)); \
    else if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                             \
    else if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));                                          \
    else if (OP == 14) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                      \
    else if (OP == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 16) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 17) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 18) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                      \
    else if (OP == 19) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                       \
    else if (OP == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pb));                     \
    else if (OP == 21) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));                                               \
    else if (OP == 22) asm volatile("v_bfe_u32 %0, %0, 12, 10" : "+v"(a[i]));                                           \
    else if (OP == 23) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                    \
    else if (OP == 24) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 25) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));                                            \
    else if (OP == 26) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 27) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 28) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                      \
    else if (OP == 29) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                   \
    else if (OP == 30) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                   \
    else if (OP == 31) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pb));                        \
    else if (OP == 32) asm volatile("v_add_f64 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                     \
    else if (OP == 33) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                     \
    else if (OP == 34) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                  \
    else if (OP == 35) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                  \
    else if (OP == 36) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(seed) : "vcc"); \
    else if (OP == 37) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));                      \
    else if (OP == 38) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                       \
    else if (OP == 39) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));                               \
    else if (OP == 40) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                      \
    else if (OP == 41) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 42) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 43) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                      \
    else if (OP == 44) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                      \
    else if (OP == 45) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(p[i]) : "v"(b));                                      \
    else if (OP == 46) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                        \
    else if (OP == 47) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));                                            \
    else if (OP == 48) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));                                            \
    else if (OP == 49) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));                    \
    else if (OP == 50) { if ((i & 1) == 0) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); } \
    else if (OP == 51) { if ((i & 3) != 3) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); } \
    else if (OP == 52) { if ((i & 1) == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            REP16(ONE)
#undef ONE
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i] ^ (int)p[i];
    if (s == 0x12345) out[0] = s;
}
template <int OP>
void run(const char *name) {
    int *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int blocks = 1024 * wps, iters = 2000;
        k<OP><<<blocks, 64>>>(d, 10, 3);
        (void)hipEventRecord(e0);
        k<OP><<<blocks, 64>>>(d, iters, 3);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
}
int main() {
    run<0>("v_add_u32"); run<1>("v_pk_add_i16"); run<2>("v_pk_min_i16"); run<7>("v_pk_sub_u16 clamp"); run<8>("v_pk_max_u16");
    run<10>("v_pk_lshrrev_b16"); run<3>("v_perm_b32"); run<4>("v_alignbit_b32"); run<9>("v_and_b32"); run<6>("v_min_i32");
    run<13>("v_mov_b32"); run<5>("v_mov_b32_dpp"); run<11>("v_min_i32_dpp");
    run<36>("v_cmp+v_cndmask (2)"); run<37>("v_cndmask sgpr mask"); run<14>("v_med3_i32"); run<15>("v_min_u32"); run<27>("v_max_i32"); run<28>("v_min3_i32");
    run<16>("v_min_f32"); run<17>("v_max_f32"); run<18>("v_med3_f32"); run<44>("v_max3_f32"); run<19>("v_fma_f32"); run<20>("v_pk_fma_f32");
    run<34>("v_pk_mul_f32"); run<35>("v_pk_add_f32"); run<21>("v_cvt_f32_u32"); run<22>("v_bfe_u32"); run<23>("v_and_or_b32");
    run<24>("v_sub_f32"); run<43>("v_add_f32"); run<26>("v_mul_f32"); run<25>("v_lshrrev_b32"); run<47>("v_lshlrev_b32"); run<48>("v_ashrrev_i32");
    run<38>("v_or_b32"); run<41>("v_xor_b32"); run<42>("v_sub_u32"); run<39>("v_lshl_or_b32"); run<40>("v_add3_u32");
    run<29>("v_mul_lo_u32"); run<30>("v_mad_u32_u24"); run<46>("v_sad_u8"); run<49>("v_pk_mad_i16");
    run<50>("mix pk_add/add_u32 1:1"); run<51>("mix pk_min x3 / add_u32 x1"); run<52>("mix add_u32/and_b32 1:1");
    run<31>("v_fma_f64"); run<32>("v_add_f64"); run<33>("v_mul_f64"); run<45>("v_cvt_f64_f32");
    return 0;
}
