// Issue rate of the vector instructions the SGM kernels are made of, per SIMD, at 1 / 2 / 4 waves per SIMD (gfx950); inline
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
    else if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(ONE)
#undef ONE
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i];
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
    run<12>("v_cndmask_b32"); run<13>("v_mov_b32"); run<5>("v_mov_b32_dpp"); run<11>("v_min_i32_dpp");
    return 0;
}
