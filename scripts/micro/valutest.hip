// Issue rate of single vector instructions on one SIMD: CHAINS independent dependency chains per wavefront, 16
// instructions per chain and loop trip, written as inline assembly so that the compiler neither fuses nor packs them.
// Prints wave-instructions per nanosecond per SIMD for 4 and 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define OP1(txt) asm volatile(txt : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc")
template <int CHAINS, int KIND> __global__ void __launch_bounds__(64) valutest(float *out, int iters, uint32_t k)
{
    uint32_t a[CHAINS], b[CHAINS], d[CHAINS];
    uint64_t w[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        a[c] = threadIdx.x * 2654435761u + c;
        b[c] = threadIdx.x * 40503u + 7 * c;
        d[c] = threadIdx.x + 13 * c;
        w[c] = ((uint64_t)a[c] << 32) | b[c];
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 4) asm volatile("v_add_f32 %0, %0, -%1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 5) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 6) asm volatile("v_add_f32 %0, %3, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %3, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 9) asm volatile("v_xor_b32 %0, %3, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 10) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 12) asm volatile("v_and_b32 %0, %3, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 13) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 15) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 16) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 18) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 19) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 20) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 21) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 22) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 23) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 24) asm volatile("v_bitop3_b32 %0, %0, %1, 1 bitop3:0x6c" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 25) asm volatile("v_bitop3_b32 %0, %0, %1, %3 bitop3:0x6c" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 26) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 27) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 28) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 29) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 30) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 31) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 33) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 34) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 35) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 36) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 37) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 38) asm volatile("v_min_f32 %0, |%0|, |%1|" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 39) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 40) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 41) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 42) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 43) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 44) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 45) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 46) asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 47) asm volatile("v_xor_b32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 48) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 49) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
                else if (KIND == 50) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(b[c]), "v"(d[c]) : "vcc");
                else if (KIND == 51) asm volatile("v_exp_f32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 52) asm volatile("v_log_f32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 53) asm volatile("v_sin_f32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
                else if (KIND == 54) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[c]) : "v"(b[c]), "v"(d[c]), "s"(k) : "vcc");
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c] + (uint32_t)w[c];
    if (s == 0x12345678u) out[0] = (float)s;
}
template <int CHAINS, int KIND> static void run(const char *name)
{
    float *out;
    (void)hipMalloc(&out, 4);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 8192;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("%-44s chains %d:", name, CHAINS);
    for (int w : {4, 8}) {
        const int grid = cus * 4 * w;
        hipLaunchKernelGGL((valutest<CHAINS, KIND>), dim3(grid), dim3(64), 0, 0, out, 16, 0x80000000u);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((valutest<CHAINS, KIND>), dim3(grid), dim3(64), 0, 0, out, iters, 0x80000000u);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves/SIMD %.3f/ns", w, (double)w * iters * 16 * CHAINS / (ms * 1e6));
    }
    printf("\n");
    (void)hipFree(out);
}
int main()
{
    run<4, 0>("v_add_f32 v,v");
    run<4, 1>("v_sub_f32 v,v");
    run<4, 2>("v_mul_f32 v,v");
    run<4, 3>("v_fma_f32 v,v,v");
    run<4, 4>("v_add_f32 v,-v (VOP3)");
    run<4, 5>("v_add_f32 1.0,v");
    run<4, 6>("v_add_f32 s,v");
    run<4, 7>("v_fma_f32 v,s,v");
    run<4, 8>("v_xor_b32 v,v");
    run<4, 9>("v_xor_b32 s,v");
    run<4, 10>("v_xor_b32 literal,v");
    run<4, 11>("v_and_b32 v,v");
    run<4, 12>("v_and_b32 s,v");
    run<4, 13>("v_or_b32 v,v");
    run<4, 14>("v_add_u32 v,v");
    run<4, 15>("v_add_u32 v,v same register");
    run<4, 16>("v_sub_u32 v,v");
    run<4, 17>("v_mov_b32");
    run<4, 18>("v_lshrrev_b32 3,v");
    run<4, 19>("v_lshrrev_b32 v,v");
    run<4, 20>("v_ashrrev_i32 3,v");
    run<4, 21>("v_lshlrev_b32 1,v");
    run<4, 22>("v_lshlrev_b32 v,v");
    run<4, 23>("v_bitop3_b32 v,v,v");
    run<4, 24>("v_bitop3_b32 v,v,inline 1");
    run<4, 25>("v_bitop3_b32 v,v,s");
    run<4, 26>("v_and_or_b32 v,v,v");
    run<4, 27>("v_or3_b32");
    run<4, 28>("v_xad_u32");
    run<4, 29>("v_add3_u32");
    run<4, 30>("v_bfe_u32 v,v,1");
    run<4, 31>("v_bfi_b32");
    run<4, 32>("v_perm_b32");
    run<4, 33>("v_alignbit_b32 v,v,1");
    run<4, 34>("v_lshl_or_b32");
    run<4, 35>("v_lshl_add_u32");
    run<4, 36>("v_bfrev_b32");
    run<4, 37>("v_min_f32 v,v (VOP2)");
    run<4, 38>("v_min_f32 |v|,|v|");
    run<4, 39>("v_max_f32 v,v (VOP2)");
    run<4, 40>("v_med3_f32");
    run<4, 41>("v_min_u32");
    run<4, 42>("v_cmp_gt_f32 -> vcc");
    run<4, 43>("v_cvt_f32_u32");
    run<4, 44>("v_mul_u32_u24");
    run<4, 45>("v_mul_lo_u32");
    run<4, 46>("v_add_f32_dpp row_ror");
    run<4, 47>("v_xor_b32_dpp quad_perm");
    run<4, 48>("v_mov_b32_dpp quad_perm");
    run<4, 49>("v_pk_add_f32 (two adds)");
    run<4, 50>("v_mad_u64_u32");
    run<4, 51>("v_exp_f32");
    run<4, 52>("v_log_f32");
    run<4, 53>("v_sin_f32");
    run<4, 54>("v_sqrt_f32");
    return 0;
}
