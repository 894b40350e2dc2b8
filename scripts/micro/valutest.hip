// Vector-ALU issue rate of one CU as a function of the wavefronts per SIMD and of how many independent instruction
// chains each wavefront carries.  Prints wave-instructions per cycle per SIMD (clock from wall_clock64 is not the
// shader clock, so the rate is also given per nanosecond).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int CHAINS, int KIND> __global__ void __launch_bounds__(64) valutest(float *out, int iters)
{
    float a[CHAINS];
    uint32_t u[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        a[c] = (float)threadIdx.x * 0.001f + (float)c;
        u[c] = threadIdx.x * 2654435761u + c;
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = __builtin_fmaf(a[c], 1.0001f, 0.5f);                     // v_fma_f32
                else if (KIND == 1) u[c] = (u[c] ^ 0x9E3779B9u) + (u[c] >> 3);                    // xor / shift-add: 2 ops
                else a[c] = __builtin_fminf(__builtin_fabsf(a[c]), 3.0f) + 1.0f;                   // min + add: 2 ops
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c] + (float)u[c];
    if (s == 12345.678f) out[0] = s;
}
template <int CHAINS, int KIND> static void run(int waves_per_simd, int ops_per_step)
{
    float *out;
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = cus * 4 * waves_per_simd;
    hipLaunchKernelGGL((valutest<CHAINS, KIND>), dim3(grid), dim3(64), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL((valutest<CHAINS, KIND>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * 16 * CHAINS * ops_per_step;
    printf("kind %d chains %d waves/SIMD %d: %.3f ms, %.3f wave-instructions per ns per SIMD (at 2.4 GHz: %.3f per cycle)\n", KIND, CHAINS,
           waves_per_simd, ms, instr_per_simd / (ms * 1e6), instr_per_simd / (ms * 1e6) / 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) run<1, 0>(w, 1);
    for (int w : {1, 2, 4, 8}) run<4, 0>(w, 1);
    for (int w : {1, 2, 4, 8}) run<4, 1>(w, 2);
    for (int w : {1, 2, 4, 8}) run<4, 2>(w, 2);
    return 0;
}
