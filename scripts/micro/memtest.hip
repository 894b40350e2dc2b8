// Ceiling of the SC kernel's access pattern: per frame 1024 floats, lane `sub` of a 16-lane group reads
// elements sub + 16 j; PASSES reads of the whole frame per decode; no other work.  Prints frames/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int PASSES, int WIDE> __global__ void __launch_bounds__(64, 5) memtest(const float *__restrict__ llr, float *out, long long B)
{
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    float acc = 0.f;
    for (long long f0 = (long long)blockIdx.x * 4; f0 < B; f0 += (long long)gridDim.x * 4) {
        const long long f = f0 + grp < B ? f0 + grp : B - 1;
#pragma unroll
        for (int p = 0; p < PASSES; p++) {
            if (WIDE) {
                const float4 *c4 = reinterpret_cast<const float4 *>(llr + f * 1024) + sub;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    float4 v = c4[16 * j];
                    acc += v.x + v.y + v.z + v.w;
                }
            } else {
                const float *c = llr + f * 1024 + sub;
#pragma unroll
                for (int j = 0; j < 64; j++) acc += c[16 * j];
            }
            acc = acc * 0.5f + (float)p;
            asm volatile("" ::: "memory");  // the next pass must load again
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main()
{
    const long long B = 10000000;
    float *llr, *out;
    hipMalloc(&llr, B * 1024 * 4);
    hipMalloc(&out, 4);
    hipMemset(llr, 0, B * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int variant = -1; variant < 4; variant++) {
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            const int grid = 256 * 20 * 8;
            if (variant <= 0) hipLaunchKernelGGL((memtest<1, 0>), dim3(grid), dim3(64), 0, 0, llr, out, B);
            if (variant == 1) hipLaunchKernelGGL((memtest<2, 0>), dim3(grid), dim3(64), 0, 0, llr, out, B);
            if (variant == 2) hipLaunchKernelGGL((memtest<1, 1>), dim3(grid), dim3(64), 0, 0, llr, out, B);
            if (variant == 3) hipLaunchKernelGGL((memtest<2, 1>), dim3(grid), dim3(64), 0, 0, llr, out, B);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 3 && variant >= 0) printf("passes %d wide %d: %.2f ms, %.0f M frames/s, %.2f TB/s requested\n", 1 + (variant & 1), variant >> 1, ms, B / ms / 1e3,
                            (1 + (variant & 1)) * B * 4096.0 / ms / 1e9);
        }
    }
    return 0;
}
