// Philox4x32-10 counter-based generator and Box-Muller transform (device side).
//   key     = (seed_lo, seed_hi)
//   counter = (frame_lo, frame_hi, block, stream)     stream 0 = info bits, 1 = channel noise
// so every frame's inputs depend only on (seed, global frame index): results do not change with the
// batch size, the chunking or the number of GPUs the frames are spread over.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

__device__ __forceinline__ uint32_t philox_vgpr(uint32_t c)
{
    uint32_t m;
    asm("v_mov_b32 %0, %1" : "=v"(m) : "s"(c));  // not volatile: one copy per kernel
    return m;
}
// On gfx950 a vector instruction with a scalar-register source issues at half rate, so the multipliers and the key
// schedule sit in vector registers, and each of the two three-way exclusive-ors of a round is one v_bitop3.
__device__ __forceinline__ void philox4x32_10(uint64_t frame, uint32_t block, uint32_t stream, uint64_t seed, uint32_t out[4])
{
    uint32_t c0 = (uint32_t)frame, c1 = (uint32_t)(frame >> 32), c2 = block, c3 = stream;
    uint32_t k0 = philox_vgpr((uint32_t)seed), k1 = philox_vgpr((uint32_t)(seed >> 32));
    const uint32_t M0 = philox_vgpr(0xD2511F53u), M1 = philox_vgpr(0xCD9E8D57u);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 multiply-add per product (v_mad_u64_u32) instead of a low and a high multiply
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// Two standard normals from two 32-bit uniforms: u1 = (a + 0.5) 2^-32 in (0,1], u2 = b 2^-32 in [0,1).
template <typename T> __device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, T &z0, T &z1);

template <> __device__ __forceinline__ void box_muller<float>(uint32_t a, uint32_t b, float &z0, float &z1)
{
    float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;
    float u2 = (float)b * 2.3283064365386963e-10f;
    // hardware log2 / sqrt / sin / cos (the last two take their argument in turns, and u2 is one): about 1e-6
    // absolute on z, a tenth of what the fp32 channel test allows
    float R = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
    z0 = R * __builtin_amdgcn_cosf(u2);
    z1 = R * __builtin_amdgcn_sinf(u2);
}

#ifndef __HIPCC_RTC__  // the run-time compiled kernels are fp32
template <> __device__ __forceinline__ void box_muller<double>(uint32_t a, uint32_t b, double &z0, double &z1)
{
    double u1 = ((double)a + 0.5) * 2.3283064365386963e-10;
    double u2 = (double)b * 2.3283064365386963e-10;
    double R = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = R * c;
    z1 = R * s;
}
#endif
