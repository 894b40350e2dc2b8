// Device helpers shared by the register-resident decoder kernels: DPP lane exchanges and the fp32
// f / g forms that stay on the vector ALU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_common.h"

// ---- DPP row rotation: lane i of a 16-lane row reads lane (i - R) mod 16 ---------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    uint64_t b = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// partner exchange for a period-P replicated vector: rotate by P/2
template <int P, typename T> __device__ __forceinline__ T partner(T v)
{
    if constexpr (P == 16) return dpp_mov<0x128>(v);       // row_ror:8
    else if constexpr (P == 8) return dpp_mov<0x124>(v);   // row_ror:4
    else if constexpr (P == 4) return dpp_mov<0x4E>(v);    // quad_perm:[2,3,0,1]
    else return dpp_mov<0xB1>(v);                          // quad_perm:[1,0,3,2]
}

template <typename T> __device__ __forceinline__ T g_apply(T a, T b, uint32_t bit) { return bit ? b - a : b + a; }

// ---- fp32 helpers that keep the bottom-subtree bit logic on the vector ALU -------------------------
// (the scalar ALU is shared by the four SIMDs of a CU; boolean algebra on lane masks would land there)
__device__ __forceinline__ uint32_t pin(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float minabs(float a, float b)
{
    float m;
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
// sign(a) sign(b) min(|a|,|b|) in three vector instructions
__device__ __forceinline__ float f_fast(float a, float b)
{
    uint32_t s = __float_as_uint(a) ^ __float_as_uint(b);
    return __uint_as_float((s & 0x80000000u) | __float_as_uint(minabs(a, b)));
}
__device__ __forceinline__ double f_fast(double a, double b) { return f_minsum(a, b); }
// b + (bit ? -a : a) with the bit as an integer 0/1
__device__ __forceinline__ float g_fast(float a, float b, uint32_t bit) { return b + __uint_as_float(__float_as_uint(a) ^ (bit << 31)); }
__device__ __forceinline__ double g_fast(double a, double b, uint32_t bit) { return g_apply(a, b, bit); }

