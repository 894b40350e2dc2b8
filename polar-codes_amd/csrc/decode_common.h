// Device helpers shared by the SC and SCL decoder kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // the run-time compiler brings its own HIP declarations but no <stdint.h>
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
#endif

__device__ __forceinline__ void wave_sync()
{
    // Lanes of one wavefront exchange data through LDS; LDS operations of a wavefront complete in
    // order, so only the compiler has to be kept from moving accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// min-sum f: sign(a) sign(b) min(|a|,|b|)
__device__ __forceinline__ float f_minsum(float a, float b)
{
    float m = fminf(fabsf(a), fabsf(b));
    uint32_t s = (__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u;
    return __uint_as_float(__float_as_uint(m) | s);
}
__device__ __forceinline__ double f_minsum(double a, double b)
{
    double m = fmin(fabs(a), fabs(b));
    uint64_t s = ((uint64_t)__double_as_longlong(a) ^ (uint64_t)__double_as_longlong(b)) & 0x8000000000000000ull;
    return __longlong_as_double((long long)((uint64_t)__double_as_longlong(m) | s));
}

template <int G> __device__ __forceinline__ uint64_t group_ballot(bool pred, int grp)
{
    uint64_t b = __ballot(pred);
    if constexpr (G == 64) return b;
    return (b >> (grp * G)) & ((1ull << G) - 1ull);
}

// Packs pred(k), k in [0, count), into 32-bit words and hands them to store(word_index, value).
// Every lane of the group sees the same packed words.
template <int G, typename Pred, typename Store>
__device__ __forceinline__ void pack_bits_group(int count, int sub, int grp, Pred pred, Store store)
{
    uint32_t acc = 0;
#pragma unroll 4  // lets the table and bit reads of neighbouring iterations overlap
    for (int k0 = 0; k0 < count; k0 += G) {
        int k = k0 + sub;
        bool p = (k < count) && pred(k);
        uint64_t gb = group_ballot<G>(p, grp);
        if constexpr (G >= 32) {
            store(k0 >> 5, (uint32_t)gb);
            if (G == 64 && k0 + 32 < count) store((k0 >> 5) + 1, (uint32_t)(gb >> 32));
        } else {
            acc |= (uint32_t)gb << (k0 & 31);
            if (((k0 + G) & 31) == 0 || k0 + G >= count) {
                store(k0 >> 5, acc);
                acc = 0;
            }
        }
    }
}

__device__ __forceinline__ uint32_t butterfly_word(uint32_t x, int n)
{
    if (n > 0) x ^= (x >> 1) & 0x55555555u;
    if (n > 1) x ^= (x >> 2) & 0x33333333u;
    if (n > 2) x ^= (x >> 4) & 0x0F0F0F0Fu;
    if (n > 3) x ^= (x >> 8) & 0x00FF00FFu;
    if (n > 4) x ^= (x >> 16) & 0x0000FFFFu;
    return x;
}


// ---- list decoders: information leaves every path is sure of --------------------------------------------------------
// The branch metric of deciding u at decision LLR v is 1 - log2(1 + 2^e), e = (u ? v : -v) / ln 2 (cn/polarCode.py:982-988).
// sure_exponent gives a lower bound of |e| for the price of one multiplication and says whether |e| is so large that the
// extension agreeing with the sign of v adds exactly 1 to its path's metric in the metric's own arithmetic: 2^-|e| is then
// below half a unit in the last place of 1, the sum is exactly 1 and its logarithm exactly 0.  The other extension then adds
// 1 - |e| up to rounding, at most 2 - |e| (lower bound of |e|) with a margin no rounding reaches, and since rounding is
// monotone the bound holds for the rounded sums as well.  A list decoder uses this on a full list: if in every path |e| is
// that large, the agreeing extensions pm + 1 are in strictly falling order by list position, and every other extension's
// bound pm + (2 - |e|) lies strictly below the last of them, then the stable ranking of the 2L candidates puts each path's
// agreeing extension at the path's own position: no path moves, every metric grows by exactly 1, and the leaf is decided
// without evaluating a metric or ranking anything -- the same bits and metrics, not an approximation.
__device__ __forceinline__ bool sure_exponent(float v, float &ae_lo)
{
    ae_lo = fabsf(v * 1.44269504088896340736f);  // the fp32 metric's own |e|
    return ae_lo >= 25.0f;
}
__device__ __forceinline__ bool sure_exponent(double v, double &ae_lo)
{
    const double a = fabs(v);
    ae_lo = a * 1.4426950408889;  // the fp64 metric divides by ln 2: this constant is below 1 / ln 2 by more than any rounding
    return a >= 37.5;             // |v| / ln 2 >= 54.1
}
