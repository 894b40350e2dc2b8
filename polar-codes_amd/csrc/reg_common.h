// Device helpers shared by the register-resident decoder kernels: DPP lane exchanges and the fp32
// f / g forms that stay on the vector ALU.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "decode_common.h"

// ---- DPP row rotation: lane i of a 16-lane row reads lane (i - R) mod 16 ---------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
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
// the exchange limited to the quads named by BANKS (bit q = lanes 4q .. 4q+3 of every row); other lanes get zero
template <int CTRL, int BANKS> __device__ __forceinline__ uint32_t dpp_banks_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, BANKS, false);
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
#if defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__
    // no NaN to quiet: the compiler folds the absolute values into the minimum's source modifiers itself and,
    // unlike around inline assembly, needs no hazard padding
    return __builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b));
#else
    float m;
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return m;
#endif
}
// Issue cost on gfx950 (scripts/micro/valutest.hip): add / sub / mul / fma, and / or / xor, right shifts and v_bitop3
// with three VGPR sources run at full rate; anything with a scalar-register source, left shifts, min / max, the other
// three-operand integer instructions (v_and_or, v_lshl_or, v_alignbit, v_bfe ...), comparisons and every DPP form at
// half of it.  Hence: constants that feed bit logic sit in VGPRs, (a & b) | c is spelt as v_bitop3, and a left shift
// by one is an addition.
__device__ __forceinline__ uint32_t vgpr_const(uint32_t c)
{
    uint32_t m;
    asm("v_mov_b32 %0, %1" : "=v"(m) : "s"(c));  // not volatile: one copy per kernel is enough
    return m;
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }
__device__ __forceinline__ uint32_t twice(uint32_t x)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));  // x << 1 at the full rate (the compiler would turn x + x into a shift)
    return r;
}
// sign(a) sign(b) min(|a|,|b|) in three vector instructions
__device__ __forceinline__ float f_fast(float a, float b)
{
    const uint32_t s = __float_as_uint(a) ^ __float_as_uint(b);
    return __uint_as_float(and_or(s, vgpr_const(0x80000000u), __float_as_uint(minabs(a, b))));
}
__device__ __forceinline__ double f_fast(double a, double b) { return f_minsum(a, b); }
// The same function in two instructions where both operands are plain registers: the median of (a, -a, b) is
// sign(b) min(|a|, |b|), and one v_bitop3 gives it the sign of a as well.  (min / max / med3 issue at half rate, so this is
// three issue slots against four; with a DPP operand v_min_f32_dpp does the fetch for nothing and f_fast stays.)  Only
// the sign of a zero result can differ from f_fast, which no decision depends on.
__device__ __forceinline__ float f_med(float a, float b)
{
    const float m = __builtin_amdgcn_fmed3f(a, -a, b);
    return __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(a), vgpr_const(0x80000000u), __float_as_uint(m), 0x6A));  // (a & sign) ^ m
}
__device__ __forceinline__ double f_med(double a, double b) { return f_minsum(a, b); }
// b + (bit ? -a : a) with the bit as an integer 0/1
__device__ __forceinline__ float g_fast(float a, float b, uint32_t bit) { return b + __uint_as_float(__float_as_uint(a) ^ (bit << 31)); }
__device__ __forceinline__ double g_fast(double a, double b, uint32_t bit) { return g_apply(a, b, bit); }
// v with its sign inverted when bit (0/1) is set
__device__ __forceinline__ float flip_sign(float v, uint32_t bit) { return __uint_as_float(__float_as_uint(v) ^ (bit << 31)); }
__device__ __forceinline__ double flip_sign(double v, uint32_t bit)
{
    return __hiloint2double(__double2hiint(v) ^ (int)(bit << 31), __double2loint(v));
}

// ---- 16-leaf nodes decoded in registers by one DPP row ---------------------------------------------
// Lane constants of the replicated butterfly: for period P, lanes whose element index is in the upper half.
struct RowConst {
    uint32_t lowmask[4];  // P = 16, 8, 4, 2: all-ones when (sub & P/2) == 0
    uint32_t low31[4];    // the same with bit 31 only
    bool up[4];
};

// Decodes a node of P leaves whose P LLRs are replicated with period P over the row.
// mask: P bits (wave-uniform), bit i = leaf i carries information.  Returns the node's partial-sum bit of
// element (sub mod P) as an integer 0/1.  SHORT enables the rate-0 / rate-1 short-cuts at P = 8 and 4
// (same decisions).
template <int P, bool SHORT, typename T> __device__ __forceinline__ uint32_t sub_node(T v, uint32_t mask, const RowConst &rc)
{
    if constexpr (P == 1) {
        const uint32_t neg = pin(v < (T)0 ? 1u : 0u);
        return neg & mask;  // mask is the single (uniform) info bit
    } else {
        constexpr uint32_t FULL = (1u << P) - 1u, HALF = (1u << (P / 2)) - 1u;
        constexpr int I = P == 16 ? 0 : P == 8 ? 1 : P == 4 ? 2 : 3;
        if (SHORT && P < 16) {
            if (mask == 0u) return 0u;
            if (mask == FULL) return pin(v < (T)0 ? 1u : 0u);
            // only the last leaf carries information: every left child is all-frozen, so the node reduces to
            // the g-chain with zero partial sums, hi + lo level by level, and one decision shared by all leaves
            if (mask == (1u << (P - 1))) {
                T s = v + partner<P>(v);
                if constexpr (P >= 4) s = s + partner<P / 2>(s);
                if constexpr (P >= 8) s = s + partner<P / 4>(s);
                return pin(s < (T)0 ? 1u : 0u);
            }
            if constexpr (P == 4) {
                // leaves 2, 3: the left pair is frozen, the right pair is decided by sign
                if (mask == 0xCu) {
                    const T s = v + partner<4>(v);
                    return pin(s < (T)0 ? 1u : 0u);
                }
                // leaves 1, 2, 3: left pair = one decision on the sum of the two f outputs, right pair by sign
                if (mask == 0xEu) {
                    const T o4 = partner<4>(v);
                    const T vl4 = f_fast(v, o4);
                    const T sl = vl4 + partner<2>(vl4);
                    const uint32_t y4 = pin(sl < (T)0 ? 1u : 0u) & rc.lowmask[I];
                    const T t4 = flip_sign(v, y4);
                    const T vr4 = t4 + partner<4>(t4);
                    const uint32_t xr4 = pin(vr4 < (T)0 ? 1u : 0u);
                    return xr4 ^ y4;
                }
            }
        }
        // g without selecting lo / hi: the lanes of the lower half hold lo and apply the sign themselves, then
        // every lane adds its partner's value (hi + (+-lo) and (+-lo) + hi are the same sum)
        const T o = partner<P>(v);
        const T vl = f_fast(v, o);
        const uint32_t xl = sub_node<P / 2, SHORT>(vl, mask & HALF, rc);
        const uint32_t y = xl & rc.lowmask[I];
        const T t = flip_sign(v, y);
        const T vr = t + partner<P>(t);
        const uint32_t xr = sub_node<P / 2, SHORT>(vr, (mask >> (P / 2)) & HALF, rc);
        return xr ^ y;
    }
}

// The same decoder (fp32) with the rate profile of the node known at compile time: no branch is left.  Decisions
// and partial sums travel in bit 31 of a word, where g needs them (applying a sum to an LLR is one exclusive-or and
// no 0/1 value is shifted into place); the other 31 bits of the returned word are arbitrary.  A decision is the
// LLR itself after adding +0, which turns -0 into +0 and changes no other value: bit 31 is then exactly (v < 0).
// An all-frozen left half needs no f: its sums are zero and the right half sees hi + lo; a node whose last leaf
// alone carries information is the sum of its LLRs.
__device__ __forceinline__ uint32_t decision_word(float v) { return __float_as_uint(v + 0.0f); }
template <int P, uint32_t MASK> __device__ __forceinline__ uint32_t sub_node_s(float v, const RowConst &rc)
{
    constexpr uint32_t FULL = (P >= 32) ? 0xFFFFFFFFu : ((1u << P) - 1u), HALF = (1u << (P / 2)) - 1u;
    constexpr int I = P == 16 ? 0 : P == 8 ? 1 : P == 4 ? 2 : 3;
    if constexpr (MASK == 0u) {
        return 0u;
    } else if constexpr (MASK == FULL) {
        return decision_word(v);
    } else if constexpr (MASK == (1u << (P - 1))) {
        float s = v + partner<P>(v);
        if constexpr (P >= 4) s = s + partner<P / 2>(s);
        if constexpr (P >= 8) s = s + partner<P / 4>(s);
        if constexpr (P >= 16) s = s + partner<P / 8>(s);
        return decision_word(s);
    } else {
        constexpr uint32_t ML = MASK & HALF, MR = (MASK >> (P / 2)) & HALF;
        const float o = partner<P>(v);
        if constexpr (ML == 0u) {
            return sub_node_s<P / 2, MR>(v + o, rc);
        } else {
            const float vl = f_fast(v, o);
            const uint32_t xl = sub_node_s<P / 2, ML>(vl, rc);
            const uint32_t y = xl & rc.low31[I];
            const float t = __uint_as_float(__float_as_uint(v) ^ y);
            const float vr = t + partner<P>(t);
            const uint32_t xr = sub_node_s<P / 2, MR>(vr, rc);
            return xr ^ y;
        }
    }
}

// The 25 rate profiles of a 16-leaf node that respect the partial order of the bit channels (an information
// position implies every position reachable by setting a bit or moving a one towards the MSB); a code built from
// any reliability ordering only has these, all-frozen and all-information.  Leaf i = bit i.
#define POLAR_PROFILES16(X)                                                                                         \
    X(1, 0x8000u) X(2, 0xC000u) X(3, 0xE000u) X(4, 0xE800u) X(5, 0xE880u) X(6, 0xF000u) X(7, 0xF800u) X(8, 0xF880u)   \
    X(9, 0xFC00u) X(10, 0xFC80u) X(11, 0xFCC0u) X(12, 0xFE00u) X(13, 0xFE80u) X(14, 0xFEC0u) X(15, 0xFEE0u)          \
    X(16, 0xFEE8u) X(17, 0xFF00u) X(18, 0xFF80u) X(19, 0xFFC0u) X(20, 0xFFE0u) X(21, 0xFFE8u) X(22, 0xFFF0u)         \
    X(23, 0xFFF8u) X(24, 0xFFFCu) X(25, 0xFFFEu)

// The 16-leaf decoder for the flip search: every leaf is visited (no short-cuts), its decision LLR is recorded
// in dl[pos + leaf] and its decision is inverted when the leaf is one of the (wave-uniform) positions fl0..fl2.
// OFF = index of the node's first leaf inside the 16-leaf block that starts at code position pos.
template <int P, int OFF, typename T>
__device__ __forceinline__ uint32_t sub_node_flip(T v, uint32_t mask, const RowConst &rc, int pos, int fl0, int fl1, int fl2, T *dl,
                                                  bool writer)
{
    if constexpr (P == 1) {
        const int i = pos + OFF;
        uint32_t bit = pin(v < (T)0 ? 1u : 0u) & mask;
        if (i == fl0 || i == fl1 || i == fl2) bit ^= 1u;
        if (writer) dl[i] = v;
        return bit;
    } else {
        constexpr uint32_t HALF = (1u << (P / 2)) - 1u;
        constexpr int I = P == 16 ? 0 : P == 8 ? 1 : P == 4 ? 2 : 3;
        const T o = partner<P>(v);
        const T vl = f_fast(v, o);
        const uint32_t xl = sub_node_flip<P / 2, OFF>(vl, mask & HALF, rc, pos, fl0, fl1, fl2, dl, writer);
        const uint32_t y = xl & rc.lowmask[I];
        const T t = flip_sign(v, y);
        const T vr = t + partner<P>(t);
        const uint32_t xr = sub_node_flip<P / 2, OFF + P / 2>(vr, (mask >> (P / 2)) & HALF, rc, pos, fl0, fl1, fl2, dl, writer);
        return xr ^ y;
    }
}
