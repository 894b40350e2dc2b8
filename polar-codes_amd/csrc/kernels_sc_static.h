// Register-resident SC decoder state, and the decoder specialised at compile time for one rate profile.
//
// This header is device code only (it needs <hip/hip_runtime.h> and <stdint.h>, nothing from the host side), so
// that the run-time compiler can build it: with POLAR_STATIC_NLOG and POLAR_STATIC_WORDS defined (log2 N and
// the N/32 rate-profile words, bit i of word w = position 32 w + i carries information) it also defines the
// kernel sc_static_kernel, in which every decision about the code tree -- which nodes are all-frozen or
// all-information, the profile of every 16-leaf node, the plane a node's sums go to -- is a compile-time
// constant.  The result is straight-line code: no loop, no branch, no scalar bookkeeping in the tree walk
// (unrolled decoders in the sense of Giard et al.).  Arithmetic and results are those of sc_reg_kernel.
#pragma once
#include "reg_common.h"

typedef uint64_t ps_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int NLOG, int KEEP = 0> struct RegDec {
    static constexpr int S = 1 << (NLOG - 4);  // slice length per lane
    T R[S];                                    // level l (4 <= l < NLOG): 2^(l-4) values at offset 2^(l-4)
    // the first KEEP channel pairs of the lane's slice stay in registers between the top-level f and g, the rest
    // is read from memory a second time
    T CA[KEEP > 0 ? KEEP : 1], CB[KEEP > 0 ? KEEP : 1];
    ps_t ps0, ps1;                             // partial sums (plane 0 / 1), same offsets, one bit per value
    ps_t root;

    // f or g from level LEV to LEV-1 (5 <= LEV <= NLOG); the top level reads the channel from memory
    template <int LEV, bool IS_G> __device__ __forceinline__ void step(const T *chan_lane)
    {
        constexpr int c = 1 << (LEV - 5);
#pragma unroll
        for (int j = 0; j < c; j++) {
            T a, b;
            if constexpr (LEV == NLOG) {
                if (IS_G && j < KEEP) {
                    a = CA[j];
                    b = CB[j];
                } else {
                    a = chan_lane[16 * j];
                    b = chan_lane[16 * (j + c)];
                    if (!IS_G && j < KEEP) {
                        CA[j] = a;
                        CB[j] = b;
                    }
                }
            } else {
                a = R[2 * c + j];
                b = R[3 * c + j];
            }
            if constexpr (IS_G) R[c + j] = g_fast(a, b, (uint32_t)(ps0 >> (c + j)) & 1u);
            else R[c + j] = f_fast(a, b);
        }
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane)
    {
        constexpr int c = 1 << (LEV - 5);
        const ps_t m = (c >= 64) ? ~(ps_t)0 : (((ps_t)1 << c) - 1);
        const ps_t xl = (ps0 >> c) & m, xr = (ps1 >> c) & m;
        if constexpr (LEV == NLOG) {
            root = (xl ^ xr) | (c >= 64 ? 0 : (xr << c));
        } else {
            const ps_t x = (xl ^ xr) | (xr << c);
            const ps_t m2 = (((ps_t)1 << (2 * c)) - 1) << (2 * c);
            if (plane) ps1 = (ps1 & ~m2) | (x << (2 * c));
            else ps0 = (ps0 & ~m2) | (x << (2 * c));
        }
    }
    template <int LEV> __device__ __forceinline__ void rate0(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        const ps_t m2 = (((ps_t)1 << c) - 1) << c;
        if (plane) ps1 &= ~m2;
        else ps0 &= ~m2;
    }
    template <int LEV> __device__ __forceinline__ void rate1(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        ps_t x = 0;
#pragma unroll
        for (int j = 0; j < c; j++) x |= (ps_t)(R[c + j] < (T)0 ? 1u : 0u) << j;
        const ps_t m2 = (((ps_t)1 << c) - 1) << c;
        if (plane) ps1 = (ps1 & ~m2) | (x << c);
        else ps0 = (ps0 & ~m2) | (x << c);
    }
};


// ---- compile-time rate profile --------------------------------------------------------------------------------
template <int NW> struct StaticProfile {
    uint32_t w[NW];
    // information leaves of node idx at level lev (2^lev leaves)
    constexpr int ninfo(int lev, int idx) const
    {
        int c = 0;
        for (int i = 0; i < (1 << lev); i++) {
            const int pos = (idx << lev) + i;
            c += (int)((w[pos >> 5] >> (pos & 31)) & 1u);
        }
        return c;
    }
    constexpr uint32_t mask16(int idx) const { return (w[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu; }
};

// Decoder state of the specialised kernel.  The LLR slice is that of RegDec; the partial sums have one register
// per level and plane, X[l - 4][plane] for a node of level l (4 <= l < NLOG), element j of the lane's slice at bit
// 31 - j: a 16-leaf node delivers its bit in bit 31 (other bits arbitrary), which g applies with one masked
// exclusive-or, and combining two
// children is (xl ^ xr) | (xr >> c) with nothing to clear or to insert, since which register a node writes is
// known when the kernel is compiled.
// Of the lane's 2^(NLOG-5) channel pairs the first KEEP stay in registers between the top-level f and g, the next KL in
// LDS (kept[k * 64], k = 2 pair, 2 pair + 1: the wavefront's region, one column per lane); the rest is read again.
template <int NLOG_, int KEEP = 0, int KL = 0> struct StaticDec {
    static constexpr int NLOG = NLOG_;
    static constexpr int S = 1 << (NLOG - 4);
    float R[S];
    float CA[KEEP > 0 ? KEEP : 1], CB[KEEP > 0 ? KEEP : 1];
    float *kept;
    uint32_t X[NLOG - 4][2];
    ps_t root;  // element j at bit j, as the output stage takes it

    template <int LEV, bool IS_G> __device__ __forceinline__ void step(const float *chan_lane)
    {
        constexpr int c = 1 << (LEV - 5);
        if constexpr (LEV == NLOG && IS_G) {
            // the second read of the channel goes through an offset the compiler cannot see is zero: reusing
            // the values of the first read would keep the whole slice in registers, KEEP decides how much of it is
            uint32_t zero = 0;
            asm volatile("" : "+v"(zero));
            chan_lane += zero;
        }
        auto fetch = [&](int j, float &a, float &b) {
            if constexpr (LEV == NLOG) {
                if (IS_G && j < KEEP) {
                    a = CA[j];
                    b = CB[j];
                } else if (IS_G && j < KEEP + KL) {
                    a = kept[(2 * (j - KEEP)) * 64];
                    b = kept[(2 * (j - KEEP) + 1) * 64];
                } else {
                    a = chan_lane[16 * j];
                    b = chan_lane[16 * (j + c)];
                    if (!IS_G && j < KEEP) {
                        CA[j] = a;
                        CB[j] = b;
                    } else if (!IS_G && j < KEEP + KL) {
                        kept[(2 * (j - KEEP)) * 64] = a;
                        kept[(2 * (j - KEEP) + 1) * 64] = b;
                    }
                }
            } else {
                a = R[2 * c + j];
                b = R[3 * c + j];
            }
        };
        const uint32_t sg = vgpr_const(0x80000000u);
        if constexpr (IS_G && c >= 2) {
            // two elements per addition (packed fp32 add); the sum of element j reaches bit 31 by doubling
            uint32_t t = X[LEV - 5][0];
#pragma unroll
            for (int j = 0; j < c; j += 2) {
                float a0, b0, a1, b1;
                fetch(j, a0, b0);
                fetch(j + 1, a1, b1);
                const uint32_t s0 = t & sg;
                t = twice(t);
                const uint32_t s1 = t & sg;
                t = twice(t);
                const f32x2 r = f32x2{b0, b1} + f32x2{__uint_as_float(__float_as_uint(a0) ^ s0), __uint_as_float(__float_as_uint(a1) ^ s1)};
                R[c + j] = r.x;
                R[c + j + 1] = r.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < c; j++) {
                float a, b;
                fetch(j, a, b);
                if constexpr (IS_G) {
                    const uint32_t sgn = X[LEV - 5][0] & sg;  // c == 1
                    R[c + j] = b + __uint_as_float(__float_as_uint(a) ^ sgn);
                } else {
                    R[c + j] = f_fast(a, b);
                }
            }
        }
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane)
    {
        constexpr int c = 1 << (LEV - 5);  // elements per child
        uint32_t xl = X[LEV - 5][0], xr = X[LEV - 5][1];
        if constexpr (LEV == 5) {  // 16-leaf nodes deliver bit 31 only
            const uint32_t sg = vgpr_const(0x80000000u);
            xl &= sg;
            xr &= sg;
        }
        if constexpr (LEV == NLOG) {
            if constexpr (c == 32) root = (ps_t)__builtin_bitreverse32(xl ^ xr) | ((ps_t)__builtin_bitreverse32(xr) << 32);
            else root = (ps_t)__builtin_bitreverse32((xl ^ xr) | (xr >> c));
        } else {
            X[LEV - 4][plane] = (xl ^ xr) | (xr >> c);
        }
    }
    template <int LEV> __device__ __forceinline__ void rate0(int plane) { X[LEV - 4][plane] = 0u; }
    template <int LEV> __device__ __forceinline__ void rate1(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        if constexpr (c == 1) {
            X[0][plane] = decision_word(R[1]);
        } else {
            // (x << 1) | decision per element, then the c bits moved to the top: element j at bit 31 - j
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < c; j++) x = __builtin_amdgcn_alignbit(x, decision_word(R[c + j]), 31);
            X[LEV - 4][plane] = x << (32 - c);
        }
    }
};

// Node idx of level LEV (>= 5), not all-frozen and not all-information; PLANE = 0 left child, 1 right child.
template <typename P, int LEV, int IDX, int PLANE, typename Dec>
__device__ __forceinline__ void decode_static(Dec &d, const float *chan_lane, const RowConst &rc)
{
    constexpr auto prof = P::value();  // StaticProfile<N/32>
    constexpr int full = 1 << (LEV - 1);
    constexpr int nl = prof.ninfo(LEV - 1, 2 * IDX), nr = prof.ninfo(LEV - 1, 2 * IDX + 1);
    // left child (an all-frozen left child needs no LLRs: its sums are zero)
    if constexpr (nl == 0) {
        d.template rate0<LEV - 1>(0);
    } else {
        d.template step<LEV, false>(chan_lane);
        if constexpr (nl == full) d.template rate1<LEV - 1>(0);
        else if constexpr (LEV - 1 == 4) d.X[0][0] = sub_node_s<16, prof.mask16(2 * IDX)>(d.R[1], rc);
        else decode_static<P, LEV - 1, 2 * IDX, 0>(d, chan_lane, rc);
    }
    // right child
    if constexpr (nr == 0) {
        d.template rate0<LEV - 1>(1);
    } else {
        d.template step<LEV, true>(chan_lane);
        if constexpr (nr == full) d.template rate1<LEV - 1>(1);
        else if constexpr (LEV - 1 == 4) d.X[0][1] = sub_node_s<16, prof.mask16(2 * IDX + 1)>(d.R[1], rc);
        else decode_static<P, LEV - 1, 2 * IDX + 1, 1>(d, chan_lane, rc);
    }
    d.template combine<LEV>(PLANE);
}

// Output stage shared with sc_reg_kernel: u_hat = x_hat F^{(x)n} on the sliced layout, staged in LDS, then lane `sub` of the frame's row assembles output words sub and sub + 16 on its own, one staged bit
// per step, and the words leave as contiguous stores.  Asl is the gather table in LDS, [bit i][word w] with
// GATHER_TW words per row: lanes of a row read neighbouring entries, the four rows the same ones.
constexpr int SC_STAGE_WORDS = 256;  // per wavefront: 32 words per frame, one frame's word every 32 bytes
constexpr int SC_STATIC_WAVES = 4;   // wavefronts per workgroup of the specialised kernel
template <int NLOG> struct GatherShape {
    static constexpr int TW = (NLOG >= 10) ? 32 : 16;  // words per table row (>= 16: every lane has an entry)
    static constexpr int WORDS = 32 * TW;
};
// Error counting fused into the output stage: per-lane running sums of bit errors, frame errors and frames
// (only the first lane of each frame's row counts), added to the global counters once per wavefront.
struct ErrAcc {
    uint32_t be = 0, fe = 0, fr = 0;
};
__device__ __forceinline__ void err_acc_flush(const ErrAcc &a, unsigned long long *counters)
{
    if (a.fr) {
        atomicAdd(&counters[0], (unsigned long long)a.fr);
        if (a.fe) atomicAdd(&counters[1], (unsigned long long)a.fe);
        if (a.be) atomicAdd(&counters[2], (unsigned long long)a.be);
    }
}

// truth (packed [B x Kw], or NULL): the transmitted information words; their first nbits bits are compared with the
// decoded ones and the result goes into acc.  info_hat may then be NULL (nothing is stored).
template <int NLOG>
__device__ __forceinline__ void sc_reg_epilogue(ps_t root, uint32_t *stage, const uint16_t *Asl, uint32_t *__restrict__ info_hat,
                                                int64_t fc, bool active, int lane, int K, int Kw,
                                                const uint32_t *__restrict__ truth = nullptr, int nbits = 0, ErrAcc *acc = nullptr)
{
    const int grp = lane >> 4, sub = lane & 15;
    uint32_t lo = (uint32_t)root, hi = (uint32_t)(root >> 32);
    // stages 4 .. NLOG-1 pair (j, j + 2^(s-4)) inside a lane (hi is zero below N = 1024)
    if constexpr (NLOG > 9) lo ^= hi;
    if constexpr (NLOG > 4) lo ^= (lo >> 1) & 0x55555555u;
    if constexpr (NLOG > 5) lo ^= (lo >> 2) & 0x33333333u;
    if constexpr (NLOG > 6) lo ^= (lo >> 4) & 0x0F0F0F0Fu;
    if constexpr (NLOG > 7) lo ^= (lo >> 8) & 0x00FF00FFu;
    if constexpr (NLOG > 8) lo ^= (lo >> 16) & 0x0000FFFFu;
    if constexpr (NLOG > 9) {
        hi ^= (hi >> 1) & 0x55555555u;
        hi ^= (hi >> 2) & 0x33333333u;
        hi ^= (hi >> 4) & 0x0F0F0F0Fu;
        hi ^= (hi >> 8) & 0x00FF00FFu;
        hi ^= (hi >> 16) & 0x0000FFFFu;
    }
    // stages 0 .. 3 pair lanes (l, l + 2^s) of the row; the lower lane of a pair takes the upper one's word.  Inside
    // a quad the word is masked to the upper lanes before the exchange; across quads the exchange itself is limited
    // to the banks (quads) of the lower lanes.
    {
        const uint32_t odd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, odd2 = (sub & 2) ? 0xFFFFFFFFu : 0u;
        lo ^= dpp_mov_u32<0xB1>(lo & odd1);  // quad_perm:[1,0,3,2]
        lo ^= dpp_mov_u32<0x4E>(lo & odd2);  // quad_perm:[2,3,0,1]
        lo ^= dpp_banks_u32<0x104, 0x5>(lo);  // row_shl:4 into quads 0 and 2
        lo ^= dpp_banks_u32<0x108, 0x3>(lo);  // row_shl:8 into quads 0 and 1
        if constexpr (NLOG > 9) {
            hi ^= dpp_mov_u32<0xB1>(hi & odd1);
            hi ^= dpp_mov_u32<0x4E>(hi & odd2);
            hi ^= dpp_banks_u32<0x104, 0x5>(hi);
            hi ^= dpp_banks_u32<0x108, 0x3>(hi);
        }
        // staged word w (= 2 lane + half) of the frame in row g at byte 32 w + 4 g: a gather address is then the
        // table entry with its low five bits (the bit index) replaced
        stage[16 * sub + grp] = lo;
        if constexpr (NLOG > 9) stage[16 * sub + 8 + grp] = hi;
    }
    wave_sync();
    if (info_hat || truth) {
        constexpr int TW = GatherShape<NLOG>::TW;
        const unsigned char *base = reinterpret_cast<const unsigned char *>(stage);
        const uint32_t goff = (uint32_t)grp << 2, wordmask = vgpr_const(~31u);
        auto gather = [&](int w) {
            const uint16_t *t = Asl + w;
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const uint32_t e = t[i * TW];  // (staged word << 5) | bit
                const uint32_t word = *reinterpret_cast<const uint32_t *>(base + and_or(e, wordmask, goff));
                acc = __builtin_amdgcn_alignbit(word >> (e & 31u), acc, 1);  // bit i of acc after 32 steps
            }
            const int valid = K - 32 * w;  // unused table entries point at bit 0 of the slice
            return acc & (valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u)));
        };
        const uint32_t w_lo = gather(sub);
        const uint32_t w_hi = (TW > 16 && Kw > 16) ? gather(sub + 16) : 0u;
        if (info_hat && active) {
            uint32_t *dst = info_hat + fc * Kw;
            if (sub < Kw) dst[sub] = w_lo;
            if (sub + 16 < Kw) dst[sub + 16] = w_hi;
            if (K == 0 && sub == 0) dst[0] = 0u;
        }
        if (truth) {
            const uint32_t *tw = truth + fc * Kw;
            uint32_t e = 0;
            if (sub < Kw) {
                const int valid = nbits - 32 * sub;
                const uint32_t m = valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
                e += __popc((w_lo ^ tw[sub]) & m);
            }
            if (sub + 16 < Kw) {
                const int valid = nbits - 32 * (sub + 16);
                const uint32_t m = valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
                e += __popc((w_hi ^ tw[sub + 16]) & m);
            }
            // sum over the 16 lanes of the frame's row
            e += dpp_mov_u32<0x128>(e);
            e += dpp_mov_u32<0x124>(e);
            e += dpp_mov_u32<0x122>(e);
            e += dpp_mov_u32<0x121>(e);
            if (active && sub == 0) {
                acc->be += e;
                acc->fe += e ? 1u : 0u;
                acc->fr += 1u;
            }
        }
    }
    wave_sync();
}

#if defined(POLAR_STATIC_NLOG) && defined(POLAR_STATIC_WORDS)
struct PolarStaticProfile {
    static constexpr StaticProfile<((1 << POLAR_STATIC_NLOG) + 31) / 32> value()
    {
        return StaticProfile<((1 << POLAR_STATIC_NLOG) + 31) / 32>{{POLAR_STATIC_WORDS}};
    }
};

// Channel pairs per lane that stay on chip between the top-level f and g: POLAR_STATIC_KEEP in registers (16 is what
// fits beside the decoder state at four wavefronts per SIMD) and up to POLAR_STATIC_KEEP_LDS more in LDS; what is
// left is read a second time, mostly from the L2 / Infinity Cache.
#ifndef POLAR_STATIC_KEEP
#define POLAR_STATIC_KEEP 16
#endif
#ifndef POLAR_STATIC_KEEP_LDS
#define POLAR_STATIC_KEEP_LDS 16
#endif
#ifndef POLAR_STATIC_WAVES
#define POLAR_STATIC_WAVES 4
#endif
struct PolarStaticShape {
    static constexpr int NLOG = POLAR_STATIC_NLOG;
    static constexpr int PAIRS = 1 << (NLOG - 5);
    static constexpr int KEEP = POLAR_STATIC_KEEP < PAIRS ? POLAR_STATIC_KEEP : PAIRS;
    static constexpr int KL = (PAIRS - KEEP) < POLAR_STATIC_KEEP_LDS ? (PAIRS - KEEP) : POLAR_STATIC_KEEP_LDS;
    static constexpr int KEPT_WORDS = 2 * KL * 64;  // per wavefront
};
// Workgroups of SC_STATIC_WAVES wavefronts share the gather table; everything else is per wavefront (the wavefronts
// of a workgroup never wait for each other after the table is loaded).
extern "C" __global__ void __launch_bounds__(64 * SC_STATIC_WAVES, POLAR_STATIC_WAVES) sc_static_kernel(
    const float *__restrict__ llr, const uint16_t *__restrict__ Asl, uint32_t *__restrict__ info_hat, long long B, int K, int Kw,
    const uint32_t *__restrict__ truth, int nbits, unsigned long long *__restrict__ counters)
{
    constexpr int NLOG = POLAR_STATIC_NLOG;
    constexpr int N = 1 << NLOG;
    constexpr int KEPT = PolarStaticShape::KEPT_WORDS;
    __shared__ __attribute__((aligned(32))) uint32_t stage_s[SC_STATIC_WAVES * SC_STAGE_WORDS];
    __shared__ uint16_t asl_s[GatherShape<NLOG>::WORDS];
    __shared__ float kept_s[SC_STATIC_WAVES * (KEPT > 0 ? KEPT : 1)];
    for (int k = threadIdx.x; k < GatherShape<NLOG>::WORDS; k += 64 * SC_STATIC_WAVES) asl_s[k] = Asl[k];
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    uint32_t *stage = stage_s + wave * SC_STAGE_WORDS;
    ErrAcc acc;
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (sub & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }
    StaticDec<NLOG, PolarStaticShape::KEEP, PolarStaticShape::KL> d;
    d.kept = kept_s + wave * KEPT + lane;
    const long long stride = (long long)gridDim.x * (4 * SC_STATIC_WAVES);
    for (long long f0 = ((long long)blockIdx.x * SC_STATIC_WAVES + wave) * 4; f0 < B; f0 += stride) {
        const long long f = f0 + grp;
        const bool active = f < B;
        const long long fc = active ? f : B - 1;
        decode_static<PolarStaticProfile, NLOG, 0, 0>(d, llr + fc * N + sub, rc);
        sc_reg_epilogue<NLOG>(d.root, stage, asl_s, info_hat, fc, active, lane, K, Kw, truth, nbits, &acc);
    }
    if (truth) err_acc_flush(acc, counters);
}
#endif
