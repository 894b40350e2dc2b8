// SC decoder for codes longer than the register kernels hold (2048 <= N <= 65536, fp32, information bits).
//
// The same mapping as sc_reg_kernel -- sixteen lanes per frame, four frames per wavefront, element k of every
// tree level in lane k mod 16.  Where a level lives depends on its width:
//   * levels of 4096 elements and more (below the channel) in a slab of global memory per frame (level l at offset
//     2^l - 4096) with their partial sums beside them as per-lane bit slices.  A lane only ever touches the elements
//     of its own residue, so the slab needs no synchronisation at all, and the order of a level inside the slab is
//     free: element 16 j + s sits at index 64 (j / 4) + 4 s + j % 4, four elements of a lane side by side, so that a
//     step moves 16 bytes per lane and 256 contiguous bytes per frame and instruction;
//   * level 11 (2048 elements) on chip: its first half in 64 registers per lane, its second half in LDS (16 KB per
//     wavefront);
//   * level 10, the LLRs of one 1024-leaf subtree, nowhere: the two top-level steps of the subtree walk recompute
//     them from level 11 (one f or g each) when they need them;
//   * everything below in the registers of the subtree walk of sc_reg_kernel (kernels_sc_node.h).
// A step above the subtrees (f or g from level l to l-1) also does the f steps that would follow it down the left
// spine while its results are still in registers (up to three levels at once), so a level of the slab is written
// once and read once, by the g step of its node.  HBM traffic per frame at N = 16384: the channel twice, levels 13
// and 12 written and read once each, 428 KB measured (1041 KB with every level stored and every step on its own).
// All-frozen and all-information nodes of 1024 leaves and more are short-cut as in the other kernels.  The output
// stage transforms the sliced root (x_hat -> u_hat), writes it to the slab and gathers the information bits
// through a table.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "decode_common.h"
#include "reg_common.h"
#include "kernels_sc_node.h"
#include "polar_internal.h"

struct BigTables {
    const uint32_t *maskw;        // N / 32 rate-profile words
    const uint32_t *pid16;        // N / 16 profile numbers of the 16-leaf nodes
    const NodeClasses *sub;       // N / 1024: node classes inside each 1024-leaf subtree
    const uint64_t *up_r0;        // [l - 10], l = 10 .. n-1: bit i = node i of level l is all-frozen
    const uint64_t *up_r1;        //                                        ... all-information
    const uint32_t *gather;       // [32][KwPad]: information bit 32 w + i sits in u_hat word (entry >> 5) of the slab, bit (entry & 31)
    int KwPad;
};

// Slab of one frame, in 32-bit words: the LLR levels that are stored (12 .. n-1; none for N = 2048), then the partial-sum
// planes.  Partial sums of level l (10 <= l <= n): 2^(l-9) words per lane, word w of lane s at [w * 16 + s]; two planes.
template <int NLOG> struct BigLayout {
    static constexpr int N = 1 << NLOG;
    static constexpr int LLR_BASE = NLOG > 11 ? 4096 : N;  // levels below 12 live on chip
    static constexpr int LLR_WORDS = N - LLR_BASE;
    static __device__ __host__ constexpr int llr_off(int l) { return (1 << l) - LLR_BASE; }
    static __device__ __host__ constexpr int ps_words(int l) { return 16 << (l - 9); }           // one plane of level l
    static __device__ __host__ constexpr int ps_off(int l, int plane)                            // levels 10 .. n
    {
        return LLR_WORDS + 2 * ((16 << (l - 9)) - 32) + plane * (16 << (l - 9));                  // 2 * sum_{k<l} ps_words(k)
    }
    static constexpr int FRAME_WORDS = LLR_WORDS + 2 * ((16 << (NLOG + 1 - 9)) - 32);
};

// Decoder of one 1024-leaf subtree.  CHIP (codes longer than 2048): the subtree's own LLRs are stored nowhere.  The
// level-11 node above it stays on chip instead -- its first half in Ch (element 16 j + lane at Ch[j]), its second half
// in LDS (hi[64 j], one column per lane) -- and the subtree is that node's left child, LLR j = f(Ch[j], hi[64 j]), or
// its right child, g of the same pair with the left child's partial sums `lbits`; both top-level steps of the subtree
// recompute their operands from there.  Not CHIP (N = 2048): Ch holds the subtree's LLRs themselves.
// Pointers that went through an opaque copy keep their address space in the type, so the accesses stay global_* / ds_*
// instructions (a generic pointer would make them flat_*, which wait on both memory counters).
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(1))) float glb_float;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4f glb_v4f;
template <bool CHIP> struct BigDec : RegDec<float, 10, 0, false, true> {
    typedef RegDec<float, 10, 0, false, true> Base;
    lds_float *hi;
    uint64_t lbits;
    bool right;
    // The operands of one pass over the subtree's LLRs.  They go through an opaque copy on purpose: the pass sits inside
    // the subtree walk's loops, and loop-invariant code motion would otherwise compute all 64 LLRs (and 64 shifted
    // partial-sum bits) ahead of the loops and keep them in registers, which is exactly what this layout avoids.
    struct Operands {
        const lds_float *hi;
        uint32_t lo, up;
    };
    __device__ __forceinline__ Operands operands() const
    {
        Operands o = {hi, (uint32_t)lbits, (uint32_t)(lbits >> 32)};
        asm volatile("" : "+v"(o.hi), "+v"(o.lo), "+v"(o.up));
        return o;
    }
    template <bool RIGHT> __device__ __forceinline__ float ch(const Operands &o, int j) const
    {
        const float a = Ch[j], b = o.hi[64 * j];
        if constexpr (RIGHT) return g_fast(a, b, ((j < 32 ? o.lo : o.up) >> (j & 31)) & 1u);
        else return f_med(a, b);
    }
    template <bool IS_G, bool RIGHT> __device__ __forceinline__ void top()
    {
        const Operands o = operands();
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const float x = ch<RIGHT>(o, j), y = ch<RIGHT>(o, j + 32);
            if constexpr (IS_G) R[32 + j] = g_fast(x, y, (uint32_t)(ps0 >> (32 + j)) & 1u);
            else R[32 + j] = f_med(x, y);
        }
    }
    template <int LEV, bool IS_G> __device__ __forceinline__ void step(const float *chan_lane)
    {
        if constexpr (CHIP && LEV == 10) {
            if (right) top<IS_G, true>();
            else top<IS_G, false>();
        } else {
            Base::template step<LEV, IS_G>(chan_lane);
        }
    }
};

// Channel and slab levels are streams: every value is read once per pass and, in the slab, long after it was written.
#ifdef POLAR_BIG_NT
#define BIG_LD(p) __builtin_nontemporal_load(p)
#define BIG_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define BIG_LD(p) (*(p))
#define BIG_ST(v, p) (*(p) = (v))
#endif

template <int NLOG> struct BigCtx {
    static constexpr bool CHIP = NLOG > 11;
    static constexpr int FLOOR = CHIP ? 11 : 10;  // the lowest level a step above the subtrees writes
    BigDec<CHIP> d;
    RowConst rc;
    BigTables t;
    float *slab;          // this frame's slab
    const float *chan;    // this frame's channel LLRs
    int sub;
};

// the lane's index / the frame's slab pointer through an opaque copy (see big_chain)
__device__ __forceinline__ int opaque_idx(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ glb_float *opaque_ptr(float *p)
{
    glb_float *g = (glb_float *)p;
    asm volatile("" : "+v"(g));
    return g;
}

// f of four pairs.  Measured per length with both forms of f (M frames/s, f_med / f_fast): 8192: 27.3 / 22.7, 16384: 10.1 / 8.8,
// 32768: 3.68 / 3.36, 65536: 1.49 / 1.38, but 4096: 48.4 / 58.1 (there the only such step reads the channel).
template <int NLOG> __device__ __forceinline__ float4 f_fast4(const float4 &a, const float4 &b)
{
    if constexpr (NLOG == 12) return make_float4(f_fast(a.x, b.x), f_fast(a.y, b.y), f_fast(a.z, b.z), f_fast(a.w, b.w));
    else return make_float4(f_med(a.x, b.x), f_med(a.y, b.y), f_med(a.z, b.z), f_med(a.w, b.w));
}

// f / g from level LEV (the channel, or a level >= 12 of the slab) to LEV-1: elements sub + 16 j of the lane's residue,
// four (j = 4 J .. 4 J + 3) per trip.  The channel itself is in natural order (4-byte accesses); every level of the
// slab is in the order described above.
template <int LEV, int NLOG, bool IS_G> __device__ __forceinline__ float4 big_trip(BigCtx<NLOG> &cx, const glb_float *slab, const glb_u32 *bits, int J, int sq)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int half = 1 << (LEV - 1);
    float4 a, b;
    if constexpr (LEV == NLOG) {
        const float *in = cx.chan + 64 * J + sq;
        a = make_float4(BIG_LD(in), BIG_LD(in + 16), BIG_LD(in + 32), BIG_LD(in + 48));
        b = make_float4(BIG_LD(in + half), BIG_LD(in + half + 16), BIG_LD(in + half + 32), BIG_LD(in + half + 48));
    } else {
        static_assert(LEV >= 12, "levels below 12 are not read from the slab");
        const glb_float *in = slab + Lay::llr_off(LEV) + 64 * J + 4 * sq;
        const v4f va = BIG_LD((const glb_v4f *)in), vb = BIG_LD((const glb_v4f *)(in + half));
        a = make_float4(va.x, va.y, va.z, va.w);
        b = make_float4(vb.x, vb.y, vb.z, vb.w);
    }
    if (IS_G) {
        const uint32_t w = bits[(J >> 3) * 16] >> ((4 * J) & 31);
        return make_float4(g_fast(a.x, b.x, w & 1u), g_fast(a.y, b.y, (w >> 1) & 1u), g_fast(a.z, b.z, (w >> 2) & 1u),
                           g_fast(a.w, b.w, (w >> 3) & 1u));
    }
    return f_fast4<NLOG>(a, b);
}
// trip J of level L goes to its home: the slab (L >= 12), the on-chip level 11 (registers for the first half, LDS for
// the second) or, for N = 2048, the subtree's LLR registers.  For L <= 11 J must be a compile-time constant after unrolling.
template <int L, int NLOG> __device__ __forceinline__ void big_put(BigCtx<NLOG> &cx, glb_float *slab, int J, const float4 &r, int sq)
{
    typedef BigLayout<NLOG> Lay;
    if constexpr (L == 10) {
        static_assert(!BigCtx<NLOG>::CHIP, "level 10 is stored for N = 2048 only");
        cx.d.Ch[4 * J] = r.x;
        cx.d.Ch[4 * J + 1] = r.y;
        cx.d.Ch[4 * J + 2] = r.z;
        cx.d.Ch[4 * J + 3] = r.w;
    } else if constexpr (L == 11) {
        if (J < 16) {
            cx.d.Ch[4 * J] = r.x;
            cx.d.Ch[4 * J + 1] = r.y;
            cx.d.Ch[4 * J + 2] = r.z;
            cx.d.Ch[4 * J + 3] = r.w;
        } else {
            lds_float *h = cx.d.hi + 256 * (J - 16);
            h[0] = r.x;
            h[64] = r.y;
            h[128] = r.z;
            h[192] = r.w;
        }
    } else {
        v4f v = {r.x, r.y, r.z, r.w};
        BIG_ST(v, (glb_v4f *)(slab + Lay::llr_off(L) + 4 * sq + 64 * J));
    }
}
// the f steps that follow at once: level L from pairs (i, i + CNT) of the trips of level L + 1 held in v, and so on down
template <int L, int NLOG, int CNT, int TM, int W> __device__ __forceinline__ void big_descend(BigCtx<NLOG> &cx, glb_float *slab, float4 (&v)[W], int t, int sq)
{
    if constexpr (CNT >= 1) {
#pragma unroll
        for (int i = 0; i < CNT; i++) {
            v[i] = f_fast4<NLOG>(v[i], v[i + CNT]);
            big_put<L, NLOG>(cx, slab, t + i * TM, v[i], sq);
        }
        big_descend<L - 1, NLOG, CNT / 2, TM, W>(cx, slab, v, t, sq);
    }
}
// The step from level LEV to LEV-1 (f, or g with the left sibling's partial sums) together with the D-1 f steps down
// the left spine that would follow it at once: levels LEV-1 .. LEV-D are written, and none of them is read back for it
// (trip t of the lowest level needs trips t + i TM of the levels above, which are still in registers).
template <int LEV, int NLOG, bool IS_G, int D> __device__ __forceinline__ void big_chain(BigCtx<NLOG> &cx)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int M = LEV - D;          // lowest level written
    constexpr int TM = (1 << M) / 64;   // its trips: four elements per lane and trip
    constexpr int W = 1 << (D - 1);     // trips of level LEV-1 per trip of level M
    static_assert(M >= BigCtx<NLOG>::FLOOR && D >= 1 && D <= 3, "chain depth");
    // the lane's index goes through an opaque copy: addresses derived from it are formed here, base plus instruction
    // offset, instead of being hoisted out of the frame loop as one register pair per access
    int sq = cx.sub;
    asm volatile("" : "+v"(sq));
    glb_float *slab = opaque_ptr(cx.slab);
    const glb_u32 *bits = (const glb_u32 *)slab + Lay::ps_off(LEV - 1, 0) + sq;
    auto body = [&](int t) {
        float4 v[W];
#pragma unroll
        for (int i = 0; i < W; i++) v[i] = big_trip<LEV, NLOG, IS_G>(cx, slab, bits, t + i * TM, sq);
#pragma unroll
        for (int i = 0; i < W; i++) big_put<LEV - 1, NLOG>(cx, slab, t + i * TM, v[i], sq);
        big_descend<LEV - 2, NLOG, W / 2, TM, W>(cx, slab, v, t, sq);
        // Trips are kept apart where a trip is long (two or three levels at once, or the 4-byte loads of the channel going
        // straight to the on-chip level): letting the scheduler interleave them costs more registers than there are.
        // Measured, M frames/s without / with: N = 16384: 10.0 / 10.8, 8192: 27.4 / 30.7, 4096: 57 / 64; a barrier after
        // the other single steps loses (2048: 189 -> 163).
        if constexpr (D >= 2 || (LEV == NLOG && LEV == 12)) __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (M <= 11) {
#pragma unroll
        for (int t = 0; t < TM; t++) body(t);
    } else {
#pragma unroll 2
        for (int t = 0; t < TM; t++) body(t);
    }
}

// partial sums of a node of level L (>= 10) whose leaves are all frozen / all information
template <int L, int NLOG> __device__ __forceinline__ void big_rate0(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    glb_u32 *ps = (glb_u32 *)opaque_ptr(cx.slab) + Lay::ps_off(L, plane) + opaque_idx(cx.sub);
    for (int w = 0; w < (1 << (L - 9)); w++) ps[w * 16] = 0u;
}
template <int L, int NLOG> __device__ __forceinline__ void big_rate1(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    constexpr bool CHIP = BigCtx<NLOG>::CHIP;
    const int sq = opaque_idx(cx.sub);
    glb_float *slab = opaque_ptr(cx.slab);
    glb_u32 *ps = (glb_u32 *)slab + Lay::ps_off(L, plane);
    if constexpr (L == 10) {
#pragma unroll
        for (int w = 0; w < 2; w++) {
            uint32_t x = 0;
            if constexpr (CHIP) {
                const auto o = cx.d.operands();
                if (cx.d.right) {
#pragma unroll
                    for (int i = 0; i < 32; i++) x |= (cx.d.template ch<true>(o, 32 * w + i) < 0.0f ? 1u : 0u) << i;
                } else {
#pragma unroll
                    for (int i = 0; i < 32; i++) x |= (cx.d.template ch<false>(o, 32 * w + i) < 0.0f ? 1u : 0u) << i;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; i++) x |= (cx.d.Ch[32 * w + i] < 0.0f ? 1u : 0u) << i;
            }
            ps[w * 16 + sq] = x;
        }
    } else if constexpr (L == 11 && CHIP) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const int q = 32 * w + i;
                const float v = q < 64 ? cx.d.Ch[q < 64 ? q : 0] : cx.d.hi[64 * (q < 64 ? 0 : q - 64)];
                x |= (v < 0.0f ? 1u : 0u) << i;
            }
            ps[w * 16 + sq] = x;
        }
    } else {
        const glb_float *v = slab + Lay::llr_off(L);
        for (int w = 0; w < (1 << (L - 9)); w++) {
            uint32_t x = 0;
#pragma unroll 8
            for (int i = 0; i < 32; i++) x |= (v[64 * ((32 * w + i) >> 2) + 4 * sq + (i & 3)] < 0.0f ? 1u : 0u) << i;
            ps[w * 16 + sq] = x;
        }
    }
}
// parent (level LEV, plane) = [xl ^ xr, xr] of its children (level LEV-1, planes 0 and 1): whole words per lane
template <int LEV, int NLOG> __device__ __forceinline__ void big_combine(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    glb_u32 *base = (glb_u32 *)opaque_ptr(cx.slab) + opaque_idx(cx.sub);
    const glb_u32 *xl = base + Lay::ps_off(LEV - 1, 0), *xr = base + Lay::ps_off(LEV - 1, 1);
    glb_u32 *x = base + Lay::ps_off(LEV, plane);
    constexpr int nw = 1 << (LEV - 10);  // words per lane of a child
    for (int w = 0; w < nw; w++) {
        const uint32_t l = xl[w * 16], r = xr[w * 16];
        x[w * 16] = l ^ r;
        x[(nw + w) * 16] = r;
    }
}

// the step into child cidx of a node of level LEV with as many of the f steps down the child's left spine as apply:
// the spine node must be a mixed one and its left child not all-frozen.  Returns the number of levels written.
template <int LEV, int NLOG, bool IS_G> __device__ __forceinline__ int big_step(BigCtx<NLOG> &cx, uint32_t cidx, bool mixed)
{
    constexpr int DMAX = (LEV - BigCtx<NLOG>::FLOOR) < 3 ? (LEV - BigCtx<NLOG>::FLOOR) : 3;
    int D = 1;
#pragma unroll
    for (int k = 1; k < DMAX; k++) {
        // spine node of level LEV-k (the child itself for k = 1) and its left child
        const uint32_t si = cidx << (k - 1);
        const bool s_mixed = k == 1 ? mixed : !(((cx.t.up_r0[LEV - k - 10] | cx.t.up_r1[LEV - k - 10]) >> si) & 1ull);
        const bool left_frozen = (cx.t.up_r0[LEV - k - 1 - 10] >> (2 * si)) & 1ull;
        if (D == k && s_mixed && !left_frozen) D = k + 1;
    }
    if constexpr (DMAX >= 3) {
        if (D == 3) {
            big_chain<LEV, NLOG, IS_G, 3>(cx);
            return 3;
        }
    }
    if constexpr (DMAX >= 2) {
        if (D == 2) {
            big_chain<LEV, NLOG, IS_G, 2>(cx);
            return 2;
        }
    }
    big_chain<LEV, NLOG, IS_G, 1>(cx);
    return 1;
}

// node idx of level LEV >= 11, not all-frozen and not all-information; a two-trip loop over the children as in decode_node.
// have: how many levels of LLRs down this node's left spine (left child, its left child, ...) are there already.
template <int LEV, int NLOG> __device__ __forceinline__ void big_node(BigCtx<NLOG> &cx, uint32_t idx, int plane, int have)
{
    typedef BigLayout<NLOG> Lay;
    constexpr bool CHIP = BigCtx<NLOG>::CHIP;
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t cidx = 2 * idx + h;
        const bool is0 = (cx.t.up_r0[LEV - 1 - 10] >> cidx) & 1ull;
        const bool is1 = (cx.t.up_r1[LEV - 1 - 10] >> cidx) & 1ull;
        if (is0) {
            big_rate0<LEV - 1, NLOG>(cx, (int)h);  // an all-frozen child needs no LLRs
            continue;
        }
        int child_have = 0;
        if constexpr (CHIP && LEV == 11) {
            // the child's LLRs are made from this node's, which are on chip, whenever they are needed
            cx.d.right = h != 0;
            if (h) {
                const glb_u32 *ps = (const glb_u32 *)opaque_ptr(cx.slab) + Lay::ps_off(10, 0) + opaque_idx(cx.sub);
                cx.d.lbits = (uint64_t)ps[0] | ((uint64_t)ps[16] << 32);
            }
        } else {
            if (h == 0 && have > 0) child_have = have - 1;
            else if (h == 0) child_have = big_step<LEV, NLOG, false>(cx, cidx, !is1) - 1;
            else child_have = big_step<LEV, NLOG, true>(cx, cidx, !is1) - 1;
        }
        if (is1) {
            big_rate1<LEV - 1, NLOG>(cx, (int)h);
        } else if constexpr (LEV - 1 == 10) {
            // one 1024-leaf subtree in registers
            const int lane = (int)threadIdx.x;
            const uint32_t maskv = cx.t.maskw[32 * cidx + (lane & 31)];
            const uint32_t pidv = cx.t.pid16[64 * cidx + lane];
            const NodeClasses nc = cx.t.sub[cidx];
            cx.d.ps0 = cx.d.ps1 = 0;
            cx.d.root = 0;
#ifndef POLAR_BIG_SKIP_WALK  // timing experiment: everything but the subtree walks (results are wrong then)
            decode_node<10, 10, true>(cx.d, nc, maskv, pidv, nullptr, 0u, 0, cx.rc);
#else
            cx.d.root = (uint64_t)(maskv ^ pidv ^ nc.r0[0]) | ((uint64_t)__float_as_uint(cx.d.Ch[0] + cx.d.Ch[63]) << 32);
#endif
            glb_u32 *ps = (glb_u32 *)opaque_ptr(cx.slab) + Lay::ps_off(10, (int)h) + opaque_idx(cx.sub);
            ps[0] = (uint32_t)cx.d.root;
            ps[16] = (uint32_t)(cx.d.root >> 32);
        } else {
            big_node<LEV - 1, NLOG>(cx, cidx, (int)h, child_have);
        }
    }
    big_combine<LEV, NLOG>(cx, plane);
}

#ifndef POLAR_BIG_MINWAVES
#define POLAR_BIG_MINWAVES 2  // wavefronts per SIMD the register allocation aims for (N = 2048 keeps nothing of level 11 in registers: 3, 192 against 163 M frames/s)
#endif
template <int NLOG>
__global__ void __launch_bounds__(64, NLOG == 11 ? 3 : POLAR_BIG_MINWAVES) sc_big_kernel(const float *__restrict__ llr, BigTables t, float *__restrict__ slabs,
                                                    uint32_t *__restrict__ info_hat, int64_t B, int K, int Kw)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int N = 1 << NLOG;
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    // second half of the level-11 node, one column per lane; the output stage keeps the four frames' u_hat there
    constexpr int UHAT_LDS_WORDS = BigCtx<NLOG>::CHIP ? 64 * 64 : 256;
    __shared__ float hi_lds[UHAT_LDS_WORDS];
    BigCtx<NLOG> cx;
    cx.t = t;
    cx.sub = sub;
    cx.d.hi = (lds_float *)hi_lds + (BigCtx<NLOG>::CHIP ? lane : 0);
    cx.d.right = false;
    cx.d.lbits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        cx.rc.up[i] = (sub & (8 >> i)) != 0;
        cx.rc.lowmask[i] = cx.rc.up[i] ? 0u : 0xFFFFFFFFu;
        cx.rc.low31[i] = cx.rc.up[i] ? 0u : 0x80000000u;
    }
    cx.slab = slabs + ((size_t)blockIdx.x * 4 + grp) * Lay::FRAME_WORDS;
    for (int64_t f0 = (int64_t)blockIdx.x * 4; f0 < B; f0 += (int64_t)gridDim.x * 4) {
        const int64_t f = f0 + grp;
        const bool active = f < B;
        const int64_t fc = active ? f : B - 1;
        cx.chan = llr + fc * N;
        big_node<NLOG, NLOG>(cx, 0u, 0, 0);

        // u_hat = x_hat F^{(x)n} on the sliced root (bit j of lane s = element 16 j + s; 2^(NLOG-9) words per lane)
        uint32_t *x = reinterpret_cast<uint32_t *>(cx.slab) + Lay::ps_off(NLOG, 0);
        constexpr int W = 1 << (NLOG - 9);
        constexpr bool IN_LDS = 4 * 16 * W <= UHAT_LDS_WORDS;  // the four frames' u_hat fit the LDS area: transform in registers
        const uint32_t *xg;                                     // where the gather below reads word i of this frame's u_hat: xg[i]
        if constexpr (IN_LDS) {
            uint32_t xw[W];
#pragma unroll
            for (int w = 0; w < W; w++) xw[w] = x[w * 16 + sub];
            // stages 4 .. 8 pair bits inside a word, stages 9 .. pair words of the lane, stages 0 .. 3 lanes of the row
            const uint32_t odd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, odd2 = (sub & 2) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int w = 0; w < W; w++) {
                uint32_t v = xw[w];
                v ^= (v >> 1) & 0x55555555u;
                v ^= (v >> 2) & 0x33333333u;
                v ^= (v >> 4) & 0x0F0F0F0Fu;
                v ^= (v >> 8) & 0x00FF00FFu;
                v ^= (v >> 16) & 0x0000FFFFu;
                xw[w] = v;
            }
#pragma unroll
            for (int d = 1; d < W; d <<= 1)
#pragma unroll
                for (int w = 0; w < W; w++)
                    if (!(w & d)) xw[w] ^= xw[w + d];
            uint32_t *xl = reinterpret_cast<uint32_t *>(hi_lds) + grp * 16 * W;
#pragma unroll
            for (int w = 0; w < W; w++) {
                uint32_t v = xw[w];
                v ^= dpp_mov_u32<0xB1>(v & odd1);
                v ^= dpp_mov_u32<0x4E>(v & odd2);
                v ^= dpp_banks_u32<0x104, 0x5>(v);
                v ^= dpp_banks_u32<0x108, 0x3>(v);
                xl[w * 16 + sub] = v;
            }
            wave_sync();
            xg = xl;
        } else {
            // stages 4 .. 8 pair bits inside a word, stages 9 .. pair words of the lane
            for (int w = 0; w < W; w++) {
                uint32_t v = x[w * 16 + sub];
                v ^= (v >> 1) & 0x55555555u;
                v ^= (v >> 2) & 0x33333333u;
                v ^= (v >> 4) & 0x0F0F0F0Fu;
                v ^= (v >> 8) & 0x00FF00FFu;
                v ^= (v >> 16) & 0x0000FFFFu;
                x[w * 16 + sub] = v;
            }
            for (int d = 1; d < W; d <<= 1)
                for (int w = 0; w < W; w++)
                    if (!(w & d)) x[w * 16 + sub] ^= x[(w + d) * 16 + sub];
            // stages 0 .. 3 pair lanes of the row
            {
                const uint32_t odd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, odd2 = (sub & 2) ? 0xFFFFFFFFu : 0u;
                for (int w = 0; w < W; w++) {
                    uint32_t v = x[w * 16 + sub];
                    v ^= dpp_mov_u32<0xB1>(v & odd1);
                    v ^= dpp_mov_u32<0x4E>(v & odd2);
                    v ^= dpp_banks_u32<0x104, 0x5>(v);
                    v ^= dpp_banks_u32<0x108, 0x3>(v);
                    x[w * 16 + sub] = v;
                }
            }
            // the other lanes of the row read these words next
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            xg = x;
        }
        // lane s assembles output words s, s + 16, ...
        for (int w = sub; w < Kw; w += 16) {
            uint32_t acc = 0;
            const uint32_t *g = t.gather + w;
#pragma unroll 8
            for (int i = 0; i < 32; i++) {
                const uint32_t e = g[(size_t)i * t.KwPad];
                acc = __builtin_amdgcn_alignbit(xg[e >> 5] >> (e & 31u), acc, 1);
            }
            const int valid = K - 32 * w;
            if (valid < 32) acc &= (1u << valid) - 1u;
            if (active) info_hat[fc * Kw + w] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        wave_sync();  // the LDS area goes back to the decoder
    }
}

// ------------------------------------------------------------------------------------------ host side
struct BigDeviceTables {
    NodeClasses *sub = nullptr;
    uint64_t *up = nullptr;     // r0 then r1, 6 levels each
    uint32_t *gather = nullptr;
    int KwPad = 0;
};

bool sc_big_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr, int flags)
{
    return dispatch_sc(c, dtype, u_hat, dec_llr, flags) == KF_SC_BIG;
}

static int build_big_tables(polar_code *c)
{
    if (c->big_tables) return POLAR_OK;
    const int N = c->N, n = c->n;
    std::vector<NodeClasses> sub(N / 1024);
    memset(sub.data(), 0, sub.size() * sizeof(NodeClasses));
    for (int s = 0; s < N / 1024; s++)
        for (int lev = 4; lev <= 10; lev++) {
            const int w = 1 << lev;
            for (int i = 0; i < (1024 >> lev); i++) {
                int ninfo = 0;
                for (int k = 0; k < w; k++) ninfo += c->mask[(size_t)s * 1024 + (size_t)i * w + k];
                if (ninfo == 0) sub[s].r0[lev - 4] |= 1ull << i;
                if (ninfo == w) sub[s].r1[lev - 4] |= 1ull << i;
            }
        }
    std::vector<uint64_t> up(12, 0);
    for (int lev = 10; lev < n; lev++) {
        const size_t w = (size_t)1 << lev;
        for (int i = 0; i < (N >> lev); i++) {
            size_t ninfo = 0;
            for (size_t k = 0; k < w; k++) ninfo += c->mask[(size_t)i * w + k];
            if (ninfo == 0) up[lev - 10] |= 1ull << i;
            if (ninfo == w) up[6 + lev - 10] |= 1ull << i;
        }
    }
    // information bit k = 32 w + i at code position p: u_hat word (p / 16 / 32) * 16 + p % 16 of the root plane, bit (p / 16) % 32
    const int KwPad = (c->Kw + 15) & ~15;
    std::vector<uint32_t> gather((size_t)32 * KwPad, 0u);
    int k = 0;
    for (int p = 0; p < N; p++)
        if (c->mask[p]) {
            const uint32_t j = (uint32_t)p >> 4, s = (uint32_t)p & 15u;
            gather[(size_t)(k & 31) * KwPad + (k >> 5)] = ((((j >> 5) * 16u) + s) << 5) | (j & 31u);
            k++;
        }
    BigDeviceTables *t = new BigDeviceTables;
    t->KwPad = KwPad;
    c->big_tables = t;  // from here on sc_big_release gives back whatever exists, also on the error paths below
    hipError_t e = hipMalloc((void **)&t->sub, sub.size() * sizeof(NodeClasses));
    if (e == hipSuccess) e = hipMalloc((void **)&t->up, 12 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&t->gather, gather.size() * 4);
    if (e != hipSuccess) {
        polar_set_error("out of device memory for the long-code SC tables: %s", hipGetErrorString(e));
        sc_big_release(c);
        return POLAR_ENOMEM;
    }
    e = hipMemcpy(t->sub, sub.data(), sub.size() * sizeof(NodeClasses), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->up, up.data(), 12 * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->gather, gather.data(), gather.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        polar_set_error("long-code SC tables: %s", hipGetErrorString(e));
        sc_big_release(c);
        return POLAR_EHIP;
    }
    return POLAR_OK;
}

void sc_big_release(polar_code *c)
{
    BigDeviceTables *t = static_cast<BigDeviceTables *>(c->big_tables);
    if (!t) return;
    if (t->sub) (void)hipFree(t->sub);
    if (t->up) (void)hipFree(t->up);
    if (t->gather) (void)hipFree(t->gather);
    delete t;
    c->big_tables = nullptr;
}

template <int NLOG> static int launch_big_t(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B)
{
    polar_ctx *ctx = c->ctx;
    int rc;
    if ((rc = build_big_tables(c))) return rc;
    BigDeviceTables *dt = static_cast<BigDeviceTables *>(c->big_tables);
    auto kern = sc_big_kernel<NLOG>;
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, 0));
    if (occ < 1) occ = 1;
    if (const char *e = polar_switch("POLAR_BIG_WAVES")) occ = std::max(1, std::min(occ, atoi(e)));  // experiment: wavefronts per CU
    const int64_t need = (B + 3) / 4;
    const int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ);
    if ((rc = ctx->scl_slab.reserve((size_t)grid * 4 * BigLayout<NLOG>::FRAME_WORDS * 4))) return rc;
    BigTables t;
    t.maskw = c->d_maskw;
    t.pid16 = c->d_pid16;
    t.sub = dt->sub;
    t.up_r0 = dt->up;
    t.up_r1 = dt->up + 6;
    t.gather = dt->gather;
    t.KwPad = dt->KwPad;
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, ctx->stream, (const float *)llr, t, ctx->scl_slab.as<float>(), info_hat, B,
                       c->K, c->Kw);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_big_kernel<%d>", NLOG);
    c->last.lanes_per_frame = 16;
    c->last.lds_bytes_per_block = 0;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

int launch_sc_big(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B)
{
    if (B == 0) return POLAR_OK;
    switch (c->n) {
    case 11: return launch_big_t<11>(c, llr, info_hat, B);
    case 12: return launch_big_t<12>(c, llr, info_hat, B);
    case 13: return launch_big_t<13>(c, llr, info_hat, B);
    case 14: return launch_big_t<14>(c, llr, info_hat, B);
    case 15: return launch_big_t<15>(c, llr, info_hat, B);
    case 16: return launch_big_t<16>(c, llr, info_hat, B);
    default: polar_set_error("long-code SC kernel: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}
