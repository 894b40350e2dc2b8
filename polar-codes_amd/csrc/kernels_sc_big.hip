// SC decoder for codes longer than the register kernels hold (2048 <= N <= 65536, fp32, information bits).
//
// The same mapping as sc_reg_kernel -- sixteen lanes per frame, four frames per wavefront, element k of every
// tree level in lane k mod 16 -- with the levels of width >= 1024 in a slab of global memory per frame (level l at
// offset 2^l - 1024, N - 1024 floats in all; it stays in the L2 / Infinity Cache) and their partial sums beside it
// as per-lane bit slices.  A lane only ever touches the elements of its own residue, so the slab needs no
// synchronisation at all, and the order of a level inside the slab is free: element 16 j + s sits at index
// 64 (j / 4) + 4 s + j % 4, four elements of a lane side by side, so that the steps above the subtrees move 16 bytes
// per lane and 256 contiguous bytes per frame and instruction (4-byte accesses, 64 bytes per frame, reach only two
// thirds of the memory rate).  Every 1024-leaf subtree is handed to the register-resident walk of sc_reg_kernel
// (kernels_sc_node.h) with the slab's level 10 as its channel; all-frozen and all-information nodes of 1024 leaves
// and more are short-cut as in the other kernels.  The output stage transforms the sliced root (x_hat -> u_hat),
// writes it to the slab and gathers the information bits through a table.
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

// Slab of one frame, in 32-bit words: LLR levels 10 .. n-1, then the partial-sum planes.
// Partial sums of level l (10 <= l <= n): 2^(l-9) words per lane, word w of lane s at [w * 16 + s]; two planes.
template <int NLOG> struct BigLayout {
    static constexpr int N = 1 << NLOG;
    static constexpr int LLR_WORDS = N - 1024;
    static __device__ __host__ constexpr int llr_off(int l) { return (1 << l) - 1024; }
    static __device__ __host__ constexpr int ps_words(int l) { return 16 << (l - 9); }           // one plane of level l
    static __device__ __host__ constexpr int ps_off(int l, int plane)                            // levels 10 .. n
    {
        return LLR_WORDS + 2 * ((16 << (l - 9)) - 32) + plane * (16 << (l - 9));                  // 2 * sum_{k<l} ps_words(k)
    }
    static constexpr int FRAME_WORDS = LLR_WORDS + 2 * ((16 << (NLOG + 1 - 9)) - 32);
};

template <int NLOG> struct BigCtx {
    RegDec<float, 10, 0, false, true> d;  // the subtree's 1024 LLRs arrive in registers (d.Ch)
    RowConst rc;
    BigTables t;
    float *slab;          // this frame's slab
    const float *chan;    // this frame's channel LLRs
    int sub;
};

// the lane's index / the frame's slab pointer through an opaque copy (see big_step)
__device__ __forceinline__ int opaque_idx(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float *opaque_ptr(float *p)
{
    asm volatile("" : "+v"(p));
    return p;
}

// f / g from level LEV to LEV-1 (LEV >= 11): elements sub + 16 j of the lane's residue, four (j = 4 J .. 4 J + 3) per trip.
// The channel itself is in natural order (4-byte accesses); every level of the slab is in the order described above;
// level 10, the LLRs of one 1024-leaf subtree, is never stored: it goes to the registers cx.d.Ch (element 16 j + sub at Ch[j]).
// FUSE: the step that follows at once, the f from level LEV-1 to LEV-2 of the left child, is done on the way (its
// operands are trips J and J + trips/2 of this step's results), which saves reading level LEV-1 back.
template <int LEV, int NLOG, bool IS_G> __device__ __forceinline__ float4 big_trip(BigCtx<NLOG> &cx, const float *slab, const uint32_t *bits, int J, int sq)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int half = 1 << (LEV - 1);
    float4 a, b;
    if constexpr (LEV == NLOG) {
        const float *in = cx.chan + 64 * J + sq;
        a = make_float4(in[0], in[16], in[32], in[48]);
        b = make_float4(in[half], in[half + 16], in[half + 32], in[half + 48]);
    } else {
        const float *in = slab + Lay::llr_off(LEV) + 64 * J + 4 * sq;
        a = *reinterpret_cast<const float4 *>(in);
        b = *reinterpret_cast<const float4 *>(in + half);
    }
    if (IS_G) {
        const uint32_t w = bits[(J >> 3) * 16] >> ((4 * J) & 31);
        return make_float4(g_fast(a.x, b.x, w & 1u), g_fast(a.y, b.y, (w >> 1) & 1u), g_fast(a.z, b.z, (w >> 2) & 1u),
                           g_fast(a.w, b.w, (w >> 3) & 1u));
    }
    return make_float4(f_fast(a.x, b.x), f_fast(a.y, b.y), f_fast(a.z, b.z), f_fast(a.w, b.w));
}
template <int L, int NLOG> __device__ __forceinline__ void big_put(BigCtx<NLOG> &cx, float *slab, int J, const float4 &r, int sq)
{
    typedef BigLayout<NLOG> Lay;
    if constexpr (L == 10) {  // J is a compile-time constant here: the loops over level-10 trips are fully unrolled
        cx.d.Ch[4 * J] = r.x;
        cx.d.Ch[4 * J + 1] = r.y;
        cx.d.Ch[4 * J + 2] = r.z;
        cx.d.Ch[4 * J + 3] = r.w;
    } else {
        *reinterpret_cast<float4 *>(slab + Lay::llr_off(L) + 4 * sq + 64 * J) = r;
    }
}
template <int LEV, int NLOG, bool IS_G, bool FUSE> __device__ __forceinline__ void big_step(BigCtx<NLOG> &cx)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int half = 1 << (LEV - 1);
    // the lane's index goes through an opaque copy: addresses derived from it are formed here, base plus instruction
    // offset, instead of being hoisted out of the frame loop as one register pair per access
    int sq = cx.sub;
    asm volatile("" : "+v"(sq));
    float *slab = opaque_ptr(cx.slab);
    const uint32_t *bits = reinterpret_cast<const uint32_t *>(slab) + Lay::ps_off(LEV - 1, 0) + sq;
    constexpr int trips = half / 64;  // four elements per lane and trip
    if constexpr (!FUSE) {
        if constexpr (LEV - 1 == 10) {
#pragma unroll
            for (int J = 0; J < trips; J++) big_put<LEV - 1, NLOG>(cx, slab, J, big_trip<LEV, NLOG, IS_G>(cx, slab, bits, J, sq), sq);
        } else {
#pragma unroll 4
            for (int J = 0; J < trips; J++) big_put<LEV - 1, NLOG>(cx, slab, J, big_trip<LEV, NLOG, IS_G>(cx, slab, bits, J, sq), sq);
        }
    } else {
        static_assert(!FUSE || LEV >= 12, "the fused f writes level LEV-2 >= 10");
        auto body = [&](int J) {
            const float4 r1 = big_trip<LEV, NLOG, IS_G>(cx, slab, bits, J, sq), r2 = big_trip<LEV, NLOG, IS_G>(cx, slab, bits, J + trips / 2, sq);
            big_put<LEV - 1, NLOG>(cx, slab, J, r1, sq);
            big_put<LEV - 1, NLOG>(cx, slab, J + trips / 2, r2, sq);
            big_put<LEV - 2, NLOG>(cx, slab, J, make_float4(f_fast(r1.x, r2.x), f_fast(r1.y, r2.y), f_fast(r1.z, r2.z), f_fast(r1.w, r2.w)), sq);
        };
        if constexpr (LEV - 2 == 10) {
#pragma unroll
            for (int J = 0; J < trips / 2; J++) body(J);
        } else {
#pragma unroll 2
            for (int J = 0; J < trips / 2; J++) body(J);
        }
    }
}

// partial sums of a node of level L (>= 10) whose leaves are all frozen / all information
template <int L, int NLOG> __device__ __forceinline__ void big_rate0(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    uint32_t *ps = reinterpret_cast<uint32_t *>(opaque_ptr(cx.slab)) + Lay::ps_off(L, plane) + opaque_idx(cx.sub);
    for (int w = 0; w < (1 << (L - 9)); w++) ps[w * 16] = 0u;
}
template <int L, int NLOG> __device__ __forceinline__ void big_rate1(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    const int sq = opaque_idx(cx.sub);
    float *slab = opaque_ptr(cx.slab);
    uint32_t *ps = reinterpret_cast<uint32_t *>(slab) + Lay::ps_off(L, plane);
    if constexpr (L == 10) {  // the subtree's LLRs are in registers
#pragma unroll
        for (int w = 0; w < 2; w++) {
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) x |= (cx.d.Ch[32 * w + i] < 0.0f ? 1u : 0u) << i;
            ps[w * 16 + sq] = x;
        }
        return;
    }
    const float *v = slab + Lay::llr_off(L);
    for (int w = 0; w < (1 << (L - 9)); w++) {
        uint32_t x = 0;
#pragma unroll 8
        for (int i = 0; i < 32; i++) x |= (v[64 * ((32 * w + i) >> 2) + 4 * sq + (i & 3)] < 0.0f ? 1u : 0u) << i;
        ps[w * 16 + sq] = x;
    }
}
// parent (level LEV, plane) = [xl ^ xr, xr] of its children (level LEV-1, planes 0 and 1): whole words per lane
template <int LEV, int NLOG> __device__ __forceinline__ void big_combine(BigCtx<NLOG> &cx, int plane)
{
    typedef BigLayout<NLOG> Lay;
    uint32_t *base = reinterpret_cast<uint32_t *>(opaque_ptr(cx.slab)) + opaque_idx(cx.sub);
    const uint32_t *xl = base + Lay::ps_off(LEV - 1, 0), *xr = base + Lay::ps_off(LEV - 1, 1);
    uint32_t *x = base + Lay::ps_off(LEV, plane);
    constexpr int nw = 1 << (LEV - 10);  // words per lane of a child
    for (int w = 0; w < nw; w++) {
        const uint32_t l = xl[w * 16], r = xr[w * 16];
        x[w * 16] = l ^ r;
        x[(nw + w) * 16] = r;
    }
}

// node idx of level LEV >= 11, not all-frozen and not all-information; a two-trip loop over the children as in decode_node.
// have_f: the caller's step has already produced the LLRs of this node's left child (fused f).
template <int LEV, int NLOG> __device__ __forceinline__ void big_node(BigCtx<NLOG> &cx, uint32_t idx, int plane, bool have_f)
{
    typedef BigLayout<NLOG> Lay;
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t cidx = 2 * idx + h;
        const bool is0 = (cx.t.up_r0[LEV - 1 - 10] >> cidx) & 1ull;
        const bool is1 = (cx.t.up_r1[LEV - 1 - 10] >> cidx) & 1ull;
        if (is0) {
            big_rate0<LEV - 1, NLOG>(cx, (int)h);  // an all-frozen child needs no LLRs
            continue;
        }
        // the child's own first step, the f for its left child, rides along when the child is a mixed node above the
        // subtrees and that left child is not all-frozen
        bool fuse = false;
        if constexpr (LEV >= 12) fuse = !is1 && !((cx.t.up_r0[LEV - 2 - 10] >> (2 * cidx)) & 1ull);
        if (h == 0) {
            if (!have_f) {
                if constexpr (LEV >= 12) {
                    if (fuse) big_step<LEV, NLOG, false, true>(cx);
                    else big_step<LEV, NLOG, false, false>(cx);
                } else {
                    big_step<LEV, NLOG, false, false>(cx);
                }
            } else if constexpr (LEV >= 12) {
                // level LEV-1 is there already; the grandchild's f still has to be done by the child itself
                fuse = false;
            }
        } else {
            if constexpr (LEV >= 12) {
                if (fuse) big_step<LEV, NLOG, true, true>(cx);
                else big_step<LEV, NLOG, true, false>(cx);
            } else {
                big_step<LEV, NLOG, true, false>(cx);
            }
        }
        if (is1) {
            big_rate1<LEV - 1, NLOG>(cx, (int)h);
        } else if constexpr (LEV - 1 == 10) {
            // one 1024-leaf subtree in registers: its LLRs are in cx.d.Ch
            const int lane = (int)threadIdx.x;
            const uint32_t maskv = cx.t.maskw[32 * cidx + (lane & 31)];
            const uint32_t pidv = cx.t.pid16[64 * cidx + lane];
            const NodeClasses nc = cx.t.sub[cidx];
            cx.d.ps0 = cx.d.ps1 = 0;
            cx.d.root = 0;
            decode_node<10, 10, true>(cx.d, nc, maskv, pidv, nullptr, 0u, 0, cx.rc);
            uint32_t *ps = reinterpret_cast<uint32_t *>(opaque_ptr(cx.slab)) + Lay::ps_off(10, (int)h) + opaque_idx(cx.sub);
            ps[0] = (uint32_t)cx.d.root;
            ps[16] = (uint32_t)(cx.d.root >> 32);
        } else {
            big_node<LEV - 1, NLOG>(cx, cidx, (int)h, fuse);
        }
    }
    big_combine<LEV, NLOG>(cx, plane);
}

template <int NLOG>
__global__ void __launch_bounds__(64, 3) sc_big_kernel(const float *__restrict__ llr, BigTables t, float *__restrict__ slabs,
                                                    uint32_t *__restrict__ info_hat, int64_t B, int K, int Kw)
{
    typedef BigLayout<NLOG> Lay;
    constexpr int N = 1 << NLOG;
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    BigCtx<NLOG> cx;
    cx.t = t;
    cx.sub = sub;
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
        big_node<NLOG, NLOG>(cx, 0u, 0, false);

        // u_hat = x_hat F^{(x)n} on the sliced root (bit j of lane s = element 16 j + s; 2^(NLOG-9) words per lane)
        uint32_t *x = reinterpret_cast<uint32_t *>(cx.slab) + Lay::ps_off(NLOG, 0);
        constexpr int W = 1 << (NLOG - 9);
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
        // lane s assembles output words s, s + 16, ...
        for (int w = sub; w < Kw; w += 16) {
            uint32_t acc = 0;
            const uint32_t *g = t.gather + w;
#pragma unroll 8
            for (int i = 0; i < 32; i++) {
                const uint32_t e = g[(size_t)i * t.KwPad];
                acc = __builtin_amdgcn_alignbit(x[e >> 5] >> (e & 31u), acc, 1);
            }
            const int valid = K - 32 * w;
            if (valid < 32) acc &= (1u << valid) - 1u;
            if (active) info_hat[fc * Kw + w] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
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
    if (u_hat || dec_llr) return false;
    if (dtype != POLAR_F32 || !(flags & POLAR_SC_FAST)) return false;
    if (c->n < 11 || c->n > 16) return false;
    if (c->K == 0 || c->K == c->N) return false;
    const char *e = getenv("POLAR_SC_KERNEL");
    if (e && e[0] == 'l') return false;  // "lds": force the LDS-staged kernel
    return true;
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
    if (hipMalloc((void **)&t->sub, sub.size() * sizeof(NodeClasses)) != hipSuccess || hipMalloc((void **)&t->up, 12 * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&t->gather, gather.size() * 4) != hipSuccess) {
        polar_set_error("out of device memory for the long-code SC tables");
        return POLAR_ENOMEM;
    }
    HIP_TRY(hipMemcpy(t->sub, sub.data(), sub.size() * sizeof(NodeClasses), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->up, up.data(), 12 * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->gather, gather.data(), gather.size() * 4, hipMemcpyHostToDevice));
    c->big_tables = t;
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
    if (const char *e = getenv("POLAR_BIG_WAVES")) occ = std::max(1, std::min(occ, atoi(e)));  // experiment: wavefronts per CU
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
