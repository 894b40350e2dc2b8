// Batched successive-cancellation decoder.
//
// A group of G lanes (G = 8, 16, 32 or 64) of one wavefront decodes one frame; a wavefront therefore
// carries 64/G frames, all walking the same schedule (the schedule depends only on the code), so
// control flow is wave-uniform.  Per frame the LDS holds
//   - the LLR tree below the channel level: level l (2^l values) at element offset 2^l,
//   - two planes of bit-packed partial sums: level l >= 5 at word offset 2^(l-5); the five narrow
//     levels (1..16 bits) share one 32-bit register per plane (level l at bit offset 2^l).
// The channel LLRs are read straight from global memory.  For N >= 128 level n-1 is not stored either: the
// operations that read it recompute its values from the channel pair (and the left half's partial sums),
// which halves the LDS per frame.
// Lane `sub` of a group handles elements sub, sub+G, ... so LDS accesses of a group are contiguous.
// Workgroups are single wavefronts: no workgroup barrier is ever needed.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "decode_common.h"
#include "reg_common.h"
#include "polar_internal.h"

template <typename T> struct SCParams {
    const T *llr;
    const uint32_t *ops;
    const uint32_t *A;
    uint32_t *info_hat;
    uint32_t *u_hat;
    T *dec_llr;
    int64_t B;
    int N, n, Nw, K, Kw;
    int stride_bytes;  // LDS bytes between the frames of one wavefront
    int tree_bytes;    // LDS bytes of one frame's stored LLR levels
};

template <typename T, int G> __global__ void __launch_bounds__(64) sc_decode_kernel(SCParams<T> p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FPW = 64 / G;
    const int lane = threadIdx.x;
    const int grp = lane / G;
    const int sub = lane % G;
    unsigned char *base = smem + (size_t)grp * p.stride_bytes;
    T *lv = reinterpret_cast<T *>(base);
    uint32_t *p0 = reinterpret_cast<uint32_t *>(base + (size_t)p.tree_bytes);
    uint32_t *p1 = p0 + 2 * p.Nw;
    const int N = p.N, n = p.n, Nw = p.Nw;
    RowConst rc;  // position of this lane in its 16-lane row, for the in-register 16-leaf nodes
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (lane & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }

    for (int64_t f0 = (int64_t)blockIdx.x * FPW; f0 < p.B; f0 += (int64_t)gridDim.x * FPW) {
        const int64_t f = f0 + grp;
        const bool active = f < p.B;
        const int64_t fc = active ? f : p.B - 1;  // idle groups shadow the last frame, stores masked
        const T *chan = p.llr + fc * N;
        uint32_t nar0 = 0, nar1 = 0;
        const bool virt = n >= 7;  // level n-1 is virtual (its own and its children's partial sums are then whole words)
        bool right_half = false;
        auto top = [&](int k) -> T {
            const T a = chan[k], b = chan[k + (N >> 1)];
            if (!right_half) return f_minsum(a, b);
            const uint32_t bit = (p0[(1 << (n - 6)) + (k >> 5)] >> (k & 31)) & 1u;
            return bit ? b - a : b + a;
        };

        for (int pc = 0;; pc++) {
            const uint32_t op = p.ops[pc];
            const uint32_t type = OP_TYPE(op);
            if (type == OP_END) break;
            const int lev = (int)OP_LEVEL(op);
            const int plane = (int)OP_PLANE(op);
            switch (type) {
            case OP_F: {
                const int w = 1 << (lev - 1);
                if (virt && lev == n) {
                    right_half = false;
                    break;
                }
                if (virt && lev == n - 1) {
                    for (int k = sub; k < w; k += G) lv[w + k] = f_minsum(top(k), top(k + w));
                    break;
                }
                const T *src = (lev == n) ? chan : lv + (1 << lev);
                for (int k = sub; k < w; k += G) lv[w + k] = f_minsum(src[k], src[k + w]);
                break;
            }
            case OP_G: {
                const int w = 1 << (lev - 1);
                if (virt && lev == n) {
                    right_half = true;
                    break;
                }
                if (virt && lev == n - 1) {
                    const uint32_t *bits = p0 + (1 << (lev - 6));
                    for (int k = sub; k < w; k += G) {
                        T a = top(k), b = top(k + w);
                        uint32_t bit = (bits[k >> 5] >> (k & 31)) & 1u;
                        lv[w + k] = bit ? b - a : b + a;
                    }
                    break;
                }
                const T *src = (lev == n) ? chan : lv + (1 << lev);
                if (lev - 1 >= 5) {
                    const uint32_t *bits = p0 + (1 << (lev - 6));
                    for (int k = sub; k < w; k += G) {
                        T a = src[k], b = src[k + w];
                        uint32_t bit = (bits[k >> 5] >> (k & 31)) & 1u;
                        lv[w + k] = bit ? b - a : b + a;
                    }
                } else {
                    for (int k = sub; k < w; k += G) {
                        T a = src[k], b = src[k + w];
                        uint32_t bit = (nar0 >> (w + k)) & 1u;
                        lv[w + k] = bit ? b - a : b + a;
                    }
                }
                break;
            }
            case OP_LEAF_INFO:
            case OP_LEAF_FROZEN: {
                T v = lv[1];
                uint32_t bit = (type == OP_LEAF_INFO && v < (T)0) ? 1u : 0u;
                if (plane) nar1 = (nar1 & ~2u) | (bit << 1);
                else nar0 = (nar0 & ~2u) | (bit << 1);
                if (p.dec_llr && sub == 0 && active) p.dec_llr[f * N + OP_LEAFIDX(op)] = v;
                break;
            }
            case OP_SUB16: {
                // the node's 16 LLRs, one per lane of a row (every row of the group does the same work), are
                // decoded by the replicated butterfly; lane e of the row ends up with partial-sum bit e
                if constexpr (G >= 16) {
                    const T v = lv[16 + (lane & 15)];
                    const uint32_t x = sub_node<16, true>(v, OP_LEAFIDX(op) & 0xFFFFu, rc);
                    const uint32_t bits = (uint32_t)group_ballot<G>(x != 0u, grp) & 0xFFFFu;
                    if (plane) nar1 = (nar1 & 0x0000FFFFu) | (bits << 16);
                    else nar0 = (nar0 & 0x0000FFFFu) | (bits << 16);
                }
                break;
            }
            case OP_R0: {
                if (lev >= 5) {
                    uint32_t *dst = (plane ? p1 : p0) + (1 << (lev - 5));
                    for (int j = sub; j < (1 << (lev - 5)); j += G) dst[j] = 0u;
                } else {
                    const int w = 1 << lev;
                    const uint32_t m = ((1u << w) - 1u) << w;
                    if (plane) nar1 &= ~m;
                    else nar0 &= ~m;
                }
                break;
            }
            case OP_R1: {
                const int w = 1 << lev;
                const T *src = (lev == n) ? chan : lv + w;
                if (virt && lev == n - 1) {
                    uint32_t *dst = (plane ? p1 : p0) + (1 << (lev - 5));
                    pack_bits_group<G>(
                        w, sub, grp, [&](int k) { return top(k) < (T)0; },
                        [&](int wi, uint32_t v) {
                            if (sub == 0) dst[wi] = v;
                        });
                } else if (lev >= 5) {
                    uint32_t *dst = (plane ? p1 : p0) + (1 << (lev - 5));
                    pack_bits_group<G>(
                        w, sub, grp, [&](int k) { return src[k] < (T)0; },
                        [&](int wi, uint32_t v) {
                            if (sub == 0) dst[wi] = v;
                        });
                } else {
                    uint32_t acc = 0;
                    pack_bits_group<G>(
                        w, sub, grp, [&](int k) { return src[k] < (T)0; }, [&](int, uint32_t v) { acc = v; });
                    const uint32_t m = ((1u << w) - 1u) << w;
                    if (plane) nar1 = (nar1 & ~m) | (acc << w);
                    else nar0 = (nar0 & ~m) | (acc << w);
                }
                break;
            }
            case OP_COMBINE: {
                const int cl = lev - 1;  // children level
                if (cl >= 5) {
                    const int nwc = 1 << (cl - 5);
                    const uint32_t *s0 = p0 + nwc, *s1 = p1 + nwc;
                    uint32_t *dst = (plane ? p1 : p0) + 2 * nwc;
                    for (int j = sub; j < nwc; j += G) {
                        uint32_t a = s0[j], b = s1[j];
                        dst[j] = a ^ b;
                        dst[nwc + j] = b;
                    }
                } else if (cl == 4) {
                    uint32_t a = nar0 >> 16, b = nar1 >> 16;
                    uint32_t v = (a ^ b) | (b << 16);
                    if (sub == 0) (plane ? p1 : p0)[1] = v;
                } else {
                    const int hw = 1 << cl, w = 2 * hw;
                    const uint32_t hm = (1u << hw) - 1u;
                    uint32_t a = (nar0 >> hw) & hm, b = (nar1 >> hw) & hm;
                    uint32_t v = (a ^ b) | (b << hw);
                    const uint32_t m = ((1u << w) - 1u) << w;
                    if (plane) nar1 = (nar1 & ~m) | (v << w);
                    else nar0 = (nar0 & ~m) | (v << w);
                }
                break;
            }
            default: break;
            }
            wave_sync();
        }

        // Root partial sums are the re-encoded codeword estimate; u_hat = x_hat F^{(x)n}.
        uint32_t *uw = p1;  // reuse plane 1 as scratch: Nw words
        if (n >= 5) {
            for (int j = sub; j < Nw; j += G) uw[j] = butterfly_word(p0[Nw + j], n);
            for (int s = 5; s < n; s++) {
                wave_sync();
                const int d = 1 << (s - 5);
                for (int j = sub; j < Nw; j += G)
                    if (!(j & d)) uw[j] ^= uw[j + d];
            }
        } else {
            uint32_t x = (nar0 >> N) & ((1u << N) - 1u);
            if (sub == 0) uw[0] = butterfly_word(x, n);
        }
        wave_sync();
        if (p.u_hat && active)
            for (int j = sub; j < Nw; j += G) p.u_hat[f * Nw + j] = uw[j];
        if (p.info_hat) {
            uint32_t *dst = p.info_hat + fc * p.Kw;
            pack_bits_group<G>(
                p.K, sub, grp,
                [&](int k) {
                    uint32_t pos = p.A[k];
                    return ((uw[pos >> 5] >> (pos & 31)) & 1u) != 0u;
                },
                [&](int wi, uint32_t v) {
                    if (sub == 0 && active) dst[wi] = v;
                });
            if (p.K == 0 && sub == 0 && active) dst[0] = 0u;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------ launch
static int pick_lanes(int N, size_t elem)
{
    const char *e = polar_switch("POLAR_SC_LANES");
    if (e) {
        int g = atoi(e);
        if (g == 8 || g == 16 || g == 32 || g == 64) return g;
    }
    size_t tree = (size_t)N * elem;
    if (tree >= 16384) return 64;
    if (tree >= 8192) return 32;
    if (N >= 256) return 16;
    return 8;
}

template <typename T, int G>
static int launch_sc_t(polar_code *c, const void *llr, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B, int flags)
{
    polar_ctx *ctx = c->ctx;
    constexpr int FPW = 64 / G;
    SCParams<T> p;
    p.llr = (const T *)llr;
    // 16-leaf nodes run as single in-register operations unless the caller wants the per-leaf decision LLRs
    // or the group is narrower than a 16-lane row
    const bool sub16 = (G >= 16) && dec_llr == nullptr && c->n >= 5;
    p.ops = c->d_sched[((flags & POLAR_SC_FAST) ? 1 : 0) + (sub16 ? 2 : 0)];
    p.A = c->d_A;
    p.info_hat = info_hat;
    p.u_hat = u_hat;
    p.dec_llr = (T *)dec_llr;
    p.B = B;
    p.N = c->N;
    p.n = c->n;
    p.Nw = c->Nw;
    p.K = c->K;
    p.Kw = c->Kw;
    // frame stride: tree + two planes of 2*Nw words, rounded to 256 B, then skewed by one group's
    // worth of banks so that the groups of a wavefront do not collide on LDS banks
    p.tree_bytes = (int)((size_t)(c->n >= 7 ? c->N / 2 : c->N) * sizeof(T));
    size_t raw = (size_t)p.tree_bytes + (size_t)4 * c->Nw * 4;
    size_t stride = (raw + 255) & ~(size_t)255;
    if (FPW > 1) stride += ((size_t)G * sizeof(T)) % 256;
    p.stride_bytes = (int)stride;
    size_t lds = stride * FPW;
    if (lds > 160 * 1024) {
        polar_set_error("SC decoder: N=%d needs %zu B of LDS per wavefront (limit 163840)", c->N, lds);
        return POLAR_EUNSUPPORTED;
    }
    auto kern = sc_decode_kernel<T, G>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds));
    if (occ < 1) occ = 1;
    int64_t need = (B + FPW - 1) / FPW;
    int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ * 4);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, p);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_decode_kernel<%s,%d>%s", sizeof(T) == 4 ? "float" : "double", G,
             (flags & POLAR_SC_FAST) ? "/fast" : "");
    c->last.lanes_per_frame = G;
    c->last.lds_bytes_per_block = (int)lds;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <typename T>
static int launch_sc_g(polar_code *c, const void *llr, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B, int flags)
{
    switch (pick_lanes(c->N, sizeof(T))) {
    case 8: return launch_sc_t<T, 8>(c, llr, info_hat, u_hat, dec_llr, B, flags);
    case 16: return launch_sc_t<T, 16>(c, llr, info_hat, u_hat, dec_llr, B, flags);
    case 32: return launch_sc_t<T, 32>(c, llr, info_hat, u_hat, dec_llr, B, flags);
    default: return launch_sc_t<T, 64>(c, llr, info_hat, u_hat, dec_llr, B, flags);
    }
}

int launch_sc_decode(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B,
                     int flags)
{
    if (B == 0) return POLAR_OK;
    switch (dispatch_sc(c, dtype, u_hat, dec_llr, flags)) {
    case KF_SC_STATIC:
    case KF_SC_REG: return launch_sc_reg(c, llr, dtype, info_hat, B, flags);
    case KF_SC_REG_FULL: return launch_sc_reg_full(c, llr, dtype, info_hat, u_hat, dec_llr, B);
    case KF_SC_BIG: return launch_sc_big(c, llr, info_hat, B);
    case KF_SC_LDS:
        if (dtype == POLAR_F32) return launch_sc_g<float>(c, llr, info_hat, u_hat, dec_llr, B, flags);
        return launch_sc_g<double>(c, llr, info_hat, u_hat, dec_llr, B, flags);
    default:
        polar_set_error("SC decoder: no kernel serves N=%d, K=%d, %s with these outputs (polar_dispatch_table lists what is served)", c->N,
                        c->K, dtype == POLAR_F32 ? "fp32" : "fp64");
        return POLAR_EUNSUPPORTED;
    }
}
