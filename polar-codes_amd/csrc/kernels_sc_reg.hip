// Register-resident successive-cancellation decoder (N = 32 .. 1024).
//
// Sixteen lanes (one DPP row) decode one frame, four frames per wavefront.  Element k of a tree level
// lives in lane k mod 16, so for every level of width >= 32 the pair (k, k + width/2) that f and g
// combine sits in the SAME lane: those levels are plain per-lane register arithmetic on a slice of
// N/16 values per lane, statically indexed (the level is dispatched through a switch so every array
// index is a compile-time constant).  Partial sums are sliced the same way, one bit per element,
// level l occupying bits [2^(l-4), 2^(l-3)) of a 64-bit register per plane; combining two children is
// (xl ^ xr) | (xr << c) in each lane.
// The bottom four levels (a 16-leaf subtree, one LLR per lane) are decoded with DPP row rotations:
// after each f/g step the values are replicated with half the period, every lane runs the same
// instruction stream and keeps the partial-sum bit of "its" element.
// No LDS is used on the decoding path; the channel LLRs are read from global memory by the two
// top-level operations.  LDS holds the output stage: 8 bytes of u_hat per lane and the table (two bytes per
// information bit) that says where each information bit sits in them.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "decode_common.h"
#include "reg_common.h"
#include "kernels_sc_static.h"
#include "kernels_sc_node.h"
#include "polar_internal.h"

template <typename T, int NLOG, bool FAST>
__global__ void __launch_bounds__(64, 4) sc_reg_kernel(const T *__restrict__ llr, const uint32_t *__restrict__ maskw, NodeClasses nc,
                                                    const uint32_t *__restrict__ pid16, const uint32_t *__restrict__ ctab, uint32_t *__restrict__ info_hat, int64_t B,
                                                    int K, int Kw, const uint32_t *__restrict__ truth, int nbits,
                                                    unsigned long long *__restrict__ counters)
{
    __shared__ __attribute__((aligned(32))) uint32_t stage[SC_STAGE_WORDS];  // per frame: 16 lanes x 64-bit slice of u_hat
    constexpr int N = 1 << NLOG;
    __shared__ __attribute__((aligned(16))) uint32_t ctab_s[256];  // the compress / place table of the output stage
    for (int k = threadIdx.x; k < 256; k += 64) ctab_s[k] = ctab[k];
    wave_sync();
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    RegDec<T, NLOG> d;
    ErrAcc acc;
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (sub & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }
    const uint32_t maskv = maskw[(lane & 31) < (N / 32) ? (lane & 31) : 0];
    const uint32_t pidv = pid16[lane < (N / 16) ? lane : 0];

    for (int64_t f0 = (int64_t)blockIdx.x * 4; f0 < B; f0 += (int64_t)gridDim.x * 4) {
        const int64_t f = f0 + grp;
        const bool active = f < B;
        const int64_t fc = active ? f : B - 1;
        const T *chan_lane = llr + fc * N + sub;
        d.ps0 = d.ps1 = 0;
        d.root = 0;
        decode_node<NLOG, NLOG, FAST>(d, nc, maskv, pidv, chan_lane, 0u, 0, rc);

        sc_reg_epilogue_c<NLOG>(d.root, stage, ctab_s, rc, info_hat, fc, active, lane, K, Kw, truth, nbits, &acc);
    }
    if (truth) err_acc_flush<1>(acc, counters, nullptr);
}

// The every-leaf walk with the reference's side outputs: u_hat (packed, natural order) and the decision LLRs of all N
// leaves (self.uHat / self.llrs of scDecoder, cn/polarCode.py:1525-1571), fp32 or fp64.  fp64 is the reference's own
// arithmetic: f and g only compare, negate, add and subtract, so bits and decision LLRs equal the reference's exactly.
// Same mapping as sc_reg_kernel; the fp64 slice takes 128 registers per lane, two wavefronts per SIMD.
template <typename T, int NLOG>
__global__ void __launch_bounds__(64, sizeof(T) == 8 ? 2 : 4)
    sc_reg_full_kernel(const T *__restrict__ llr, const uint32_t *__restrict__ maskw, const uint16_t *__restrict__ Asl, uint32_t *__restrict__ info_hat,
                       uint32_t *__restrict__ u_hat, T *__restrict__ dec_llr, int64_t B, int K, int Kw)
{
    __shared__ __attribute__((aligned(32))) uint32_t stage[SC_STAGE_WORDS];
    constexpr int N = 1 << NLOG;
    constexpr int NW = N / 32;
    __shared__ uint16_t asl_s[GatherShape<NLOG>::WORDS];
    for (int k = threadIdx.x; k < GatherShape<NLOG>::WORDS; k += 64) asl_s[k] = Asl[k];
    wave_sync();
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    RegDec<T, NLOG> d;
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (sub & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }
    const uint32_t maskv = maskw[(lane & 31) < NW ? (lane & 31) : 0];
    NodeClasses nc;  // every leaf is visited: no node is short-cut
#pragma unroll
    for (int i = 0; i < 7; i++) nc.r0[i] = nc.r1[i] = 0;

    for (int64_t f0 = (int64_t)blockIdx.x * 4; f0 < B; f0 += (int64_t)gridDim.x * 4) {
        const int64_t f = f0 + grp;
        const bool active = f < B;
        const int64_t fc = active ? f : B - 1;  // rows past the end decode the last frame again and store the same values
        d.ps0 = d.ps1 = 0;
        d.root = 0;
        decode_node<NLOG, NLOG, false, RegDec<T, NLOG>, true>(d, nc, maskv, 0u, llr + fc * N + sub, 0u, 0, rc,
                                                             dec_llr ? dec_llr + fc * N + sub : nullptr, sub);
        sc_reg_epilogue<NLOG>(d.root, stage, asl_s, info_hat, fc, active, lane, K, Kw);
        if (u_hat) {
            // u_hat is staged sliced (element k = bit k / 16 of lane k % 16: its low word at stage[16 l + g], its high word
            // at stage[16 l + 8 + g]); lane `sub` assembles the natural-order words sub and sub + 16: bit i of word w is
            // element 32 w + i = bit 2 w + i / 16 of lane i % 16
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int w = sub + 16 * s;
                if (w < NW) {
                    uint32_t v = 0;
#pragma unroll
                    for (int l = 0; l < 16; l++) {
                        const uint32_t sw = stage[16 * l + ((2 * w) >> 5) * 8 + grp];
                        const uint32_t two = (sw >> ((2 * w) & 31)) & 3u;
                        v |= (two & 1u) << l;
                        v |= (two >> 1) << (16 + l);
                    }
                    if (active) u_hat[fc * NW + w] = v;
                }
            }
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------ host side
// Profile number (1..25) of a 16-leaf rate profile, 0 if it is not one of the compiled ones.
uint32_t sc_reg_profile_id(uint32_t m16)
{
    switch (m16) {
#define POLAR_ID16(ID, MASK) \
    case MASK: return ID;
        POLAR_PROFILES16(POLAR_ID16)
#undef POLAR_ID16
    default: return 0u;
    }
}

static NodeClasses classify_nodes(const polar_code *c, bool fast)
{
    NodeClasses nc;
    memset(&nc, 0, sizeof(nc));
    if (!fast) return nc;
    for (int lev = 4; lev <= c->n && lev <= 10; lev++) {
        const int w = 1 << lev;
        for (int i = 0; i < (c->N >> lev); i++) {
            int ninfo = 0;
            for (int k = 0; k < w; k++) ninfo += c->mask[(size_t)i * w + k];
            if (ninfo == 0) nc.r0[lev - 4] |= 1ull << i;
            if (ninfo == w) nc.r1[lev - 4] |= 1ull << i;
        }
    }
    return nc;
}

// the predicates below ask the dispatch table (host_tables.cpp): one place says which kernel serves what
bool sc_reg_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr)
{
    // either schedule: the register kernels cover both (the specialised one needs the short-cut schedule and takes over there)
    const int f = dispatch_sc(c, dtype, u_hat, dec_llr, POLAR_SC_FAST);
    return f == KF_SC_STATIC || f == KF_SC_REG;
}

// every-leaf decoding with u_hat / decision LLRs, and all fp64 decoding, N = 32 .. 1024
bool sc_reg_full_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr, int flags)
{
    return dispatch_sc(c, dtype, u_hat, dec_llr, flags) == KF_SC_REG_FULL;
}

template <typename T, int NLOG> static int launch_full_t(polar_code *c, const void *llr, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B)
{
    polar_ctx *ctx = c->ctx;
    auto kern = sc_reg_full_kernel<T, NLOG>;
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, 0));
    if (occ < 1) occ = 1;
    const int64_t need = (B + 3) / 4;
    const int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ * 8);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, ctx->stream, (const T *)llr, (const uint32_t *)c->d_maskw, (const uint16_t *)c->d_Asl,
                       info_hat, u_hat, (T *)dec_llr, B, c->K, c->Kw);
    HIP_TRY(hipGetLastError());
    const float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_reg_full_kernel<%s,%d>", sizeof(T) == 4 ? "float" : "double", NLOG);
    c->last.lanes_per_frame = 16;
    c->last.lds_bytes_per_block = 4 * SC_STAGE_WORDS + 2 * GatherShape<NLOG>::WORDS;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <typename T> static int launch_full_n(polar_code *c, const void *llr, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B)
{
    switch (c->n) {
    case 5: return launch_full_t<T, 5>(c, llr, info_hat, u_hat, dec_llr, B);
    case 6: return launch_full_t<T, 6>(c, llr, info_hat, u_hat, dec_llr, B);
    case 7: return launch_full_t<T, 7>(c, llr, info_hat, u_hat, dec_llr, B);
    case 8: return launch_full_t<T, 8>(c, llr, info_hat, u_hat, dec_llr, B);
    case 9: return launch_full_t<T, 9>(c, llr, info_hat, u_hat, dec_llr, B);
    case 10: return launch_full_t<T, 10>(c, llr, info_hat, u_hat, dec_llr, B);
    default: polar_set_error("register SC kernel: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}

int launch_sc_reg_full(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B)
{
    if (B == 0) return POLAR_OK;
    if (dtype == POLAR_F32) return launch_full_n<float>(c, llr, info_hat, u_hat, dec_llr, B);
    return launch_full_n<double>(c, llr, info_hat, u_hat, dec_llr, B);
}

template <typename T, int NLOG, bool FAST> static int launch_reg_t(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B)
{
    polar_ctx *ctx = c->ctx;
    const NodeClasses nc = classify_nodes(c, FAST);
    auto kern = sc_reg_kernel<T, NLOG, FAST>;
    // POLAR_SC_LDSPAD (bytes of unused dynamic LDS per wavefront) lowers the occupancy for experiments
    const char *pe = polar_switch("POLAR_SC_LDSPAD");
    const size_t pad = pe ? (size_t)atoi(pe) : 0;
    if (pad > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, pad));
    if (occ < 1) occ = 1;
    int64_t need = (B + 3) / 4;
    // workgroups per resident slot: 573 / 592 / 605 / 607 M frames/s at (1024,512) with 4 / 8 / 16 / 32; 8 while the staged
    // Monte-Carlo loop's generator shares the GPU (a long queue of decoder workgroups starves the other stream)
    int rounds = ctx->overlapped ? 8 : 16;
    if (const char *e = polar_switch("POLAR_SC_ROUNDS")) rounds = std::max(1, atoi(e));
    int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ * rounds);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), pad, ctx->stream, (const T *)llr, (const uint32_t *)c->d_maskw, nc,
                       (const uint32_t *)c->d_pid16, (const uint32_t *)c->d_ctab, info_hat, B, c->K, c->Kw, c->count_truth, c->count_nbits,
                       (unsigned long long *)c->count_dev);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_reg_kernel<%s,%d>%s", sizeof(T) == 4 ? "float" : "double", NLOG,
             FAST ? "/fast" : "");
    c->last.lanes_per_frame = 16;
    c->last.lds_bytes_per_block = 4 * SC_STAGE_WORDS + 2 * GatherShape<NLOG>::WORDS;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <typename T> static int launch_reg_n(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B, int flags)
{
    switch (c->n) {
    case 5: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 5, true>(c, llr, info_hat, B) : launch_reg_t<T, 5, false>(c, llr, info_hat, B);
    case 6: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 6, true>(c, llr, info_hat, B) : launch_reg_t<T, 6, false>(c, llr, info_hat, B);
    case 7: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 7, true>(c, llr, info_hat, B) : launch_reg_t<T, 7, false>(c, llr, info_hat, B);
    case 8: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 8, true>(c, llr, info_hat, B) : launch_reg_t<T, 8, false>(c, llr, info_hat, B);
    case 9: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 9, true>(c, llr, info_hat, B) : launch_reg_t<T, 9, false>(c, llr, info_hat, B);
    case 10: return (flags & POLAR_SC_FAST) ? launch_reg_t<T, 10, true>(c, llr, info_hat, B) : launch_reg_t<T, 10, false>(c, llr, info_hat, B);
    default: polar_set_error("register SC kernel: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}

int launch_sc_reg(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, int64_t B, int flags)
{
    if (B == 0) return POLAR_OK;
    if (dispatch_sc(c, dtype, nullptr, nullptr, flags) == KF_SC_STATIC) return launch_sc_static(c, llr, info_hat, B);
    if (dtype != POLAR_F32) {
        polar_set_error("register SC kernel: fp32 only");
        return POLAR_EUNSUPPORTED;
    }
    return launch_reg_n<float>(c, llr, info_hat, B, flags);
}
