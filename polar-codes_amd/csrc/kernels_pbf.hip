// SC decoding with flipped decisions, and the progressive bit-flipping search built on it.
//
// One wavefront decodes one frame (LLR tree, partial sums, decision LLRs and the search state in LDS) and runs
// the whole search for it: SC; if the CRC fails, flip candidates (first leaves of the maximal all-information
// subtrees) are ranked by |decision LLR| / sqrt(mean LLR) and tried in that order, one SC run each, level by
// level up to `max_level` simultaneous flips.  The reference keeps a queue of nodes per level
// (cn/polarCode.py:1573-1628); here the queue is not stored: level l is walked depth-first in the same
// (lexicographic) order, and an interior node is decoded again to get the decision LLRs that rank its children.
// Interior runs repeat a run that already failed its CRC, so they are neither tested nor counted.
//
// Frames are handed out through an atomic counter (a frame that needs many runs must not hold up a fixed share
// of the batch); every wavefront leaves when the counter passes the number of rows.
#include <stdio.h>

#include <algorithm>

#include "decode_common.h"
#include "reg_common.h"
#include "polar_internal.h"

template <typename T> struct PBFParams {
    const T *llr;
    const uint32_t *rows;                 // frame index of work item w, or NULL for w itself
    const unsigned long long *nrows_dev;  // number of work items when rows != NULL
    unsigned long long *work;             // atomic work counter (zero at launch)
    const uint32_t *ops;                  // every-leaf schedule (16-leaf nodes as single operations when N >= 32)
    const uint32_t *A;
    const uint32_t *crcS;                 // NULL: no CRC test (flip-list mode)
    const int32_t *S0;                    // level-0 candidates (critical set)
    const int32_t *crit_off;              // CSR over code positions: candidates after a flip at position i
    const int32_t *crit_idx;
    const double *sqrt_mu;
    const int32_t *flips_in;              // [B x 3] fixed flip lists (flip-list mode), or NULL
    uint32_t *info_hat;                   // [B x Kw]
    uint32_t *u_hat;                      // [B x Nw] or NULL
    T *dec_llr;                           // [B x N] or NULL
    int32_t *runs;                        // [B] or NULL
    uint8_t *crc_ok;                      // [B] or NULL
    int32_t *last_seq;                    // [B x 3] or NULL
    int64_t B;
    int N, n, Nw, K, Kw, nS0, maxS, max_level;
};

__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
    for (int off = 1; off < 64; off <<= 1) v ^= (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}

template <typename T> __global__ void __launch_bounds__(64) pbf_decode_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int N = p.N, n = p.n, Nw = p.Nw;
    T *lv = reinterpret_cast<T *>(smem);                       // N: LLR tree below the channel level
    T *dl = lv + N;                                            // N: decision LLRs of the last run
    double *Mv = reinterpret_cast<double *>(dl + N);           // maxS: ranking keys
    uint32_t *p0 = reinterpret_cast<uint32_t *>(Mv + p.maxS);  // 2 Nw: partial sums of left children
    uint32_t *p1 = p0 + 2 * Nw;                                // 2 Nw: partial sums of right children
    uint32_t *iw = p1 + 2 * Nw;                                // Kw: information words of the last tested run
    uint32_t *misc = iw + p.Kw;                                // 1: work item broadcast
    uint16_t *order = reinterpret_cast<uint16_t *>(misc + 1);  // 3 x maxS: ranked candidates per level
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (lane & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }

    for (;;) {
        if (lane == 0) misc[0] = (uint32_t)atomicAdd(p.work, 1ull);
        wave_sync();
        const uint64_t w = misc[0];
        wave_sync();
        const uint64_t limit = p.rows ? *p.nrows_dev : (uint64_t)p.B;
        if (w >= limit) break;
        const int64_t f = p.rows ? (int64_t)p.rows[w] : (int64_t)w;
        const T *chan = p.llr + f * N;

        // one SC run with decisions inverted at fl0..fl2; `tested`: extract the information word and test the CRC
        auto run = [&](int fl0, int fl1, int fl2, bool tested) -> bool {
            fl0 = __builtin_amdgcn_readfirstlane(fl0);
            fl1 = __builtin_amdgcn_readfirstlane(fl1);
            fl2 = __builtin_amdgcn_readfirstlane(fl2);
            uint32_t nar0 = 0, nar1 = 0;
            int blk = 0;  // 16-leaf nodes come in code order
            uint32_t next_op = ops[0];
            for (int pc = 0;; pc++) {
                const uint32_t op = next_op;
                const uint32_t type = OP_TYPE(op);
                if (type == OP_END) break;
                next_op = ops[pc + 1];  // fetched while this operation runs
                const int lev = (int)OP_LEVEL(op);
                const int plane = (int)OP_PLANE(op);
                switch (type) {
                case OP_F: {
                    const int wd = 1 << (lev - 1);
                    const T *src = (lev == n) ? chan : lv + (1 << lev);
                    for (int k = lane; k < wd; k += 64) lv[wd + k] = f_minsum(src[k], src[k + wd]);
                    break;
                }
                case OP_G: {
                    const int wd = 1 << (lev - 1);
                    const T *src = (lev == n) ? chan : lv + (1 << lev);
                    if (lev - 1 >= 5) {
                        const uint32_t *bits = p0 + (1 << (lev - 6));
                        for (int k = lane; k < wd; k += 64) {
                            T a = src[k], b = src[k + wd];
                            uint32_t bit = (bits[k >> 5] >> (k & 31)) & 1u;
                            lv[wd + k] = bit ? b - a : b + a;
                        }
                    } else {
                        for (int k = lane; k < wd; k += 64) {
                            T a = src[k], b = src[k + wd];
                            uint32_t bit = (nar0 >> (wd + k)) & 1u;
                            lv[wd + k] = bit ? b - a : b + a;
                        }
                    }
                    break;
                }
                case OP_LEAF_INFO:
                case OP_LEAF_FROZEN: {
                    const T v = lv[1];
                    const int i = (int)OP_LEAFIDX(op);
                    uint32_t bit = (type == OP_LEAF_INFO && v < (T)0) ? 1u : 0u;
                    if (i == fl0 || i == fl1 || i == fl2) bit ^= 1u;  // the sequenceU hook, frozen positions included
                    if (plane) nar1 = (nar1 & ~2u) | (bit << 1);
                    else nar0 = (nar0 & ~2u) | (bit << 1);
                    if (lane == 0) dl[i] = v;
                    break;
                }
                case OP_SUB16: {
                    // one LLR per lane of a 16-lane row (the four rows do the same work); lane e ends up with bit e
                    const T v = lv[16 + (lane & 15)];
                    const uint32_t x = sub_node_flip<16, 0>(v, OP_LEAFIDX(op) & 0xFFFFu, rc, 16 * blk, fl0, fl1, fl2, dl, lane == 0);
                    blk++;
                    const uint32_t bits = (uint32_t)__ballot(x != 0u) & 0xFFFFu;
                    if (plane) nar1 = (nar1 & 0x0000FFFFu) | (bits << 16);
                    else nar0 = (nar0 & 0x0000FFFFu) | (bits << 16);
                    break;
                }
                case OP_COMBINE: {
                    const int cl = lev - 1;
                    if (cl >= 5) {
                        const int nwc = 1 << (cl - 5);
                        const uint32_t *s0 = p0 + nwc, *s1 = p1 + nwc;
                        uint32_t *dst = (plane ? p1 : p0) + 2 * nwc;
                        for (int j = lane; j < nwc; j += 64) {
                            uint32_t a = s0[j], b = s1[j];
                            dst[j] = a ^ b;
                            dst[nwc + j] = b;
                        }
                    } else if (cl == 4) {
                        uint32_t a = nar0 >> 16, b = nar1 >> 16;
                        uint32_t v = (a ^ b) | (b << 16);
                        if (lane == 0) (plane ? p1 : p0)[1] = v;
                    } else {
                        const int hw = 1 << cl, wd = 2 * hw;
                        const uint32_t hm = (1u << hw) - 1u;
                        uint32_t a = (nar0 >> hw) & hm, b = (nar1 >> hw) & hm;
                        uint32_t v = (a ^ b) | (b << hw);
                        const uint32_t m = ((1u << wd) - 1u) << wd;
                        if (plane) nar1 = (nar1 & ~m) | (v << wd);
                        else nar0 = (nar0 & ~m) | (v << wd);
                    }
                    break;
                }
                default: break;
                }
                wave_sync();
            }
            if (!tested) return false;
            // u_hat = x_hat F^{(x)n} from the root partial sums, then the information word and its CRC remainder
            uint32_t *uw = p1;
            if (n >= 5) {
                for (int j = lane; j < Nw; j += 64) uw[j] = butterfly_word(p0[Nw + j], n);
                for (int s = 5; s < n; s++) {
                    wave_sync();
                    const int d = 1 << (s - 5);
                    for (int j = lane; j < Nw; j += 64)
                        if (!(j & d)) uw[j] ^= uw[j + d];
                }
            } else {
                uint32_t x = (nar0 >> N) & ((1u << N) - 1u);
                if (lane == 0) uw[0] = butterfly_word(x, n);
            }
            wave_sync();
            uint32_t syn = 0u;
            pack_bits_group<64>(
                p.K, lane, 0,
                [&](int k) {
                    const uint32_t pos = p.A[k];
                    const bool bit = ((uw[pos >> 5] >> (pos & 31)) & 1u) != 0u;
                    if (p.crcS && bit) syn ^= p.crcS[k];
                    return bit;
                },
                [&](int wi, uint32_t v) {
                    if (lane == 0) iw[wi] = v;
                });
            if (p.K == 0 && lane == 0) iw[0] = 0u;
            syn = wave_xor(syn);
            if (p.u_hat)
                for (int j = lane; j < Nw; j += 64) p.u_hat[f * Nw + j] = uw[j];
            wave_sync();
            return syn == 0u;
        };

        // candidates `src[0..cnt)` ranked by |decision LLR| / sqrt(mu), ascending, equal keys in list order
        auto rank = [&](const int32_t *src, int cnt, uint16_t *dst) {
            for (int j = lane; j < cnt; j += 64) {
                const int i = src[j];
                Mv[j] = fabs((double)dl[i] / p.sqrt_mu[i]);
            }
            wave_sync();
            for (int j = lane; j < cnt; j += 64) {
                const double mj = Mv[j];
                int r = 0;
                for (int i = 0; i < cnt; i++) {
                    const double mi = Mv[i];
                    r += (mi < mj || (mi == mj && i < j)) ? 1 : 0;
                }
                dst[r] = (uint16_t)src[j];
            }
            wave_sync();
        };

        int fl[3] = {-1, -1, -1};
        int last[3] = {-1, -1, -1};
        int nruns = 1;
        bool ok;
        if (p.flips_in) {
            fl[0] = p.flips_in[f * 3];
            fl[1] = p.flips_in[f * 3 + 1];
            fl[2] = p.flips_in[f * 3 + 2];
            ok = run(fl[0], fl[1], fl[2], true);
        } else {
            ok = run(-1, -1, -1, true);
            if (!ok && p.max_level > 0) {
                int cnt[3] = {p.nS0, 0, 0}, pos[3] = {0, 0, 0};
                rank(p.S0, p.nS0, order);
                for (int L = 0; L < p.max_level && !ok; L++) {
                    int d = 0;
                    pos[0] = 0;
                    for (;;) {
                        if (pos[d] >= cnt[d]) {
                            if (d == 0) break;
                            d--;
                            pos[d]++;
                            continue;
                        }
                        fl[d] = order[d * p.maxS + pos[d]];
                        for (int t = d + 1; t < 3; t++) fl[t] = -1;
                        if (d == L) {
                            ok = run(fl[0], fl[1], fl[2], true);
                            nruns++;
                            last[0] = fl[0];
                            last[1] = fl[1];
                            last[2] = fl[2];
                            if (ok) break;
                            pos[d]++;
                        } else {
                            run(fl[0], fl[1], fl[2], false);
                            const int b = p.crit_off[fl[d]];
                            cnt[d + 1] = p.crit_off[fl[d] + 1] - b;
                            rank(p.crit_idx + b, cnt[d + 1], order + (d + 1) * p.maxS);
                            d++;
                            pos[d] = 0;
                        }
                    }
                }
            }
        }
        // results of the last tested run
        for (int j = lane; j < p.Kw; j += 64) p.info_hat[f * p.Kw + j] = iw[j];
        if (p.dec_llr)
            for (int j = lane; j < N; j += 64) p.dec_llr[f * N + j] = dl[j];
        if (lane == 0) {
            if (p.runs) p.runs[f] = nruns;
            if (p.crc_ok) p.crc_ok[f] = ok ? 1 : 0;
            if (p.last_seq) {
                p.last_seq[f * 3] = last[0];
                p.last_seq[f * 3 + 1] = last[1];
                p.last_seq[f * 3 + 2] = last[2];
            }
        }
        wave_sync();
    }
}

// CRC test of decoded information words: frames that fail are appended to `rows` (any order) for the search
// kernel, all frames get the single-run defaults.  One frame per lane; the remainder table sits in LDS.
__global__ void __launch_bounds__(256) crc_filter_kernel(const uint32_t *__restrict__ info_hat, const uint32_t *__restrict__ crcS,
                                                        uint32_t *__restrict__ rows, unsigned long long *__restrict__ nrows,
                                                        int32_t *__restrict__ runs, uint8_t *__restrict__ crc_ok,
                                                        int32_t *__restrict__ last_seq, int64_t B, int K, int Kw)
{
    extern __shared__ uint32_t tab[];
    for (int k = threadIdx.x; k < K; k += 256) tab[k] = crcS[k];
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool fail = false;
    if (f < B) {
        uint32_t syn = 0u;
        for (int w = 0; w < Kw; w++) {
            uint32_t v = info_hat[f * Kw + w];
            while (v) {
                syn ^= tab[32 * w + __ffs(v) - 1];
                v &= v - 1;
            }
        }
        fail = syn != 0u;
        if (runs) runs[f] = 1;
        if (crc_ok) crc_ok[f] = fail ? 0 : 1;
        if (last_seq) last_seq[f * 3] = last_seq[f * 3 + 1] = last_seq[f * 3 + 2] = -1;
    }
    const uint64_t b = __ballot(fail);
    if (b) {
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == __ffsll((unsigned long long)b) - 1) base = atomicAdd(nrows, (unsigned long long)__popcll(b));
        base = __shfl(base, __ffsll((unsigned long long)b) - 1, 64);
        if (fail) rows[base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)f;
    }
}

template <typename T>
static int launch_pbf_t(polar_code *c, const void *llr, const uint32_t *rows, const unsigned long long *nrows_dev,
                        unsigned long long *work, const int32_t *flips_in, int max_level, bool crc, uint32_t *info_hat,
                        uint32_t *u_hat, void *dec_llr, int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B)
{
    polar_ctx *ctx = c->ctx;
    PBFParams<T> p;
    p.llr = (const T *)llr;
    p.rows = rows;
    p.nrows_dev = nrows_dev;
    p.work = work;
    p.ops = c->d_sched[0];
    p.A = c->d_A;
    p.crcS = crc ? c->d_crcS : nullptr;
    p.S0 = c->d_S0;
    p.crit_off = c->d_crit_off;
    p.crit_idx = c->d_crit_idx;
    p.sqrt_mu = c->d_sqrt_mu;
    p.flips_in = flips_in;
    p.info_hat = info_hat;
    p.u_hat = u_hat;
    p.dec_llr = (T *)dec_llr;
    p.runs = runs;
    p.crc_ok = crc_ok;
    p.last_seq = last_seq;
    p.B = B;
    p.N = c->N;
    p.n = c->n;
    p.Nw = c->Nw;
    p.K = c->K;
    p.Kw = c->Kw;
    p.nS0 = c->nS0;
    p.maxS = std::max(c->maxS, 1);
    p.max_level = max_level;
    size_t lds = (size_t)2 * c->N * sizeof(T) + (size_t)p.maxS * 8 + (size_t)4 * c->Nw * 4 + (size_t)c->Kw * 4 + 4 + (size_t)3 * p.maxS * 2;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) {
        polar_set_error("flip decoder: N=%d needs %zu B of LDS per wavefront (limit 163840)", c->N, lds);
        return POLAR_EUNSUPPORTED;
    }
    auto kern = pbf_decode_kernel<T>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds));
    if (occ < 1) occ = 1;
    int64_t grid = std::min<int64_t>(B, (int64_t)ctx->num_cus * occ);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, p, (const uint32_t *)c->d_sched[c->n >= 5 ? 2 : 0]);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "pbf_decode_kernel<%s>/T%d", sizeof(T) == 4 ? "float" : "double", max_level);
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = (int)lds;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

int launch_pbf(polar_code *c, const void *llr, int dtype, const uint32_t *rows, const unsigned long long *nrows_dev,
               unsigned long long *work, const int32_t *flips_in, int max_level, bool crc, uint32_t *info_hat, uint32_t *u_hat,
               void *dec_llr, int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B)
{
    if (B == 0) return POLAR_OK;
    if (c->n > 12) {
        polar_set_error("flip decoder: N=%d exceeds the supported maximum of 4096", c->N);
        return POLAR_EUNSUPPORTED;
    }
    if (dtype == POLAR_F32)
        return launch_pbf_t<float>(c, llr, rows, nrows_dev, work, flips_in, max_level, crc, info_hat, u_hat, dec_llr, runs, crc_ok,
                                   last_seq, B);
    return launch_pbf_t<double>(c, llr, rows, nrows_dev, work, flips_in, max_level, crc, info_hat, u_hat, dec_llr, runs, crc_ok,
                                last_seq, B);
}

int launch_crc_filter(polar_code *c, const uint32_t *info_hat, uint32_t *rows, unsigned long long *nrows, int32_t *runs,
                      uint8_t *crc_ok, int32_t *last_seq, int64_t B)
{
    if (B == 0) return POLAR_OK;
    hipLaunchKernelGGL(crc_filter_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), (size_t)std::max(c->K, 1) * 4, c->ctx->stream,
                       info_hat, (const uint32_t *)c->d_crcS, rows, nrows, runs, crc_ok, last_seq, B, c->K, c->Kw);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}
