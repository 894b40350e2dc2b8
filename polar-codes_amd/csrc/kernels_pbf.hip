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
//
// Two simultaneous flips (max_level = 2) cost 10^2 .. 10^4 runs for the few frames that need them, so one wavefront
// per frame would leave the launch waiting for its slowest frame.  That search is split instead: the kernel above
// runs the single-flip level and writes, for every frame still failing, its ranked first-level candidates to a
// record in global memory; pbf_pairs_kernel then hands out (frame, first flip) pairs, lowest first flip first, and
// every such pair is an independent piece of at most 1 + maxS runs.  The reference stops at the first success in
// lexicographic order of (first flip's rank, second flip's rank); here the smallest successful pair wins through a
// 64-bit atomicMin per frame, pairs that can no longer win are skipped, and pbf_pairs_finish_kernel repeats the one
// run whose outputs the frame keeps (the winner, or the last run of the search if nothing passed) and derives the
// run count from the candidate counts, so results do not depend on the order the pairs were worked on.  Three flips
// continue the same way on the frames no pair helped (pbf_triples_kernel), where one frame costs up to 10^5 runs.
#include <stdio.h>

#include <algorithm>

#include "decode_common.h"
#include "reg_common.h"
#include "polar_internal.h"
#include "kernels_pbf.h"

// The state of one wavefront: LDS areas, the frame it works on, one SC run with flips, candidate ranking.
template <typename T> struct PBFWave {
    const PBFParams<T> &p;
    const uint32_t *ops;
    int lane;
    T *lv;            // N: LLR tree below the channel level
    T *dl;            // N: decision LLRs of the last run
    double *Mv;       // maxS: ranking keys
    uint32_t *p0;     // 2 Nw: partial sums of left children
    uint32_t *p1;     // 2 Nw: partial sums of right children
    uint32_t *iw;     // Kw: information words of the last tested run
    uint32_t *misc;   // 1: work item broadcast
    uint16_t *order;  // 3 x maxS: ranked candidates per level
    RowConst rc;
    const T *chan;
    int64_t f;

    __device__ __forceinline__ PBFWave(const PBFParams<T> &pp, const uint32_t *o, unsigned char *smem) : p(pp), ops(o)
    {
        lane = threadIdx.x;
        lv = reinterpret_cast<T *>(smem);
        dl = lv + p.N;
        Mv = reinterpret_cast<double *>(dl + p.N);
        p0 = reinterpret_cast<uint32_t *>(Mv + p.maxS);
        p1 = p0 + 2 * p.Nw;
        iw = p1 + 2 * p.Nw;
        misc = iw + p.Kw;
        order = reinterpret_cast<uint16_t *>(misc + 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            rc.up[i] = (lane & (8 >> i)) != 0;
            rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
            rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
        }
        chan = nullptr;
        f = 0;
    }

    __device__ __forceinline__ void frame(int64_t fr)
    {
        f = fr;
        chan = p.llr + fr * p.N;
    }

    // the next value of an atomic counter, the same in every lane
    __device__ __forceinline__ uint64_t next_item(unsigned long long *ctr)
    {
        if (lane == 0) misc[0] = (uint32_t)atomicAdd(ctr, 1ull);
        wave_sync();
        const uint64_t w = misc[0];
        wave_sync();
        return w;
    }

    // one SC run with decisions inverted at fl0..fl2; `tested`: extract the information word and test the CRC
    __device__ bool run(int fl0, int fl1, int fl2, bool tested)
    {
        const int N = p.N, n = p.n, Nw = p.Nw;
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
    }

    // candidates `src[0..cnt)` ranked by |decision LLR| / sqrt(mu), ascending, equal keys in list order
    __device__ void rank(const int32_t *src, int cnt, uint16_t *dst)
    {
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
    }
};

template <typename T> __global__ void __launch_bounds__(64) pbf_decode_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PBFWave<T> w(p, ops, smem);
    const int lane = w.lane;
    const int N = p.N;
    T *dl = w.dl;
    uint32_t *iw = w.iw;
    uint16_t *order = w.order;
    auto run = [&](int a, int b, int c, bool tested) { return w.run(a, b, c, tested); };
    auto rank = [&](const int32_t *src, int cnt, uint16_t *dst) { w.rank(src, cnt, dst); };

    for (;;) {
        const uint64_t wi = w.next_item(p.work);
        const uint64_t limit = p.rows ? *p.nrows_dev : (uint64_t)p.B;
        if (wi >= limit) break;
        const int64_t f = p.rows ? (int64_t)p.rows[wi] : (int64_t)wi;
        w.frame(f);

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
        if (p.rows2 && !ok) {
            // still failing after every single flip: hand the frame and its first-level ranking to the pair search
            if (lane == 0) w.misc[0] = (uint32_t)atomicAdd(p.nrows2, 1ull);
            wave_sync();
            const uint32_t slot = w.misc[0];
            for (int j = lane; j < p.nS0; j += 64) p.order_rec[(size_t)slot * p.nS0 + j] = order[j];
            if (lane == 0) {
                p.rows2[slot] = (uint32_t)f;
                p.best[slot] = ~0ull;
                if (p.best3) p.best3[slot] = ~0ull;
            }
        }
        wave_sync();
    }
}

// Second level of the split search.  Work item w = (first flip's rank a = w / frames, frame slot = w % frames): all
// frames' most likely first flips come first, and an item whose frame already has a success at a lower rank is
// dropped without a run.  An item is one interior run (first flip alone, for the decision LLRs that rank the second
// flips) and up to cnt tested runs.
template <typename T> __global__ void __launch_bounds__(64) pbf_pairs_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PBFWave<T> w(p, ops, smem);
    const uint64_t n2 = *p.nrows2;
    const uint64_t items = n2 * (uint64_t)p.nS0;
    for (;;) {
        const uint64_t wi = w.next_item(p.work);
        if (wi >= items) break;
        const uint32_t a = (uint32_t)(wi / n2), slot = (uint32_t)(wi % n2);
        if ((load_best(p.best + slot) >> 48) < a) continue;
        const int fl0 = p.order_rec[(size_t)slot * p.nS0 + a];
        const int b0 = p.crit_off[fl0];
        const int cnt = p.crit_off[fl0 + 1] - b0;
        if (cnt == 0) continue;
        w.frame((int64_t)p.rows2[slot]);
        w.run(fl0, -1, -1, false);
        w.rank(p.crit_idx + b0, cnt, w.order);
        for (int b = 0; b < cnt; b++) {
            const unsigned long long here = ((unsigned long long)a << 16) | (unsigned long long)b;
            if ((load_best(p.best + slot) >> 32) < here) break;
            const int fl1 = w.order[b];
            if (w.run(fl0, fl1, -1, true)) {
                if (w.lane == 0) atomicMin(p.best + slot, (here << 32) | ((unsigned long long)fl0 << 16) | (unsigned long long)fl1);
                break;
            }
        }
    }
}

// Outputs of the frames the pair search worked on: the winning pair's run, or - nothing passed - the last tested run
// of the search in the reference's order (last first flip that has second flips, its last-ranked second flip); the
// number of tested runs follows from the candidate counts.  One wavefront per frame.
template <typename T> __global__ void __launch_bounds__(64) pbf_pairs_finish_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PBFWave<T> w(p, ops, smem);
    const int lane = w.lane;
    const uint64_t n2 = *p.nrows2;
    for (uint64_t slot = blockIdx.x; slot < n2; slot += gridDim.x) {
        const unsigned long long best = p.best[slot];
        const bool found = best != ~0ull;
        const int a_win = (int)(best >> 48), b_win = (int)((best >> 32) & 0xFFFFu);
        const uint16_t *rec = p.order_rec + (size_t)slot * p.nS0;
        // tested runs before the winner's first flip (or of the whole level), and the last first flip with second flips
        int before = 0, a_last = -1;
        const int upto = found ? a_win : p.nS0;
        for (int j = lane; j < p.nS0; j += 64) {
            const int fl = rec[j];
            const int cnt = p.crit_off[fl + 1] - p.crit_off[fl];
            if (j < upto) before += cnt;
            if (cnt > 0) a_last = j;
        }
        for (int off = 1; off < 64; off <<= 1) {
            before += __shfl_xor(before, off, 64);
            a_last = max(a_last, __shfl_xor(a_last, off, 64));
        }
        if (!found && a_last < 0) continue;  // no pair exists: the single-flip level's outputs stand
        const int64_t f = (int64_t)p.rows2[slot];
        w.frame(f);
        int fl0, fl1;
        if (found) {
            fl0 = (int)((best >> 16) & 0xFFFFu);
            fl1 = (int)(best & 0xFFFFu);
        } else {
            fl0 = rec[a_last];
            const int b0 = p.crit_off[fl0];
            const int cnt = p.crit_off[fl0 + 1] - b0;
            w.run(fl0, -1, -1, false);
            w.rank(p.crit_idx + b0, cnt, w.order);
            fl1 = w.order[cnt - 1];
        }
        w.run(fl0, fl1, -1, true);
        for (int j = lane; j < p.Kw; j += 64) p.info_hat[f * p.Kw + j] = w.iw[j];
        if (lane == 0) {
            if (p.runs) p.runs[f] = 1 + p.nS0 + before + (found ? b_win + 1 : 0);
            if (p.crc_ok) p.crc_ok[f] = found ? 1 : 0;
            if (p.last_seq) {
                p.last_seq[f * 3] = fl0;
                p.last_seq[f * 3 + 1] = fl1;
                p.last_seq[f * 3 + 2] = -1;
            }
        }
        wave_sync();
    }
}

// Third level of the split search, on the frames no pair helped.  Work item = (first flip's rank a, frame): the run with the
// first flip alone ranks the second flips; for each of them in rank order the run with both ranks the third flips, which are
// tested in rank order.  The reference walks level 3's nodes in the order level 2 appended them (cn/polarCode.py:1596-1626),
// i.e. the lexicographic order of the three ranks, and stops at the first success; here the smallest successful triple of
// ranks (20 bits each in a 64-bit word, all ones = none) wins through an atomicMin per frame.
template <typename T> __global__ void __launch_bounds__(64) pbf_triples_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PBFWave<T> w(p, ops, smem);
    const uint64_t n2 = *p.nrows2;
    const uint64_t items = n2 * (uint64_t)p.nS0;
    uint16_t *order1 = w.order, *order2 = w.order + p.maxS;
    for (;;) {
        const uint64_t wi = w.next_item(p.work);
        if (wi >= items) break;
        const uint32_t a = (uint32_t)(wi / n2), slot = (uint32_t)(wi % n2);
        if (load_best(p.best + slot) != ~0ull) continue;  // a pair passed
        if ((load_best(p.best3 + slot) >> 40) < a) continue;
        const int fl0 = p.order_rec[(size_t)slot * p.nS0 + a];
        const int b0 = p.crit_off[fl0];
        const int cnt1 = p.crit_off[fl0 + 1] - b0;
        if (cnt1 == 0 || p.crit_sum[fl0] == 0) continue;
        w.frame((int64_t)p.rows2[slot]);
        w.run(fl0, -1, -1, false);
        w.rank(p.crit_idx + b0, cnt1, order1);
        bool done = false;
        for (int b = 0; b < cnt1 && !done; b++) {
            if ((load_best(p.best3 + slot) >> 20) < (((unsigned long long)a << 20) | (unsigned long long)b)) break;
            const int fl1 = order1[b];
            const int c0 = p.crit_off[fl1];
            const int cnt2 = p.crit_off[fl1 + 1] - c0;
            if (cnt2 == 0) continue;
            w.run(fl0, fl1, -1, false);
            w.rank(p.crit_idx + c0, cnt2, order2);
            for (int c = 0; c < cnt2; c++) {
                const unsigned long long here = ((unsigned long long)a << 40) | ((unsigned long long)b << 20) | (unsigned long long)c;
                if (load_best(p.best3 + slot) < here) {
                    done = true;
                    break;
                }
                if (w.run(fl0, fl1, (int)order2[c], true)) {
                    if (w.lane == 0) atomicMin(p.best3 + slot, here);
                    done = true;
                    break;
                }
            }
        }
    }
}

// Outputs of the frames the pair and triple searches worked on: the winner's run, or - nothing passed - the last tested run
// of the whole search in the reference's order; the number of tested runs follows from the candidate counts (crit_sum: the
// candidates two levels below a flip).  One wavefront per frame.
template <typename T> __global__ void __launch_bounds__(64) pbf_triples_finish_kernel(PBFParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PBFWave<T> w(p, ops, smem);
    const int lane = w.lane;
    const uint64_t n2 = *p.nrows2;
    uint16_t *order1 = w.order, *order2 = w.order + p.maxS;
    auto cnt_of = [&](int fl) { return p.crit_off[fl + 1] - p.crit_off[fl]; };
    for (uint64_t slot = blockIdx.x; slot < n2; slot += gridDim.x) {
        const unsigned long long best = p.best[slot], best3 = p.best3[slot];
        const bool found2 = best != ~0ull, found3 = !found2 && best3 != ~0ull;
        const uint16_t *rec = p.order_rec + (size_t)slot * p.nS0;
        const int a2 = (int)(best >> 48), b2 = (int)((best >> 32) & 0xFFFFu);
        const int a3 = (int)(best3 >> 40), b3 = (int)((best3 >> 20) & 0xFFFFFu), c3 = (int)(best3 & 0xFFFFFu);
        const int upto2 = found2 ? a2 : p.nS0, upto3 = found2 ? 0 : found3 ? a3 : p.nS0;
        int before2 = 0, before3 = 0, last2 = -1, last3 = -1;
        for (int j = lane; j < p.nS0; j += 64) {
            const int fl = rec[j];
            const int cnt = cnt_of(fl), below = p.crit_sum[fl];
            if (j < upto2) before2 += cnt;
            if (j < upto3) before3 += below;
            if (cnt > 0) last2 = j;
            if (below > 0) last3 = j;
        }
        for (int off = 1; off < 64; off <<= 1) {
            before2 += __shfl_xor(before2, off, 64);
            before3 += __shfl_xor(before3, off, 64);
            last2 = max(last2, __shfl_xor(last2, off, 64));
            last3 = max(last3, __shfl_xor(last3, off, 64));
        }
        if (!found2 && !found3 && last2 < 0) continue;  // no pair exists (so no triple either): the single-flip level's outputs stand
        const int64_t f = (int64_t)p.rows2[slot];
        w.frame(f);
        int fl0, fl1, fl2 = -1, nruns = 1 + p.nS0 + before2;
        if (found2) {
            fl0 = (int)((best >> 16) & 0xFFFFu);
            fl1 = (int)(best & 0xFFFFu);
            nruns += b2 + 1;
        } else if (found3 || last3 >= 0) {
            fl0 = rec[found3 ? a3 : last3];
            const int cnt1 = cnt_of(fl0);
            w.run(fl0, -1, -1, false);
            w.rank(p.crit_idx + p.crit_off[fl0], cnt1, order1);
            nruns += before3;
            int bsel;
            if (found3) {
                bsel = b3;
                int s = 0;
                for (int j = lane; j < b3; j += 64) s += cnt_of((int)order1[j]);
                for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
                nruns += s + c3 + 1;
            } else {
                bsel = -1;
                for (int j = lane; j < cnt1; j += 64)
                    if (cnt_of((int)order1[j]) > 0) bsel = j;
                for (int off = 1; off < 64; off <<= 1) bsel = max(bsel, __shfl_xor(bsel, off, 64));
            }
            fl1 = order1[bsel];
            const int cnt2 = cnt_of(fl1);
            w.run(fl0, fl1, -1, false);
            w.rank(p.crit_idx + p.crit_off[fl1], cnt2, order2);
            fl2 = order2[found3 ? c3 : cnt2 - 1];
        } else {
            // no triple exists: the last pair of level 2 was the last tested run
            fl0 = rec[last2];
            const int cnt = cnt_of(fl0);
            w.run(fl0, -1, -1, false);
            w.rank(p.crit_idx + p.crit_off[fl0], cnt, order1);
            fl1 = order1[cnt - 1];
        }
        w.run(fl0, fl1, fl2, true);
        for (int j = lane; j < p.Kw; j += 64) p.info_hat[f * p.Kw + j] = w.iw[j];
        if (lane == 0) {
            if (p.runs) p.runs[f] = nruns;
            if (p.crc_ok) p.crc_ok[f] = (found2 || found3) ? 1 : 0;
            if (p.last_seq) {
                p.last_seq[f * 3] = fl0;
                p.last_seq[f * 3 + 1] = fl1;
                p.last_seq[f * 3 + 2] = fl2;
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

bool pbf_reg_supported(const polar_code *c, const uint32_t *rows, const int32_t *flips_in, int max_level, bool crc, const void *u_hat,
                       const void *dec_llr, const PbfSplit *split);
template <typename T> int launch_pbf_reg(polar_code *c, PBFParams<T> &p, int64_t B, const PbfSplit *split, int max_level);

template <typename T>
static int launch_pbf_t(polar_code *c, const void *llr, const uint32_t *rows, const unsigned long long *nrows_dev,
                        unsigned long long *work, const int32_t *flips_in, int max_level, bool crc, uint32_t *info_hat,
                        uint32_t *u_hat, void *dec_llr, int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B,
                        const PbfSplit *split)
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
    p.rows2 = split ? split->rows2 : nullptr;
    p.nrows2 = split ? split->ctr + 1 : nullptr;
    p.order_rec = split ? split->order_rec : nullptr;
    p.best = split ? split->best : nullptr;
    p.best3 = split ? split->best3 : nullptr;
    p.crit_sum = c->d_crit_sum;
    p.B = B;
    p.N = c->N;
    p.n = c->n;
    p.Nw = c->Nw;
    p.K = c->K;
    p.Kw = c->Kw;
    p.nS0 = c->nS0;
    p.maxS = std::max(c->maxS, 1);
    p.max_level = split ? 1 : max_level;
    if (pbf_reg_supported(c, rows, flips_in, max_level, crc, u_hat, dec_llr, split)) return launch_pbf_reg<T>(c, p, B, split, max_level);
    size_t lds = (size_t)2 * c->N * sizeof(T) + (size_t)p.maxS * 8 + (size_t)4 * c->Nw * 4 + (size_t)c->Kw * 4 + 4 + (size_t)3 * p.maxS * 2;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) {
        polar_set_error("flip decoder: N=%d needs %zu B of LDS per wavefront (limit 163840)", c->N, lds);
        return POLAR_EUNSUPPORTED;
    }
    const uint32_t *ops = (const uint32_t *)c->d_sched[c->n >= 5 ? 2 : 0];
    auto kern = pbf_decode_kernel<T>;
    auto kern2 = pbf_pairs_kernel<T>;
    auto kern3 = pbf_pairs_finish_kernel<T>;
    auto kern4 = pbf_triples_kernel<T>;
    auto kern5 = pbf_triples_finish_kernel<T>;
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (split) {
            HIP_TRY(hipFuncSetAttribute((const void *)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)kern3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)kern4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)kern5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
    }
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds));
    if (occ < 1) occ = 1;
    int64_t grid = std::min<int64_t>(B, (int64_t)ctx->num_cus * occ);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, p, ops);
    HIP_TRY(hipGetLastError());
    if (split) {
        // the pair search: its number of work items is only known on the device, so the grid is the resident set
        int occ2 = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, kern2, 64, lds));
        if (occ2 < 1) occ2 = 1;
        p.work = split->ctr;
        p.max_level = 2;
        hipLaunchKernelGGL(kern2, dim3((unsigned)(ctx->num_cus * occ2)), dim3(64), lds, ctx->stream, p, ops);
        HIP_TRY(hipGetLastError());
        if (max_level == 3) {
            p.work = split->ctr + 2;
            p.max_level = 3;
            hipLaunchKernelGGL(kern4, dim3((unsigned)(ctx->num_cus * occ2)), dim3(64), lds, ctx->stream, p, ops);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(kern5, dim3((unsigned)std::min<int64_t>(B, (int64_t)ctx->num_cus * occ2)), dim3(64), lds, ctx->stream, p, ops);
        } else {
            hipLaunchKernelGGL(kern3, dim3((unsigned)std::min<int64_t>(B, (int64_t)ctx->num_cus * occ2)), dim3(64), lds, ctx->stream, p, ops);
        }
        HIP_TRY(hipGetLastError());
    }
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "pbf_decode_kernel<%s>/T%d%s", sizeof(T) == 4 ? "float" : "double", max_level,
             !split ? "" : max_level == 3 ? "+pairs+triples" : "+pairs");
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = (int)lds;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

int launch_pbf(polar_code *c, const void *llr, int dtype, const uint32_t *rows, const unsigned long long *nrows_dev,
               unsigned long long *work, const int32_t *flips_in, int max_level, bool crc, uint32_t *info_hat, uint32_t *u_hat,
               void *dec_llr, int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B, const PbfSplit *split)
{
    if (B == 0) return POLAR_OK;
    if (c->n > 12) {
        polar_set_error("flip decoder: N=%d exceeds the supported maximum of 4096", c->N);
        return POLAR_EUNSUPPORTED;
    }
    if (split && ((max_level != 2 && !(max_level == 3 && split->best3)) || flips_in || u_hat || dec_llr || !crc)) {
        polar_set_error("flip decoder: the split search serves two- and three-flip CRC searches without side outputs only");
        return POLAR_EINVAL;
    }
    if (dtype == POLAR_F32)
        return launch_pbf_t<float>(c, llr, rows, nrows_dev, work, flips_in, max_level, crc, info_hat, u_hat, dec_llr, runs, crc_ok,
                                   last_seq, B, split);
    return launch_pbf_t<double>(c, llr, rows, nrows_dev, work, flips_in, max_level, crc, info_hat, u_hat, dec_llr, runs, crc_ok,
                                last_seq, B, split);
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
