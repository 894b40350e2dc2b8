// The bit-flipping search with its SC runs in registers (N = 32 .. 1024, fp32 or fp64, one to three flips).
//
// A run is the every-leaf walk of sc_reg_full_kernel -- sixteen lanes per run, element k of a tree level in lane
// k mod 16 -- with the flip hook of scDecoder(sequenceU) (cn/polarCode.py:1555-1558) in the 16-leaf nodes.  The four
// rows of a wavefront decode the SAME frame with four different flip sets: the search tries its ranked candidates four
// at a time, and the first one in rank order whose CRC holds wins, exactly the candidate the sequential search of
// progressiveBitFlipping (cn/polarCode.py:1573-1628) stops at; the SC-run count it reports is computed from the
// winner's rank, so the up to three runs made beyond it do not show.  Runs that rank candidates (the unflipped run,
// and with two flips the run with the first flip alone) are the same in all four rows; row 0 leaves its decision LLRs
// in LDS for the ranking.
//
// Five kernels share the code: the single-flip level on the frames that failed their CRC (and the hand-over of
// frames that fail it entirely to the pair search), the pair search over (first flip's rank, frame) work items, the
// pass that writes the outputs of the frames the pair search worked on, and for three flips the triple search over
// the same work items on the frames no pair helped, with its own output pass.  Bookkeeping, work distribution and
// results are those of kernels_pbf.hip, which keeps the cases this file does not cover.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "decode_common.h"
#include "reg_common.h"
#include "kernels_sc_static.h"
#include "kernels_sc_node.h"
#include "polar_internal.h"
#include "kernels_pbf.h"

namespace {

constexpr int PBF_WAVES = 4;  // wavefronts per workgroup: they share the gather table and the CRC table, nothing else

// the 16-leaf decoder that visits every leaf, inverts the decisions named by the bits of fm and (DL) keeps the decision
// LLR of leaf e in lane e of the row
template <int P, int OFF, bool DL, typename T>
__device__ __forceinline__ uint32_t sub_node_fm(T v, uint32_t mask, uint32_t fm, const RowConst &rc, int sub, T &keep)
{
    if constexpr (P == 1) {
        if constexpr (DL) keep = (sub == OFF) ? v : keep;
        return (pin(v < (T)0 ? 1u : 0u) & mask) ^ ((fm >> OFF) & 1u);  // the hook inverts frozen positions too
    } else {
        constexpr uint32_t HALF = (1u << (P / 2)) - 1u;
        constexpr int I = P == 16 ? 0 : P == 8 ? 1 : P == 4 ? 2 : 3;
        const T o = partner<P>(v);
        const T vl = f_fast(v, o);
        const uint32_t xl = sub_node_fm<P / 2, OFF, DL>(vl, mask & HALF, fm, rc, sub, keep);
        const uint32_t y = xl & rc.lowmask[I];
        const T t = flip_sign(v, y);
        const T vr = t + partner<P>(t);
        const uint32_t xr = sub_node_fm<P / 2, OFF + P / 2, DL>(vr, (mask >> (P / 2)) & HALF, fm, rc, sub, keep);
        return xr ^ y;
    }
}

// node idx of level LEV >= 5: the loop nest of decode_node without short-cuts, flips at code positions fl0 / fl1
// / fl2 (-1: none; they may differ from row to row)
template <int LEV, bool DL, typename Dec>
__device__ __forceinline__ void flip_walk(Dec &d, uint32_t maskv, const typename Dec::value_type *chan_lane, uint32_t idx, int plane,
                                          const RowConst &rc, int fl0, int fl1, int fl2, typename Dec::value_type *dl_lane, int sub)
{
    typedef typename Dec::value_type T;
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
        if (h == 0) d.template step<LEV, false>(chan_lane);
        else d.template step<LEV, true>(chan_lane);
        const uint32_t cidx = 2 * idx + h;
        if constexpr (LEV - 1 == 4) {
            const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(cidx >> 1));
            const uint32_t m16 = (mw >> (16 * (cidx & 1))) & 0xFFFFu;
            const uint32_t fm = ((uint32_t)(fl0 >> 4) == cidx ? 1u << (fl0 & 15) : 0u) | ((uint32_t)(fl1 >> 4) == cidx ? 1u << (fl1 & 15) : 0u) |
                                ((uint32_t)(fl2 >> 4) == cidx ? 1u << (fl2 & 15) : 0u);
            T keep = 0;
            const uint32_t x = sub_node_fm<16, 0, DL>(d.R[1], m16, fm, rc, sub, keep);
            if constexpr (DL) {
                if (dl_lane) dl_lane[16 * cidx] = keep;
            }
            if (h) d.ps1 = (d.ps1 & ~(ps_t)2) | ((ps_t)x << 1);
            else d.ps0 = (d.ps0 & ~(ps_t)2) | ((ps_t)x << 1);
        } else {
            flip_walk<LEV - 1, DL>(d, maskv, chan_lane, cidx, (int)h, rc, fl0, fl1, fl2, dl_lane, sub);
        }
    }
    d.template combine<LEV>(plane);
}

enum { PBF_SINGLES = 0, PBF_PAIRS = 1, PBF_FINISH = 2, PBF_TRIPLES = 3, PBF_FINISH3 = 4 };

// The smallest successful triple a frame has so far: ranks of the first, second and third flip (10 bits each from bit 54,
// 44, 34 down), then the three flipped positions (10 bits each from bit 20, 10, 0); all ones = none.  N <= 1024 here.
__device__ __forceinline__ unsigned long long triple_key(int a, int b, int c, int fl0, int fl1, int fl2)
{
    return ((unsigned long long)a << 54) | ((unsigned long long)b << 44) | ((unsigned long long)c << 34) | ((unsigned long long)fl0 << 20) |
           ((unsigned long long)fl1 << 10) | (unsigned long long)fl2;
}

template <typename T, int NLOG> struct PbfRegWave {
    static constexpr int N = 1 << NLOG;
    const PBFParams<T> &p;
    RegDec<T, NLOG> d;
    RowConst rc;
    uint32_t maskv;
    int lane, grp, sub;
    uint32_t *stage;        // this wavefront's output stage
    const uint32_t *ctab;   // compress / place table of the output stage (workgroup)
    const uint32_t *crc;    // CRC remainder of every information bit (workgroup)
    T *dl;                  // N decision LLRs of row 0's last ranking run
    double *Mv;             // maxS ranking keys
    uint16_t *order;        // maxS ranked candidates
    uint16_t *order2;       // maxS ranked candidates of the level below (three flips)
    uint32_t *misc;
    const T *chan_lane;
    int64_t f;
    uint32_t w_lo, w_hi;    // information words sub / sub + 16 of the row's last tested run

    __device__ __forceinline__ PbfRegWave(const PBFParams<T> &pp) : p(pp) {}

    __device__ __forceinline__ void frame(int64_t fr)
    {
        f = fr;
        chan_lane = p.llr + fr * N + sub;
    }
    __device__ __forceinline__ uint64_t next_item(unsigned long long *ctr)
    {
        if (lane == 0) misc[0] = (uint32_t)atomicAdd(ctr, 1ull);
        wave_sync();
        const uint64_t w = misc[0];
        wave_sync();
        return w;
    }
    // one run per row with flips fl0 (all rows) and fl1 (this row's).  DL: the run's decision LLRs rank the next level's
    // candidates; such a run is the same in every row, and row 0 records them.
    template <bool DL, bool THREE = false> __device__ __forceinline__ void walk(int fl0, int fl1, int fl2 = -1)
    {
        d.ps0 = d.ps1 = 0;
        d.root = 0;
        flip_walk<NLOG, DL>(d, maskv, chan_lane, 0u, 0, rc, fl0, fl1, THREE ? fl2 : -1, DL && grp == 0 ? dl + sub : (T *)nullptr, sub);
        if constexpr (DL) wave_sync();
    }
    __device__ __forceinline__ void run_ranking(int fl0, int fl1 = -1) { walk<true>(fl0, fl1); }
    // information words and CRC of the rows' last runs; true in the lanes of the rows whose CRC holds
    __device__ __forceinline__ bool test()
    {
        sc_reg_words_c<NLOG>(d.root, stage, ctab, lane, rc, w_lo, w_hi);
        uint32_t syn = 0u;
        for (uint32_t v = w_lo; v; v &= v - 1u) syn ^= crc[32 * sub + __ffs((int)v) - 1];
        for (uint32_t v = w_hi; v; v &= v - 1u) syn ^= crc[32 * (sub + 16) + __ffs((int)v) - 1];
        syn ^= dpp_mov_u32<0x128>(syn);
        syn ^= dpp_mov_u32<0x124>(syn);
        syn ^= dpp_mov_u32<0x122>(syn);
        syn ^= dpp_mov_u32<0x121>(syn);
        wave_sync();  // the stage is free again
        return syn == 0u;
    }
    __device__ __forceinline__ bool run_tested(int fl0, int fl1)
    {
        walk<false>(fl0, fl1);
        return test();
    }
    __device__ __forceinline__ bool run_tested3(int fl0, int fl1, int fl2)
    {
        walk<false, true>(fl0, fl1, fl2);
        return test();
    }
    // candidates src[0..cnt) ranked by |decision LLR| / sqrt(mu), ascending, equal keys in list order
    __device__ __forceinline__ void rank(const int32_t *src, int cnt) { rank(src, cnt, order); }
    __device__ __forceinline__ void rank(const int32_t *src, int cnt, uint16_t *dst)
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
    // the information words of row g become the frame's output
    __device__ __forceinline__ void store_words(int g)
    {
        if (grp == g) {
            uint32_t *dst = p.info_hat + f * p.Kw;
            if (sub < p.Kw) dst[sub] = w_lo;
            if (sub + 16 < p.Kw) dst[sub + 16] = w_hi;
        }
    }
    __device__ __forceinline__ void store_side(int nruns, bool ok, int s0, int s1, int s2 = -1)
    {
        if (lane == 0) {
            if (p.runs) p.runs[f] = nruns;
            if (p.crc_ok) p.crc_ok[f] = ok ? 1 : 0;
            if (p.last_seq) {
                p.last_seq[f * 3] = s0;
                p.last_seq[f * 3 + 1] = s1;
                p.last_seq[f * 3 + 2] = s2;
            }
        }
    }
    // four candidates of order[] at a time, each with fl0 as well (-1: single flips): rank of the first that passes, or -1
    // after all cnt failed.  stop(b0): the candidates from rank b0 on cannot matter any more.
    template <typename Stop> __device__ __forceinline__ int try_ranked(int fl0, int cnt, Stop stop, int &flip)
    {
        for (int b0 = 0; b0 < cnt; b0 += 4) {
            if (stop(b0)) return -2;
            const int mine = b0 + grp;
            const int fl = mine < cnt ? (int)order[mine] : -1;
            const bool ok = run_tested(fl0 >= 0 ? fl0 : fl, fl0 >= 0 ? fl : -1) && mine < cnt;
            const uint64_t rows = __ballot(ok);
            if (rows) {
                const int g = (__ffsll((unsigned long long)rows) - 1) >> 4;
                flip = __builtin_amdgcn_readlane(fl, 16 * g);
                return b0 + g;
            }
        }
        return -1;
    }
    // the same one level down: four candidates of order2[] at a time, each together with fl0 and fl1
    template <typename Stop> __device__ __forceinline__ int try_ranked3(int fl0, int fl1, int cnt, Stop stop, int &flip)
    {
        for (int c0 = 0; c0 < cnt; c0 += 4) {
            if (stop(c0)) return -2;
            const int mine = c0 + grp;
            const int fl = mine < cnt ? (int)order2[mine] : -1;
            const bool ok = run_tested3(fl0, fl1, fl) && mine < cnt;
            const uint64_t rows = __ballot(ok);
            if (rows) {
                const int g = (__ffsll((unsigned long long)rows) - 1) >> 4;
                flip = __builtin_amdgcn_readlane(fl, 16 * g);
                return c0 + g;
            }
        }
        return -1;
    }
    __device__ __forceinline__ int crit_count(int fl) const { return p.crit_off[fl + 1] - p.crit_off[fl]; }
    // candidates one level below the flip at position fl, summed over the level in between
    __device__ __forceinline__ int crit_below(int fl) const { return p.crit_sum[fl]; }
    template <typename F> __device__ __forceinline__ int wave_sum(int n, F term)
    {
        int v = 0;
        for (int j = lane; j < n; j += 64) v += term(j);
        for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        return v;
    }
};

template <typename T, int NLOG, int MODE>
__global__ void __launch_bounds__(64 * PBF_WAVES, sizeof(T) == 8 ? 2 : 4) pbf_reg_kernel(PBFParams<T> p, const uint32_t *__restrict__ maskw,
                                                                                      const uint32_t *__restrict__ ctab)
{
    constexpr int N = 1 << NLOG;
    __shared__ __attribute__((aligned(32))) uint32_t stage_s[PBF_WAVES * SC_STAGE_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t ctab_s[256];
    __shared__ uint32_t misc_s[PBF_WAVES];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];  // CRC table, then per wavefront: dl, Mv, order
    uint32_t *crc_s = reinterpret_cast<uint32_t *>(dyn);
    for (int k = threadIdx.x; k < 256; k += 64 * PBF_WAVES) ctab_s[k] = ctab[k];
    for (int k = threadIdx.x; k < 1024; k += 64 * PBF_WAVES) crc_s[k] = k < p.K ? p.crcS[k] : 0u;  // 32 entries per lane and word slot
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PbfRegWave<T, NLOG> w(p);
    w.lane = threadIdx.x & 63;
    w.grp = w.lane >> 4;
    w.sub = w.lane & 15;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w.rc.up[i] = (w.sub & (8 >> i)) != 0;
        w.rc.lowmask[i] = w.rc.up[i] ? 0u : 0xFFFFFFFFu;
        w.rc.low31[i] = w.rc.up[i] ? 0u : 0x80000000u;
    }
    w.maskv = maskw[(w.lane & 31) < (N / 32) ? (w.lane & 31) : 0];
    w.stage = stage_s + wave * SC_STAGE_WORDS;
    w.ctab = ctab_s;
    w.crc = crc_s;
    w.misc = misc_s + wave;
    const size_t ord = ((size_t)p.maxS * 2 + 15) & ~(size_t)15;
    const size_t per_wave = (size_t)N * sizeof(T) + (size_t)p.maxS * 8 + 2 * ord;
    unsigned char *mine = dyn + 4096 + (size_t)wave * per_wave;
    w.dl = reinterpret_cast<T *>(mine);
    w.Mv = reinterpret_cast<double *>(mine + (size_t)N * sizeof(T));
    w.order = reinterpret_cast<uint16_t *>(mine + (size_t)N * sizeof(T) + (size_t)p.maxS * 8);
    w.order2 = reinterpret_cast<uint16_t *>(mine + (size_t)N * sizeof(T) + (size_t)p.maxS * 8 + ord);
    auto never = [](int) { return false; };

    if constexpr (MODE == PBF_SINGLES) {
        for (;;) {
            const uint64_t wi = w.next_item(p.work);
            if (wi >= *p.nrows_dev) break;
            w.frame((int64_t)p.rows[wi]);
            // the unflipped run, every leaf visited: its CRC decides whether anything is searched at all
            w.run_ranking(-1);
            bool ok = w.test();
            ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;  // the four rows are the same run
            int nruns = 1, last = -1, out_row = 0;
            if (!ok && p.max_level > 0 && p.nS0 > 0) {
                w.rank(p.S0, p.nS0);
                int flip = -1;
                const int r = w.try_ranked(-1, p.nS0, never, flip);
                if (r >= 0) {
                    ok = true;
                    nruns = 1 + r + 1;
                    last = flip;
                    out_row = r & 3;
                } else {
                    nruns = 1 + p.nS0;
                    last = w.order[p.nS0 - 1];
                    out_row = (p.nS0 - 1) & 3;
                }
            }
            w.store_words(out_row);
            w.store_side(nruns, ok, last, -1);
            if (p.rows2 && !ok) {
                // still failing after every single flip: hand the frame and its first-level ranking to the pair search
                if (w.lane == 0) w.misc[0] = (uint32_t)atomicAdd(p.nrows2, 1ull);
                wave_sync();
                const uint32_t slot = w.misc[0];
                for (int j = w.lane; j < p.nS0; j += 64) p.order_rec[(size_t)slot * p.nS0 + j] = w.order[j];
                if (w.lane == 0) {
                    p.rows2[slot] = (uint32_t)w.f;
                    p.best[slot] = ~0ull;
                    if (p.best3) p.best3[slot] = ~0ull;
                }
            }
            wave_sync();
        }
    } else if constexpr (MODE == PBF_PAIRS) {
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
            w.run_ranking(fl0);
            w.rank(p.crit_idx + b0, cnt);
            int fl1 = -1;
            const int r = w.try_ranked(
                fl0, cnt, [&](int b) { return (load_best(p.best + slot) >> 32) < (((unsigned long long)a << 16) | (unsigned long long)b); }, fl1);
            if (r >= 0 && w.lane == 0)
                atomicMin(p.best + slot, ((((unsigned long long)a << 16) | (unsigned long long)r) << 32) | ((unsigned long long)fl0 << 16) |
                                             (unsigned long long)fl1);
        }
    } else if constexpr (MODE == PBF_TRIPLES) {
        // Third level, on the frames no pair helped.  Work item = (first flip's rank a, frame): the run with the first flip
        // alone ranks the second flips; for each of them in rank order the run with both ranks the third flips, which are
        // tested four at a time.  The reference stops at the first success in lexicographic order of the three ranks
        // (cn/polarCode.py:1596-1626 walks level 3's nodes in the order level 2 appended them); here the smallest successful
        // triple wins through a 64-bit atomicMin per frame and what can no longer win is skipped.
        const uint64_t n2 = *p.nrows2;
        const uint64_t items = n2 * (uint64_t)p.nS0;
        for (;;) {
            const uint64_t wi = w.next_item(p.work);
            if (wi >= items) break;
            const uint32_t a = (uint32_t)(wi / n2), slot = (uint32_t)(wi % n2);
            if (load_best(p.best + slot) != ~0ull) continue;  // a pair passed
            if ((load_best(p.best3 + slot) >> 54) < a) continue;
            const int fl0 = p.order_rec[(size_t)slot * p.nS0 + a];
            const int cnt1 = w.crit_count(fl0);
            if (cnt1 == 0 || w.crit_below(fl0) == 0) continue;
            w.frame((int64_t)p.rows2[slot]);
            w.run_ranking(fl0);
            w.rank(p.crit_idx + p.crit_off[fl0], cnt1, w.order);
            for (int b = 0; b < cnt1; b++) {
                if ((load_best(p.best3 + slot) >> 44) < (((unsigned long long)a << 10) | (unsigned long long)b)) break;
                const int fl1 = w.order[b];
                const int cnt2 = w.crit_count(fl1);
                if (cnt2 == 0) continue;
                w.run_ranking(fl0, fl1);
                w.rank(p.crit_idx + p.crit_off[fl1], cnt2, w.order2);
                int fl2 = -1;
                const int r = w.try_ranked3(
                    fl0, fl1, cnt2,
                    [&](int c) {
                        return (load_best(p.best3 + slot) >> 34) <
                               (((unsigned long long)a << 20) | ((unsigned long long)b << 10) | (unsigned long long)c);
                    },
                    fl2);
                if (r >= 0) {
                    if (w.lane == 0) atomicMin(p.best3 + slot, triple_key((int)a, b, r, fl0, fl1, fl2));
                    break;
                }
                if (r == -2) break;
            }
        }
    } else if constexpr (MODE == PBF_FINISH3) {
        // Outputs of the frames the pair and triple searches worked on: the winner's run, or -- nothing passed -- the last
        // tested run of the whole search in the reference's order; the number of tested runs follows from the candidate counts.
        const uint64_t n2 = *p.nrows2;
        for (uint64_t slot = (uint64_t)blockIdx.x * PBF_WAVES + wave; slot < n2; slot += (uint64_t)gridDim.x * PBF_WAVES) {
            const unsigned long long best = p.best[slot], best3 = p.best3[slot];
            const bool found2 = best != ~0ull, found3 = !found2 && best3 != ~0ull;
            const uint16_t *rec = p.order_rec + (size_t)slot * p.nS0;
            const int a2 = (int)(best >> 48), b2 = (int)((best >> 32) & 0xFFFFu);
            const int a3 = (int)(best3 >> 54), b3 = (int)((best3 >> 44) & 1023u), c3 = (int)((best3 >> 34) & 1023u);
            // tested runs of level 2 (before the winning pair's first flip, or all of them), of level 3 before the winning
            // triple's first flip (or all of them); the last first flip with pairs / with triples below it
            const int upto2 = found2 ? a2 : p.nS0, upto3 = found2 ? 0 : found3 ? a3 : p.nS0;
            int before2 = 0, before3 = 0, last2 = -1, last3 = -1;
            for (int j = w.lane; j < p.nS0; j += 64) {
                const int fl = rec[j];
                const int cnt = w.crit_count(fl), below = w.crit_below(fl);
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
            w.frame((int64_t)p.rows2[slot]);
            int fl0, fl1, fl2 = -1, nruns = 1 + p.nS0 + before2;
            if (found2) {
                fl0 = (int)((best >> 16) & 0xFFFFu);
                fl1 = (int)(best & 0xFFFFu);
                nruns += b2 + 1;
            } else if (found3 || last3 >= 0) {
                // the second flips of the (winning / last) first flip in rank order: the tested runs under those before the
                // winner, or the last of them that has third flips
                fl0 = found3 ? (int)((best3 >> 20) & 1023u) : (int)rec[last3];
                const int cnt1 = w.crit_count(fl0);
                w.run_ranking(fl0);
                w.rank(p.crit_idx + p.crit_off[fl0], cnt1, w.order);
                nruns += before3;
                if (found3) {
                    fl1 = (int)((best3 >> 10) & 1023u);
                    fl2 = (int)(best3 & 1023u);
                    nruns += w.wave_sum(b3, [&](int j) { return w.crit_count((int)w.order[j]); }) + c3 + 1;
                } else {
                    int bl = -1;
                    for (int j = w.lane; j < cnt1; j += 64)
                        if (w.crit_count((int)w.order[j]) > 0) bl = j;
                    for (int off = 1; off < 64; off <<= 1) bl = max(bl, __shfl_xor(bl, off, 64));
                    fl1 = w.order[bl];
                    const int cnt2 = w.crit_count(fl1);
                    w.run_ranking(fl0, fl1);
                    w.rank(p.crit_idx + p.crit_off[fl1], cnt2, w.order2);
                    fl2 = w.order2[cnt2 - 1];
                }
            } else {
                // no triple exists: the last pair of level 2 was the last tested run
                fl0 = rec[last2];
                const int cnt = w.crit_count(fl0);
                w.run_ranking(fl0);
                w.rank(p.crit_idx + p.crit_off[fl0], cnt, w.order);
                fl1 = w.order[cnt - 1];
            }
            w.run_tested3(fl0, fl1, fl2);  // the same run in every row
            w.store_words(0);
            w.store_side(nruns, found2 || found3, fl0, fl1, fl2);
            wave_sync();
        }
    } else {
        const uint64_t n2 = *p.nrows2;
        for (uint64_t slot = (uint64_t)blockIdx.x * PBF_WAVES + wave; slot < n2; slot += (uint64_t)gridDim.x * PBF_WAVES) {
            const unsigned long long best = p.best[slot];
            const bool found = best != ~0ull;
            const int a_win = (int)(best >> 48), b_win = (int)((best >> 32) & 0xFFFFu);
            const uint16_t *rec = p.order_rec + (size_t)slot * p.nS0;
            // tested runs before the winner's first flip (or of the whole level), and the last first flip with second flips
            int before = 0, a_last = -1;
            const int upto = found ? a_win : p.nS0;
            for (int j = w.lane; j < p.nS0; j += 64) {
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
            w.frame((int64_t)p.rows2[slot]);
            int fl0, fl1;
            if (found) {
                fl0 = (int)((best >> 16) & 0xFFFFu);
                fl1 = (int)(best & 0xFFFFu);
            } else {
                fl0 = rec[a_last];
                const int b0 = p.crit_off[fl0];
                const int cnt = p.crit_off[fl0 + 1] - b0;
                w.run_ranking(fl0);
                w.rank(p.crit_idx + b0, cnt);
                fl1 = w.order[cnt - 1];
            }
            w.run_tested(fl0, fl1);  // the same run in every row
            w.store_words(0);
            w.store_side(1 + p.nS0 + before + (found ? b_win + 1 : 0), found, fl0, fl1);
            wave_sync();
        }
    }
}

template <typename T, int NLOG>
int launch_pbf_reg_t(polar_code *c, PBFParams<T> &p, int64_t B, const PbfSplit *split, int max_level)
{
    polar_ctx *ctx = c->ctx;
    const size_t per_wave = (size_t)c->N * sizeof(T) + (size_t)p.maxS * 8 + 2 * (((size_t)p.maxS * 2 + 15) & ~(size_t)15);
    const size_t lds = 4096 + PBF_WAVES * per_wave;
    auto k0 = pbf_reg_kernel<T, NLOG, PBF_SINGLES>;
    auto k1 = pbf_reg_kernel<T, NLOG, PBF_PAIRS>;
    auto k2 = pbf_reg_kernel<T, NLOG, PBF_FINISH>;
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k0, 64 * PBF_WAVES, lds));
    if (occ < 1) occ = 1;
    const int64_t resident = (int64_t)ctx->num_cus * occ;
    const int64_t grid = std::min<int64_t>((B + PBF_WAVES - 1) / PBF_WAVES, resident);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(k0, dim3((unsigned)grid), dim3(64 * PBF_WAVES), lds, ctx->stream, p, (const uint32_t *)c->d_maskw, (const uint32_t *)c->d_ctab);
    HIP_TRY(hipGetLastError());
    if (split) {
        p.work = split->ctr;
        p.max_level = 2;
        hipLaunchKernelGGL(k1, dim3((unsigned)resident), dim3(64 * PBF_WAVES), lds, ctx->stream, p, (const uint32_t *)c->d_maskw, (const uint32_t *)c->d_ctab);
        HIP_TRY(hipGetLastError());
        if (max_level == 3) {
            auto k3 = pbf_reg_kernel<T, NLOG, PBF_TRIPLES>;
            auto k4 = pbf_reg_kernel<T, NLOG, PBF_FINISH3>;
            p.work = split->ctr + 2;
            p.max_level = 3;
            hipLaunchKernelGGL(k3, dim3((unsigned)resident), dim3(64 * PBF_WAVES), lds, ctx->stream, p, (const uint32_t *)c->d_maskw, (const uint32_t *)c->d_ctab);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k4, dim3((unsigned)grid), dim3(64 * PBF_WAVES), lds, ctx->stream, p, (const uint32_t *)c->d_maskw, (const uint32_t *)c->d_ctab);
        } else {
            hipLaunchKernelGGL(k2, dim3((unsigned)grid), dim3(64 * PBF_WAVES), lds, ctx->stream, p, (const uint32_t *)c->d_maskw, (const uint32_t *)c->d_ctab);
        }
        HIP_TRY(hipGetLastError());
    }
    const float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "pbf_reg_kernel<%s,%d>/T%d%s", sizeof(T) == 4 ? "float" : "double", NLOG, max_level,
             !split ? "" : max_level == 3 ? "+pairs+triples" : "+pairs");
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = (int)lds + 4 * PBF_WAVES * SC_STAGE_WORDS + 1024;
    c->last.block_threads = 64 * PBF_WAVES;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <typename T> int launch_pbf_reg_n(polar_code *c, PBFParams<T> &p, int64_t B, const PbfSplit *split, int max_level)
{
    switch (c->n) {
    case 5: return launch_pbf_reg_t<T, 5>(c, p, B, split, max_level);
    case 6: return launch_pbf_reg_t<T, 6>(c, p, B, split, max_level);
    case 7: return launch_pbf_reg_t<T, 7>(c, p, B, split, max_level);
    case 8: return launch_pbf_reg_t<T, 8>(c, p, B, split, max_level);
    case 9: return launch_pbf_reg_t<T, 9>(c, p, B, split, max_level);
    case 10: return launch_pbf_reg_t<T, 10>(c, p, B, split, max_level);
    default: polar_set_error("register flip search: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}

}  // namespace

// The register search serves the CRC-driven search over a list of failed frames with one flip, or two or three flips in pieces.
bool pbf_reg_supported(const polar_code *c, const uint32_t *rows, const int32_t *flips_in, int max_level, bool crc, const void *u_hat,
                       const void *dec_llr, const PbfSplit *split)
{
    // the table of host_tables.cpp says what this kernel serves; it works on the list of frames that failed their CRC, so a
    // call without that list counts as a call with a flip list
    DispatchQuery q{DQ_OP_PBF, c->n, c->K, 0, max_level, DQ_OUT_INFO | (u_hat ? DQ_OUT_UHAT : 0) | (dec_llr ? DQ_OUT_DECLLR : 0),
                    (crc ? DQ_CRC : 0) | (split ? DQ_SPLIT : 0) | ((flips_in || !rows) ? DQ_FLIP_LIST : 0) | dq_switch_flags("POLAR_PBF_KERNEL")};
    const int row = dispatch_select(q);
    return row >= 0 && dispatch_table(nullptr)[row].family == KF_PBF_REG;
}

template <typename T> int launch_pbf_reg(polar_code *c, PBFParams<T> &p, int64_t B, const PbfSplit *split, int max_level)
{
    return launch_pbf_reg_n<T>(c, p, B, split, max_level);
}
template int launch_pbf_reg<float>(polar_code *, PBFParams<float> &, int64_t, const PbfSplit *, int);
template int launch_pbf_reg<double>(polar_code *, PBFParams<double> &, int64_t, const PbfSplit *, int);
