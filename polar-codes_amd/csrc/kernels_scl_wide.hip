// Register-resident list decoder for long lists: up to L = 32 paths (L = 16 leaves half of the positions empty; at N = 2048
// and 4096 also L = 2, 4, 8), N = 64 .. 4096, fp32 or fp64, information bits of the best path (or of the first path in metric order that passes the
// CRC / equals the transmitted word) and, when asked for, the whole list in metric order with its metrics: one wavefront
// per frame, two lanes per path.
//
// Lane 2p + h belongs to list position p.  Element k of every tree level of a path lives in the lane with h = k & 1,
// at index k >> 1, so the partners (k, k + half) of every f and g down to level 2 sit in one lane and are plain
// register arithmetic; only the 2-leaf nodes exchange values between the two lanes of a path (one DPP swap).
// Levels 1 .. 6 (63 values per lane) stay in registers (level 6 in LDS from N = 256 on); the widest level below the
// channel is stored nowhere: it is one f or g away from the channel LLRs, which every path reads and which are staged
// once per frame in LDS, split by parity, so the two steps that consume it recompute their operands from there (with
// the path's own partial sums of the root's left half, for the right half); the levels between (N = 512: level 7,
// 16 KB per wavefront; N = 1024: levels 7 and 8, 48 KB; N = 2048: levels 7 to 9, 112 KB; N = 4096: 7 to 10, 240 KB) live lane-major in a slab of global
// memory per wavefront.
// Partial sums are bit slices in registers.
//
// Forking never copies a slice: as in the 8-lanes-per-path kernel (kernels_scl_reg.hip) every path keeps rows of
// 5-bit "owner" positions per level for its LLRs and its left-child partial sums, all paths rewrite a level in
// lockstep into their own registers, and g / combine fetch from the owner's lanes (ds_bpermute, or the owner's slots
// of the slab).
//
// At an information bit lane 2p + h IS candidate (p, u = h): with 32 paths the 64 candidates fill the wavefront.
// Every lane publishes an integer key with the order of its metric in LDS at its place in the candidate order
// copies(u=1) ++ originals(u=0) and counts the keys larger than its own (16 broadcast reads of four keys).  Exactly
// equal metrics are rare; they show as a missing or doubly claimed rank and send the wavefront through a second
// count that adds the equal keys standing earlier in the candidate order -- the stable descending sort of the
// reference's polarSCLFork (cn/polarCode.py:969-1005).  Rank r < L becomes list position r.
// Metric, arithmetic and order of operations are those of scl_decode_kernel<T, L>: the two give the same bits, and in
// fp64 (the reference's arithmetic: the metric as cn/polarCode.py:982-988 writes it, operation by operation) the same
// metrics.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "polar_internal.h"
#include "reg_common.h"

namespace {

// -DPOLAR_WIDE_STATS: event counts of the ranking, summed over wavefronts (development aid, read by polar_debug_wide_stats)
#ifdef POLAR_WIDE_STATS
__device__ unsigned long long g_wide_stats[8];
#define WIDE_STAT(i)                                                \
    do {                                                            \
        if (lane == 0) atomicAdd(&g_wide_stats[i], 1ull);           \
    } while (0)
#else
#define WIDE_STAT(i)
#endif

__device__ __forceinline__ float wb_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(lane << 2, __float_as_int(v))); }
__device__ __forceinline__ double wb_f(double v, int lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t wb_u(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)v); }
__device__ __forceinline__ uint64_t wb_u64(uint64_t v, int lane)
{
    return ((uint64_t)wb_u((uint32_t)(v >> 32), lane) << 32) | wb_u((uint32_t)v, lane);
}
template <typename T> __device__ __forceinline__ T swap2(T v) { return dpp_mov<0xB1>(v); }  // quad_perm:[1,0,3,2]: the other lane of my path
__device__ __forceinline__ float lane_of(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double lane_of(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// fp32 branch metric of deciding u at decision LLR v, as in the other list kernels (overflow-free; -inf only where the
// reference overflows)
__device__ __forceinline__ float wide_metric(float v, uint32_t u)
{
    const float Li = v * 1.44269504088896340736f;
    const float e = u ? Li : -Li;
    const float m = fmaxf(e, 0.0f);
    const float r = 1.0f - (m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(e))));
    return e >= 1024.0f ? -INFINITY : r;
}
// fp64: the reference's expression, operation by operation (cn/polarCode.py:982-988 and :2066-2067)
__device__ __forceinline__ double wide_metric(double v, uint32_t u)
{
    const double Li = v / 0.693147180559945309417232121458;
    const double e = u ? Li : -Li;
    return 1.0 - log2(1.0 + exp2(e));
}
// integer with the order of the metric (no NaN and no -0 occur among metrics)
__device__ __forceinline__ uint32_t order_key(float v)
{
    const uint32_t x = __float_as_uint(v);
    return x ^ ((uint32_t)((int32_t)x >> 31) | 0x80000000u);
}
__device__ __forceinline__ uint64_t order_key(double v)
{
    const uint64_t x = (uint64_t)__double_as_longlong(v);
    return x ^ ((uint64_t)((int64_t)x >> 63) | 0x8000000000000000ull);
}
template <typename T> struct KeyOf { using type = uint32_t; };
template <> struct KeyOf<double> { using type = uint64_t; };

template <int NLOG, typename T> struct WideStore {
    static constexpr int TOP = NLOG - 1;                         // widest stored level
    static constexpr bool SPLIT = NLOG >= (sizeof(T) == 4 ? 8 : 7);  // from N = 256 on (fp64: 128) level 6 (32 values per lane) lives in LDS, lane-major
    static constexpr int LDS_LEVEL = SPLIT ? 6 : 99;
    static constexpr int LDS_VALUES = SPLIT ? 32 * 64 : 1;       // values per wavefront
    static constexpr int REG_TOP = SPLIT ? 5 : (TOP < 6 ? TOP : 6);  // levels 1 .. REG_TOP in registers
    static constexpr int REG_VALUES = 1 << REG_TOP;              // level l: 2^(l-1) values at offset 2^(l-1)
    static constexpr bool RECOMP = TOP >= 7;                     // level TOP is recomputed from the channel by its two consumers
    static constexpr int MEM_TOP = RECOMP ? TOP - 1 : TOP;       // widest level kept in the slab
    static constexpr int mem_off(int l)                          // first slab row of level l (rows of 64 lanes)
    {
        int o = 0;
        for (int i = 7; i < l; i++) o += 1 << (i - 1);
        return o;
    }
    static constexpr int MEM_ROWS = MEM_TOP >= 7 ? mem_off(MEM_TOP + 1) : 0;
};

template <int NLOG, int LMAX, typename T> struct WideDec {
    using St = WideStore<NLOG, T>;
    using Key = typename KeyOf<T>::type;
    static constexpr int N = 1 << NLOG;
    T R[St::REG_VALUES];
    T *memu;               // the wavefront's slab (wave-uniform)
    T *lds6u;              // the wavefront's LDS level (wave-uniform)
    const T *chs;          // the frame's channel LLRs in LDS, [parity][N/2], + my parity's half
    int lane, p, h;
    // partial sums per plane: a level whose nodes have C bits per lane sits in W[0] at bit offset C for C <= 32, in W[1]
    // for C = 64, in W[C / 64 .. C / 32) from there on (W[2..3] for C = 128, W[4..7] for 256, W[8..15] for 512); the root
    // (N / 2 bits per lane) fills Rt from bit 0
    static constexpr int SW = NLOG >= 10 ? N / 128 : 4;
    static constexpr int BW = SW >= 8 ? SW / 2 : 4;  // words of the widest child
    uint64_t A[SW], Bp[SW], Rt[SW];
#ifndef POLAR_WIDE_EAGER_A
#define POLAR_WIDE_EAGER_A 1
#endif
    static constexpr bool EAGER_A = POLAR_WIDE_EAGER_A != 0;
#ifndef POLAR_WIDE_EAGER_L_TOP
#define POLAR_WIDE_EAGER_L_TOP 1
#endif
    static constexpr int EAGER_L_TOP = POLAR_WIDE_EAGER_L_TOP;  // LLR levels up to this one travel with the path (level 1 is x)
    T pm;
    uint32_t oL[2], oP[2];  // owner rows, 5 bits per level: levels 1 .. 6 in word 0, from 7 on in word 1; oL[1] bit 31: see node2
    T x;                    // my element of the enclosing 2-leaf node's LLRs while its first leaf is decided
    int Lcur;
    bool right;             // the walk is in the right half of the root (wave-uniform): level TOP is then g of the channel

    template <int LEV> static __device__ __forceinline__ uint32_t own_get(const uint32_t *row)
    {
        return (LEV <= 6 ? row[0] >> (5 * (LEV - 1)) : row[1] >> (5 * (LEV - 7))) & 31u;
    }
    template <int LEV> static __device__ __forceinline__ void own_set(uint32_t *row, uint32_t pos)
    {
        if constexpr (LEV <= 6) row[0] = (row[0] & ~(31u << (5 * (LEV - 1)))) | (pos << (5 * (LEV - 1)));
        else row[1] = (row[1] & ~(31u << (5 * (LEV - 7)))) | (pos << (5 * (LEV - 7)));
    }

    // Accesses to the levels in memory take the lane number through an opaque copy made inside the step (`lq`): their
    // addresses are then formed there, base plus instruction offset, instead of being hoisted out of the frame loop
    // as one 64-bit register pair per slab row.
    static __device__ __forceinline__ int opaque(int v)
    {
        asm volatile("" : "+v"(v));
        return v;
    }
    template <int L> __device__ __forceinline__ T ld_own(int j, int lq) const
    {
        if constexpr (L >= 7) return memu[(St::mem_off(L) + j) * 64 + lq];
        else if constexpr (L == St::LDS_LEVEL) return lds6u[j * 64 + lq];
        else return R[(1 << (L - 1)) + j];
    }
    template <int L> __device__ __forceinline__ T ld_from(int j, int src) const
    {
        if constexpr (L <= EAGER_L_TOP) return R[(1 << (L - 1)) + j];  // travels with the path
        else if constexpr (L >= 7) return memu[(St::mem_off(L) + j) * 64 + src];
        else if constexpr (L == St::LDS_LEVEL) return lds6u[j * 64 + src];
        else return wb_f(R[(1 << (L - 1)) + j], src);
    }
    template <int L> __device__ __forceinline__ void st_own(int j, T v, int lq)
    {
        if constexpr (L >= 7) memu[(St::mem_off(L) + j) * 64 + lq] = v;
        else if constexpr (L == St::LDS_LEVEL) lds6u[j * 64 + lq] = v;
        else R[(1 << (L - 1)) + j] = v;
    }
    template <int L> static __device__ __forceinline__ void level_written()
    {
        if constexpr (L >= 7) {  // other lanes of the wavefront read the slab rows next
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else if constexpr (L == St::LDS_LEVEL) {
            wave_sync();
        }
    }

    // ---- bit slices -----------------------------------------------------------------------------------------
    struct Bits {
        uint64_t w[BW];
    };
    template <int C> static __device__ __forceinline__ Bits get_bits(const uint64_t *W)
    {
        Bits b{};
        if constexpr (C <= 32) b.w[0] = (W[0] >> C) & ((((uint64_t)1) << C) - 1);
        else if constexpr (C == 64) b.w[0] = W[1];
        else {
#pragma unroll
            for (int i = 0; i < C / 64; i++) b.w[i] = W[C / 64 + i];
        }
        return b;
    }
    // the C bits of plane-0 level that lanes `src` hold
    // (the levels of up to 32 bits per lane, all in W[0], travel with the path at a fork or prune -- take_over -- so they are
    // the lane's own: the narrow levels are walked most often, and a fetch there is a dependent round trip per step)
    template <int C> __device__ __forceinline__ Bits fetch_bits(int src) const
    {
        Bits b{};
        if constexpr (C <= 16 && EAGER_A) b.w[0] = (uint64_t)(((uint32_t)A[0] >> C) & ((1u << C) - 1u));
        else if constexpr (C == 32 && EAGER_A) b.w[0] = A[0] >> 32;
        else if constexpr (C <= 16) b.w[0] = (uint64_t)((wb_u((uint32_t)A[0], src) >> C) & ((1u << C) - 1u));
        else if constexpr (C == 32) b.w[0] = (uint64_t)wb_u((uint32_t)(A[0] >> 32), src);
        else if constexpr (C == 64) b.w[0] = wb_u64(A[1], src);
        else {
#pragma unroll
            for (int i = 0; i < C / 64; i++) b.w[i] = wb_u64(A[C / 64 + i], src);
        }
        return b;
    }
    // store the 2C bits (xl ^ xr, xr) of a node whose children have C bits per lane
    template <int C> static __device__ __forceinline__ void put_pair(uint64_t *W, bool root, const Bits &xl, const Bits &xr)
    {
        if constexpr (C <= 16) {
            const uint64_t x = (xl.w[0] ^ xr.w[0]) | (xr.w[0] << C);
            if (root) W[0] = x;
            else {
                const uint64_t m = ((((uint64_t)1) << (2 * C)) - 1) << (2 * C);
                W[0] = (W[0] & ~m) | (x << (2 * C));
            }
        } else if constexpr (C == 32) {
            const uint64_t x = (xl.w[0] ^ xr.w[0]) | (xr.w[0] << 32);
            if (root) W[0] = x;
            else W[1] = x;
        } else {  // whole words: the node's 2 C bits at W[2 C / 64 ..), or from W[0] for the root
            constexpr int nw = C / 64;
            const int o = root ? 0 : 2 * nw;
#pragma unroll
            for (int i = 0; i < nw; i++) {
                W[o + i] = xl.w[i] ^ xr.w[i];
                W[o + nw + i] = xr.w[i];
            }
        }
    }

    // ---- f / g steps, level LEV -> LEV - 1 (2 <= LEV <= NLOG): c results per lane -------------------------------
    // element j of level TOP, recomputed from the channel: f for the root's left half, g with the path's partial sums of the
    // left half (tb) for the right half
    template <bool RIGHT> __device__ __forceinline__ T top_val(int j, const Bits &tb) const
    {
        constexpr int ct = 1 << (NLOG - 2);
        const T a = chs[j], b = chs[j + ct];
        if constexpr (RIGHT) return g_fast(a, b, (uint32_t)(tb.w[j >> 6] >> (j & 63)) & 1u);
        else return f_med(a, b);
    }
    template <int LEV> __device__ __forceinline__ void step_f()
    {
        constexpr int c = 1 << (LEV - 2);
        if constexpr (St::RECOMP && LEV == NLOG) {
            right = false;  // level TOP is made where it is used
            return;
        }
        constexpr bool FROM_TOP = St::RECOMP && LEV == St::TOP;
        constexpr bool MEM = LEV == NLOG || LEV >= 7 || LEV == St::LDS_LEVEL;
        constexpr int BMAX = sizeof(T) == 4 ? 16 : 8;  // pairs in flight: 32 registers of operands either way
        constexpr int BT = !MEM ? 1 : c < BMAX ? c : BMAX;
        const int lq = opaque(lane);
        Bits tb{};
        if constexpr (FROM_TOP)
            if (right) tb = fetch_bits<(1 << (NLOG - 2))>((int)own_get<St::TOP>(oP) * 2 + h);
#pragma unroll
        for (int j0 = 0; j0 < c; j0 += BT) {
            T a[BT], b[BT];
            if constexpr (FROM_TOP) {
                if (right) {
#pragma unroll
                    for (int t = 0; t < BT; t++) {
                        a[t] = top_val<true>(j0 + t, tb);
                        b[t] = top_val<true>(c + j0 + t, tb);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < BT; t++) {
                        a[t] = top_val<false>(j0 + t, tb);
                        b[t] = top_val<false>(c + j0 + t, tb);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < BT; t++) {
                    if constexpr (LEV == NLOG) {
                        a[t] = chs[j0 + t];
                        b[t] = chs[j0 + t + c];
                    } else {
                        a[t] = ld_own<LEV>(j0 + t, lq);
                        b[t] = ld_own<LEV>(c + j0 + t, lq);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < BT; t++) st_own<LEV - 1>(j0 + t, f_med(a[t], b[t]), lq);
        }
        level_written<LEV - 1>();
        own_set<LEV - 1>(oL, (uint32_t)p);
    }
    template <int LEV> __device__ __forceinline__ void step_g()
    {
        constexpr int c = 1 << (LEV - 2);
        if constexpr (St::RECOMP && LEV == NLOG) {
            right = true;
            return;
        }
        constexpr bool FROM_TOP = St::RECOMP && LEV == St::TOP;
        constexpr bool MEM = LEV == NLOG || LEV >= 7 || LEV == St::LDS_LEVEL;
        constexpr int BMAX = sizeof(T) == 4 ? 16 : 8;
        constexpr int BT = MEM ? (c < BMAX ? c : BMAX) : (c < 4 ? c : 4);
        const int lq = opaque(lane);
        const int srcP = (int)own_get<LEV - 1>(oP) * 2 + h;
        const int srcL = opaque((LEV == NLOG || FROM_TOP) ? 0 : (int)own_get<(LEV == NLOG ? 1 : LEV)>(oL) * 2 + h);
        const Bits bits = fetch_bits<c>(srcP);
        Bits tb{};
        if constexpr (FROM_TOP)
            if (right) tb = fetch_bits<(1 << (NLOG - 2))>((int)own_get<St::TOP>(oP) * 2 + h);
#pragma unroll
        for (int j0 = 0; j0 < c; j0 += BT) {
            T a[BT], b[BT];
            if constexpr (FROM_TOP) {
                if (right) {
#pragma unroll
                    for (int t = 0; t < BT; t++) {
                        a[t] = top_val<true>(j0 + t, tb);
                        b[t] = top_val<true>(c + j0 + t, tb);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < BT; t++) {
                        a[t] = top_val<false>(j0 + t, tb);
                        b[t] = top_val<false>(c + j0 + t, tb);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < BT; t++) {
                    if constexpr (LEV == NLOG) {
                        a[t] = chs[j0 + t];
                        b[t] = chs[j0 + t + c];
                    } else {
                        a[t] = ld_from<LEV>(j0 + t, srcL);
                        b[t] = ld_from<LEV>(c + j0 + t, srcL);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < BT; t++) {
                const int j = j0 + t;
                st_own<LEV - 1>(j, g_fast(a[t], b[t], (uint32_t)(bits.w[j >> 6] >> (j & 63)) & 1u), lq);
            }
        }
        level_written<LEV - 1>();
        own_set<LEV - 1>(oL, (uint32_t)p);
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane)
    {
        constexpr int C = 1 << (LEV - 2);  // bits per lane of each child
        const int srcP = (int)own_get<LEV - 1>(oP) * 2 + h;
        const Bits xl = fetch_bits<C>(srcP);
        const Bits xr = get_bits<C>(Bp);
        if constexpr (LEV == NLOG) {
            put_pair<C>(Rt, true, xl, xr);
        } else if (plane) {
            put_pair<C>(Bp, false, xl, xr);
        } else {
            put_pair<C>(A, false, xl, xr);
            own_set<LEV>(oP, (uint32_t)p);
        }
    }

    // ---- fork / prune -------------------------------------------------------------------------------------------
    // my position becomes the continuation u = b of the path in position par (cpm: every lane's candidate metric);
    // WITH_X: the first leaf of a 2-leaf node, whose LLRs travel with the path
    template <bool WITH_X> __device__ __forceinline__ void take_over(T cpm, int par, uint32_t b)
    {
        const int src = par * 2 + h;
        pm = wb_f(cpm, par * 2 + (int)b);
        oL[0] = wb_u(oL[0], src);
        oL[1] = wb_u(oL[1], src);
        oP[0] = wb_u(oP[0], src);
        oP[1] = wb_u(oP[1], src);
        if constexpr (EAGER_A) A[0] = wb_u64(A[0], src);
#pragma unroll
        for (int i = 2; i < (2 << (EAGER_L_TOP - 1)); i++) R[i] = wb_f(R[i], src);
        if (WITH_X) x = wb_f(x, src);
    }

    // One leaf; v = its decision LLR (the same in both lanes of a path).  Returns the bit of the path that holds
    // this position afterwards.  keys: 64 words of LDS, inv: 32 words.
    template <bool WITH_X> __device__ __forceinline__ uint32_t leaf(T v, uint32_t is_info, Key *keys, uint32_t *inv)
    {
        if (!is_info) {
            pm += wide_metric(v, 0u);
            return 0u;
        }
        if (2 * Lcur > LMAX) {
            // The list is full.  A leaf every path is sure of ends before any metric is evaluated or ranked (sure_exponent in
            // decode_common.h has the argument): the ranking below would give each position its own path's agreeing extension
            // with this very metric, and nothing would move.
            T ae;
            const bool big = sure_exponent(v, ae);
            const T cpm_a = pm + (T)1;
            const T next = next_position(cpm_a);  // the agreeing extension of the next list position
            const T other = pm + ((T)2 - ae);
            const T last = lane_of(cpm_a, 2 * (LMAX - 1));
            // one expression without short-circuit evaluation (nested branches on the lanes' own outcomes are slower)
            const bool sure = (bool)((int)big & (int)(other < last) & ((int)(p == LMAX - 1) | (int)(cpm_a > next)));
            if (__ballot(sure || (LMAX < 32 && p >= LMAX)) == ~0ull) {
                pm = cpm_a;
                return v < (T)0 ? 1u : 0u;
            }
        }
        const T cpm = pm + wide_metric(v, (uint32_t)h);
        if (2 * Lcur <= LMAX) {  // list = originals ++ copies
            const bool is_copy = p >= Lcur && p < 2 * Lcur;
            const int par = is_copy ? p - Lcur : p;
            const uint32_t b = is_copy ? 1u : 0u;
            Lcur *= 2;
            take_over<WITH_X>(cpm, par, b);
            return b;
        }
        const bool alive = LMAX == 32 || p < LMAX;
        const Key key = alive ? order_key(cpm) : (Key)0;  // below the key of -inf
        uint32_t win;
        WIDE_STAT(0);
        if (merge_rank(v, key, alive, inv, win)) return settle<WITH_X>(cpm, win, alive);
        const int mypos = h ? p : 32 + p;                  // copies (u = 1) come first in the candidate order
        keys[mypos] = key;
        if (lane < LMAX) inv[lane] = 0xFFFFFFFFu;
        wave_sync();
        int cnt = 0;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < 64; i += 4) {
                const uint4 k4 = *reinterpret_cast<const uint4 *>(keys + i);
                cnt += (k4.x > key) ? 1 : 0;
                cnt += (k4.y > key) ? 1 : 0;
                cnt += (k4.z > key) ? 1 : 0;
                cnt += (k4.w > key) ? 1 : 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 64; i += 2) {
                const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(keys + i);
                cnt += (k2.x > key) ? 1 : 0;
                cnt += (k2.y > key) ? 1 : 0;
            }
        }
        if (cnt < LMAX) inv[cnt] = (uint32_t)lane;
        wave_sync();
        win = inv[p < LMAX ? p : 0];
        const uint64_t claimed = __ballot(cnt < LMAX);
        const uint64_t empty = __ballot(alive && win == 0xFFFFFFFFu);
        if (__popcll(claimed) != LMAX || empty != 0) {
            // equal metrics among the candidates that matter: the exact order, with the position as second key
            wave_sync();
            if (lane < LMAX) inv[lane] = 0xFFFFFFFFu;
            wave_sync();
            cnt = 0;
#pragma unroll 4
            for (int i = 0; i < 64; i++) {
                const Key k = keys[i];
                cnt += (k > key || (k == key && i < mypos)) ? 1 : 0;
            }
            if (cnt < LMAX) inv[cnt] = (uint32_t)lane;
            wave_sync();
            win = inv[p < LMAX ? p : 0];
        }
        wave_sync();
        return settle<WITH_X>(cpm, win, alive);
    }

    // my position takes the candidate in lane `win` (= 2 * path + bit)
    template <bool WITH_X> __device__ __forceinline__ uint32_t settle(T cpm, uint32_t win, bool alive)
    {
        const uint32_t b = win & 1u;
        const int par = (int)(win >> 1);
        if (__ballot(alive && par != p) == 0) {  // nothing moves between positions
            const T other = swap2(cpm);
            pm = (b == (uint32_t)h) ? cpm : other;
            return b;
        }
        take_over<WITH_X>(cpm, par, b);
        return b;
    }

    static __device__ __forceinline__ Key key_of_lane(Key k, int src)
    {
        if constexpr (sizeof(Key) == 4) return (Key)__builtin_amdgcn_readlane((int)k, src);
        else return ((Key)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, src);
    }
    // the value the next list position (lane + 2) holds: two wavefront shifts by one lane, which stay on the vector ALU (a
    // ds_bpermute is a round trip through the LDS pipeline in the middle of the leaf's dependent chain); lanes 62 and 63 get
    // what the shift leaves there and belong to the last position, which has no next
    template <typename V> static __device__ __forceinline__ V next_position(V v)
    {
        return dpp_mov<0x130>(dpp_mov<0x130>(v));  // wave_shl:1 twice
    }
    static __device__ __forceinline__ Key next_position_key(Key k)
    {
        if constexpr (sizeof(Key) == 4) return dpp_mov_u32<0x130>(dpp_mov_u32<0x130>(k));
        else return ((Key)dpp_mov_u32<0x130>(dpp_mov_u32<0x130>((uint32_t)(k >> 32))) << 32) | dpp_mov_u32<0x130>(dpp_mov_u32<0x130>((uint32_t)k));
    }

    // Ranking by merging, for a common shape of a full list: the agreeing extensions (the candidate of each path that
    // follows its LLR's sign) stand in strictly descending order of their metrics by list position, and only a few
    // disagreeing extensions -- contenders -- reach the metric of the last of them.  The agreeing extension of position p
    // then has rank p + (contenders above it); a contender has rank (agreeing above it) + (contenders above it), one
    // wave-wide comparison each; every other disagreeing extension has all L agreeing ones above it and is out.  These are
    // the counts of larger keys that the ranking over all 64 keys makes when no two keys that matter are equal; an agreeing
    // pair out of order, any equality with a contender or more than MERGE_MAX contenders returns false, and the caller
    // ranks the long way.  win: the lane whose candidate takes my position.
    static constexpr int MERGE_MAX = 6;
    __device__ __forceinline__ bool merge_rank(T v, Key key, bool alive, uint32_t *inv, uint32_t &win)
    {
        const uint32_t a = v < (T)0 ? 1u : 0u;
        const bool keeper = (uint32_t)h == a;              // my candidate is my path's agreeing extension
        const Key okey = swap2_key(key);
        const Key kk = keeper ? key : okey;                // my path's agreeing extension, in both of its lanes
        const Key kk_next = next_position_key(kk);
        const bool in_order = !alive || p == LMAX - 1 || kk > kk_next;
        const Key bar = key_of_lane(kk, 2 * (LMAX - 1));
        const bool contender = alive && !keeper && key >= bar;
        const uint64_t cont = __ballot(contender);
        if (__ballot(in_order) != ~0ull || __popcll(cont) > MERGE_MAX) return false;
        WIDE_STAT(1);
        win = (uint32_t)(2 * p) + a;                       // no contender: every position keeps its path
        if (cont == 0) return true;
        int above = 0;                 // contenders above my path's agreeing extension
        int myrank = 64;               // contenders get theirs below, the other disagreeing extensions are out
        bool tie = false;
        for (uint64_t rest = cont; rest != 0; rest &= rest - 1) {
            WIDE_STAT(2);
            const int l = __builtin_ctzll(rest);
            const Key f = key_of_lane(key, l);
            const uint64_t agree_above = __ballot(alive && kk > f) & 0x5555555555555555ull;
            const uint64_t cont_above = __ballot(contender && key > f);
            tie |= __ballot((alive && kk == f) || (contender && key == f && lane != l)) != 0;
            above += (f > kk) ? 1 : 0;
            if (lane == l) myrank = __popcll(agree_above) + __popcll(cont_above);
        }
        if (tie) return false;
        if (keeper) myrank = p + above;
        if (alive && myrank < LMAX) inv[myrank] = (uint32_t)lane;  // ranks 0 .. L-1 are taken once each
        wave_sync();
        win = inv[p < LMAX ? p : 0];
        wave_sync();
        return true;
    }
    static __device__ __forceinline__ Key swap2_key(Key k)
    {
        if constexpr (sizeof(Key) == 4) return dpp_mov_u32<0xB1>(k);
        else return ((Key)dpp_mov_u32<0xB1>((uint32_t)(k >> 32)) << 32) | dpp_mov_u32<0xB1>((uint32_t)k);
    }

    // A 2-leaf node; xin = my element of its LLRs, m2 = its rate profile.  Returns my element of its partial sums.
    __device__ __forceinline__ uint32_t node2(T xin, uint32_t m2, Key *keys, uint32_t *inv)
    {
        x = xin;
        const T v0 = f_fast(x, swap2(x));
        const uint32_t u0 = leaf<true>(v0, m2 & 1u, keys, inv);
        oL[1] = (oL[1] & 0x7FFFFFFFu) | (u0 << 31);  // travels with the path through the forks of the second leaf
        const T t = h ? x : flip_sign(x, u0);
        const T v1 = t + swap2(t);
        const uint32_t u1 = leaf<false>(v1, (m2 >> 1) & 1u, keys, inv);
        const uint32_t u0_now = oL[1] >> 31;
        return h ? u1 : (u0_now ^ u1);
    }
};

template <int LEV, int NLOG, int LMAX, typename T>
__device__ __forceinline__ void wide_node(WideDec<NLOG, LMAX, T> &d, uint64_t maskv, uint32_t idx, int plane, typename KeyOf<T>::type *keys,
                                          uint32_t *inv)
{
#pragma nounroll
    for (uint32_t hh = 0; hh < 2; hh++) {
        if (hh == 0) d.template step_f<LEV>();
        else d.template step_g<LEV>();
        const uint32_t cidx = 2 * idx + hh;
        if constexpr (LEV == 2) {
            // rate-profile word cidx >> 4 (32 positions): lane w holds word w in the low half of maskv and, for N = 4096,
            // word 64 + w in the high half
            uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)maskv, (int)((cidx >> 4) & 63u));
            if constexpr (NLOG >= 12) {
                const uint32_t mh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(maskv >> 32), (int)((cidx >> 4) & 63u));
                mw = (cidx >> 4) >= 64u ? mh : mw;
            }
            const uint32_t m2 = (mw >> (2 * (cidx & 15))) & 3u;
            const uint32_t xb = d.node2(d.R[1], m2, keys, inv);
            if (hh) {
                d.Bp[0] = (d.Bp[0] & ~(uint64_t)2) | ((uint64_t)xb << 1);
            } else {
                d.A[0] = (d.A[0] & ~(uint64_t)2) | ((uint64_t)xb << 1);
                WideDec<NLOG, LMAX, T>::template own_set<1>(d.oP, (uint32_t)d.p);
            }
        } else {
            wide_node<LEV - 1, NLOG, LMAX, T>(d, maskv, cidx, (int)hh, keys, inv);
        }
    }
    d.template combine<LEV>(plane);
}

// 16 bits spread to the even bit positions of a word
__device__ __forceinline__ uint32_t spread16(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// List outputs (pm_out, u_hat_all; either may be NULL) are those of scl_decode_kernel: row f * LMAX + r holds the path of
// rank r in the stable descending order of the metrics, rows of empty positions keep their index, -inf and zeros.
template <int NLOG, int LMAX, typename T>
__global__ void __launch_bounds__(64, sizeof(T) == 4 ? (NLOG >= 12 ? 1 : NLOG == 11 ? 2 : 3) : (NLOG >= 11 ? 1 : 2))
scl_wide_kernel(const T *__restrict__ llr, const uint32_t *__restrict__ maskw, const uint32_t *__restrict__ Apos,
                uint32_t *__restrict__ info_hat, int64_t B, int K, int Kw, T *__restrict__ slab, const uint32_t *__restrict__ crcS,
                const uint32_t *__restrict__ truth, double *__restrict__ pm_out, uint32_t *__restrict__ u_hat_all)
{
    constexpr int N = 1 << NLOG;
    constexpr int NW = N / 32;
    using Dec = WideDec<NLOG, LMAX, T>;
    using Key = typename Dec::Key;
    __shared__ __attribute__((aligned(16))) T chan_s[N];  // [parity][N/2]; the output stage reuses it for bit words
    __shared__ __attribute__((aligned(16))) Key keys[64];
    __shared__ uint32_t inv[32];
    __shared__ T lvl6_s[WideStore<NLOG, T>::LDS_VALUES];
    Dec d;
    d.lane = threadIdx.x;
    d.p = d.lane >> 1;
    d.h = d.lane & 1;
    d.memu = slab + (size_t)blockIdx.x * (WideStore<NLOG, T>::MEM_ROWS > 0 ? WideStore<NLOG, T>::MEM_ROWS : 1) * 64;
    d.chs = chan_s + d.h * (N / 2);
    d.lds6u = lvl6_s;
    uint64_t maskv = NW > 32 ? maskw[d.lane < NW ? d.lane : 0] : maskw[(d.lane & 31) < NW ? (d.lane & 31) : 0];
    if constexpr (NW > 64) maskv |= (uint64_t)maskw[64 + d.lane] << 32;

    for (int64_t f = blockIdx.x; f < B; f += gridDim.x) {
        // channel LLRs of the frame: coalesced 16-byte loads, stored split by parity (element 2j + h at [h][j])
        if constexpr (sizeof(T) == 4) {
            const float4 *src = reinterpret_cast<const float4 *>(llr + f * N);
#pragma unroll
            for (int t = d.lane; t < N / 4; t += 64) {
                const float4 v = src[t];
                *reinterpret_cast<float2 *>(chan_s + 2 * t) = make_float2(v.x, v.z);
                *reinterpret_cast<float2 *>(chan_s + N / 2 + 2 * t) = make_float2(v.y, v.w);
            }
        } else {
            const double2 *src = reinterpret_cast<const double2 *>(llr + f * N);
#pragma unroll
            for (int t = d.lane; t < N / 2; t += 64) {
                const double2 v = src[t];
                chan_s[t] = v.x;
                chan_s[N / 2 + t] = v.y;
            }
        }
#pragma unroll
        for (int i = 0; i < Dec::SW; i++) d.A[i] = d.Bp[i] = d.Rt[i] = 0;
        d.pm = (T)0;
        d.oL[0] = d.oL[1] = d.oP[0] = d.oP[1] = 0u;
        d.x = (T)0;
        d.Lcur = 1;
        d.right = false;
        wave_sync();
        wide_node<NLOG, NLOG, LMAX, T>(d, maskv, 0u, 0, keys, inv);

        // rank of every path in the stable descending order of the list
        const bool act = d.p < d.Lcur;
        const Key key = (act && d.h == 0) ? order_key(d.pm) : (Key)0;
        keys[d.h ? 32 + d.p : d.p] = key;
        wave_sync();
        int rank = 0;
#pragma unroll 4
        for (int q = 0; q < 32; q++) {
            const Key k = keys[q];
            rank += (k > key || (k == key && q < d.p)) ? 1 : 0;
        }
        wave_sync();
        if (act && d.h == 0) inv[rank] = (uint32_t)d.p;  // active paths take the ranks 0 .. Lcur-1
        wave_sync();
        constexpr int STREAM = 2 * Dec::SW;                   // words per lane stream
        uint32_t *bw = reinterpret_cast<uint32_t *>(chan_s);  // [2][STREAM] lane streams, then NW words of x_hat / u_hat
        uint32_t *uw = bw + (2 * STREAM > 32 ? 2 * STREAM : 32);
        // u_hat of the path in list position pos, as natural-order words in LDS: its root partial sums (x_hat, lane h holds
        // elements 2j + h), then x_hat F^{(x)n}
        auto path_words = [&](int pos) {
            wave_sync();
            if (d.p == pos) {
#pragma unroll
                for (int i = 0; i < Dec::SW; i++) {
                    bw[d.h * STREAM + 2 * i] = (uint32_t)d.Rt[i];
                    bw[d.h * STREAM + 2 * i + 1] = (uint32_t)(d.Rt[i] >> 32);
                }
            }
            wave_sync();
            for (int w = d.lane; w < NW; w += 64) {
                const uint32_t s0 = (bw[w >> 1] >> (16 * (w & 1))) & 0xFFFFu, s1 = (bw[STREAM + (w >> 1)] >> (16 * (w & 1))) & 0xFFFFu;
                uw[w] = butterfly_word(spread16(s0) | (spread16(s1) << 1), NLOG);
            }
            for (int s = 5; s < NLOG; s++) {
                wave_sync();
                const int dd = 1 << (s - 5);
                for (int w = d.lane; w < NW; w += 64)
                    if (!(w & dd)) uw[w] ^= uw[w + dd];
            }
            wave_sync();
        };
        auto info_bit = [&](int k) {
            const uint32_t pos = Apos[k];
            return ((uw[pos >> 5] >> (pos & 31)) & 1u) != 0u;
        };
        // the list itself, in metric order
        if (pm_out && d.h == 0 && d.p < LMAX) pm_out[f * LMAX + (act ? rank : d.p)] = act ? (double)d.pm : -INFINITY;
        if (u_hat_all) {
            for (int r = 0; r < LMAX; r++) {
                uint32_t *row = u_hat_all + ((size_t)f * LMAX + r) * NW;
                if (r < d.Lcur) {
                    path_words((int)inv[r]);
                    for (int w = d.lane; w < NW; w += 64) row[w] = uw[w];
                } else {
                    for (int w = d.lane; w < NW; w += 64) row[w] = 0u;
                }
            }
        }
        if (info_hat) {
            if (crcS || truth) {
                // aided choice: the first path in metric order that passes, the best one if none does
                int sel = -1;
                for (int r = 0; r < d.Lcur && sel < 0; r++) {
                    path_words((int)inv[r]);
                    bool ok;
                    if (truth) {
                        const uint32_t *tw = truth + f * Kw;
                        bool differs = false;
                        pack_bits_group<64>(K, d.lane, 0, info_bit, [&](int wi, uint32_t v) { differs |= (v != tw[wi]); });
                        ok = !differs;
                    } else {
                        uint32_t syn = 0u;
                        for (int k = d.lane; k < K; k += 64)
                            if (info_bit(k)) syn ^= crcS[k];
                        for (int off = 1; off < 64; off <<= 1) syn ^= (uint32_t)__shfl_xor((int)syn, off, 64);
                        ok = syn == 0u;
                    }
                    if (ok) sel = r;
                }
                if (sel < 0) path_words((int)inv[0]);
            } else {
                path_words((int)inv[0]);
            }
            uint32_t *dst = info_hat + f * Kw;
            pack_bits_group<64>(K, d.lane, 0, info_bit, [&](int wi, uint32_t v) {
                if (d.lane == 0) dst[wi] = v;
            });
            if (K == 0 && d.lane == 0) dst[0] = 0u;
        }
        wave_sync();
    }
}

template <int NLOG, int LMAX, typename T>
int launch_wide_t(polar_code *c, const void *llr, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                  const uint32_t *truth)
{
    polar_ctx *ctx = c->ctx;
    auto kern = scl_wide_kernel<NLOG, LMAX, T>;
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, 0));
    if (occ < 1) occ = 1;
    const int64_t grid = std::min<int64_t>(B, (int64_t)ctx->num_cus * occ);
    int rc;
    if ((rc = ctx->scl_slab.reserve(std::max<size_t>(4, (size_t)grid * WideStore<NLOG, T>::MEM_ROWS * 64 * sizeof(T))))) return rc;
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, ctx->stream, (const T *)llr, (const uint32_t *)c->d_maskw,
                       (const uint32_t *)c->d_A, info_hat, B, c->K, c->Kw, ctx->scl_slab.as<T>(),
                       crc ? (const uint32_t *)c->d_crcS : (const uint32_t *)nullptr, truth, pm, u_hat_all);
    HIP_TRY(hipGetLastError());
    const float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), sizeof(T) == 4 ? "scl_wide_kernel<%d>/L%d" : "scl_wide_kernel<%d,double>/L%d", NLOG, LMAX);
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = (int)sizeof(T) * (c->N + WideStore<NLOG, T>::LDS_VALUES) + 128 + 64 * (int)sizeof(typename KeyOf<T>::type);
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <int NLOG>
int launch_wide_l(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, int L, bool crc,
                  const uint32_t *truth)
{
    if constexpr (NLOG >= 11) {  // beyond the eight-lanes-per-path kernel: the short lists run here with part of the lanes
        if (L == 8) {
            if (dtype == POLAR_F32) return launch_wide_t<NLOG, 8, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
            return launch_wide_t<NLOG, 8, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        }
        if (L == 4) {
            if (dtype == POLAR_F32) return launch_wide_t<NLOG, 4, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
            return launch_wide_t<NLOG, 4, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        }
        if (L == 2) {
            if (dtype == POLAR_F32) return launch_wide_t<NLOG, 2, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
            return launch_wide_t<NLOG, 2, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        }
    }
    if (dtype == POLAR_F32) {
        if (L == 32) return launch_wide_t<NLOG, 32, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        return launch_wide_t<NLOG, 16, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    }
    if (L == 32) return launch_wide_t<NLOG, 32, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    return launch_wide_t<NLOG, 16, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
}

}  // namespace

#ifdef POLAR_WIDE_STATS
extern "C" __attribute__((visibility("default"))) int polar_debug_wide_stats(unsigned long long *out, int clear)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

bool scl_wide_supported(const polar_code *c, int dtype, int L, const void *pm, const void *u_hat_all)
{
    return dispatch_scl(c, dtype, L, pm, u_hat_all) == KF_SCL_WIDE;  // the table of host_tables.cpp says what this kernel serves
}

int launch_scl_wide(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, int L,
                    bool crc, const uint32_t *truth)
{
    if (B == 0) return POLAR_OK;
    switch (c->n) {
#ifdef POLAR_WIDE_DEV_LONG  // development builds: a few of the N = 4096 kernels only
    case 12:
        if (L == 32 && dtype == POLAR_F32) return launch_wide_t<12, 32, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        if (L == 32) return launch_wide_t<12, 32, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        if (L == 8 && dtype == POLAR_F32) return launch_wide_t<12, 8, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        polar_set_error("development build: not compiled");
        return POLAR_EUNSUPPORTED;
#elif defined(POLAR_WIDE_DEV_ONE)  // development builds: the (512, L = 32) kernels only, seconds to compile
    case 9:
        if (L == 32 && dtype == POLAR_F32) return launch_wide_t<9, 32, float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        if (L == 32) return launch_wide_t<9, 32, double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
        polar_set_error("development build: L = 32 only");
        return POLAR_EUNSUPPORTED;
#else
    case 6: return launch_wide_l<6>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 7: return launch_wide_l<7>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 8: return launch_wide_l<8>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 9: return launch_wide_l<9>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 10: return launch_wide_l<10>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 11: return launch_wide_l<11>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case 12: return launch_wide_l<12>(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
#endif
    default: polar_set_error("wide list kernel: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}
