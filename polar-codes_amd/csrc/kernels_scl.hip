// Batched successive-cancellation LIST decoder: one wavefront per frame (per list), L <= 32 paths,
// G = 64/L lanes per path.
//
// List position j lives in lane group j.  Paths never copy LLRs or partial sums when they fork: each
// path carries, in registers, a row of slot pointers (5 bits per tree level, packed in 64 bits) for
// its LLR levels and for its left-child ("plane 0") partial-sum levels.  All paths walk the schedule in
// lockstep, so whenever one path rewrites level l every path does; each writes into the slot of its
// own list position and points its row at it, which is why a slot can never be overwritten while
// another path still needs it.  A fork or a prune therefore only moves registers between lane groups
// (pointer rows, the word of narrow partial sums, the metric).  Right-child ("plane 1") partial sums
// are produced and consumed between two leaf decisions and stay with the list position.
//
// The widest stored level is n-2: the N/2 LLRs of level n-1 are not kept per path but recomputed from the
// channel LLRs (and, in the right half of the tree, the path's left-half partial sums) by the two
// operations that read them, which halves the LDS per wavefront and doubles the lists in flight per CU.
//
// Path metric (maximised): PM += 1 - log2(1 + 2^(-L/ln2 * (-1)^u)), evaluated in that order.  Below
// saturation the list grows as originals ++ copies; at saturation the 2L candidates are ordered
// copies(u=1) ++ originals(u=0), ranked by (metric descending, candidate position ascending) -- the
// stable descending sort of the reference -- and rank r < L becomes list position r.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "decode_common.h"
#include "polar_internal.h"

template <typename T> struct SCLParams {
    const T *llr;
    const uint32_t *ops;
    const uint32_t *A;
    const uint32_t *crcS;  // per information bit: its contribution to the CRC remainder; NULL = take the best path
    const uint32_t *truth; // [B x Kw] transmitted information words; non-NULL = take the list entry equal to them
    uint32_t *info_hat;
    double *pm;
    uint32_t *u_hat_all;
    int64_t B;
    int N, n, Nw, K, Kw;
    int slot_stride;  // bytes between the LLR trees of two slots
};

__device__ __forceinline__ uint32_t ptr_get(uint64_t row, int lev) { return (uint32_t)(row >> (5 * lev)) & 31u; }
__device__ __forceinline__ uint64_t ptr_set(uint64_t row, int lev, uint32_t slot)
{
    return (row & ~(31ull << (5 * lev))) | ((uint64_t)slot << (5 * lev));
}

__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = shfl_u32((uint32_t)v, src), hi = shfl_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float shfl_real(float v, int src) { return __uint_as_float(shfl_u32(__float_as_uint(v), src)); }
__device__ __forceinline__ double shfl_real(double v, int src)
{
    return __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(v), src));
}
__device__ __forceinline__ float bcast_real(float v, int src_uniform)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src_uniform));
}
__device__ __forceinline__ double bcast_real(double v, int src_uniform)
{
    uint64_t b = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_uniform);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_uniform);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// Branch metric of deciding u for decision LLR v.
__device__ __forceinline__ double branch_metric(double v, int u)
{
    double Li = v / 0.693147180559945309417232121458;
    double e = u ? Li : -Li;
    return 1.0 - log2(1.0 + exp2(e));
}
// fp32 mode: overflow-free form; only the reference's -inf for e >= 1024 is reproduced.
__device__ __forceinline__ float branch_metric(float v, int u)
{
    float Li = v * 1.44269504088896340736f;  // 1/ln2 (the fp64 parity path divides by ln2 like the reference)
    float e = u ? Li : -Li;
    float m = fmaxf(e, 0.0f);
    // hardware exp2 / log2 (about 1 ulp): this mode only has to match the reference statistically
    float r = 1.0f - (m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(e))));
    return e >= 1024.0f ? -INFINITY : r;
}

template <typename T, int L>
__global__ void __launch_bounds__(64) scl_decode_kernel(SCLParams<T> p, const uint32_t *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int G = 64 / L;
    const int lane = threadIdx.x;
    const int grp = lane / G;  // list position
    const int sub = lane % G;
    const int N = p.N, n = p.n, Nw = p.Nw;
    uint32_t *p0_all = reinterpret_cast<uint32_t *>(smem + (size_t)L * p.slot_stride);
    uint32_t *p1_all = p0_all + (size_t)L * 2 * Nw;
    uint32_t *surv = p1_all + (size_t)L * 2 * Nw;
    // slot s holds levels 0 .. n-2 of one path: level l (2^l values) at element offset 2^l
    auto tree = [&](uint32_t slot) { return reinterpret_cast<T *>(smem + (size_t)slot * p.slot_stride); };
    auto pl0 = [&](uint32_t slot) { return p0_all + (size_t)slot * 2 * Nw; };
    uint32_t *my_p1 = p1_all + (size_t)grp * 2 * Nw;

    for (int64_t f = blockIdx.x; f < p.B; f += gridDim.x) {
        const T *chan = p.llr + f * N;
        uint64_t ptrL = 0, ptrP = 0;  // slot rows; slot 0 initially
        uint32_t nar0 = 0, nar1 = 0;
        T pm = (T)0;
        int Lcur = 1;
        bool right_half = false;  // which child of the root is being decoded
        // LLR k of level n-1 for this path: f of the channel pair in the left half, g with the path's left-half
        // partial sum in the right half
        auto top = [&](int k) -> T {
            const T a = chan[k], b = chan[k + (N >> 1)];
            if (!right_half) return f_minsum(a, b);
            uint32_t bit;
            if (n - 1 >= 5) bit = (pl0(ptr_get(ptrP, n - 1))[(1 << (n - 6)) + (k >> 5)] >> (k & 31)) & 1u;
            else bit = (nar0 >> ((N >> 1) + k)) & 1u;
            return bit ? b - a : b + a;
        };

        uint32_t next_op = ops[0];
        for (int pc = 0;; pc++) {
            const uint32_t op = next_op;
            const uint32_t type = OP_TYPE(op);
            if (type == OP_END) break;
            next_op = ops[pc + 1];  // fetched while this operation runs
            const int lev = (int)OP_LEVEL(op);
            const int plane = (int)OP_PLANE(op);
            const bool act = grp < Lcur;
            switch (type) {
            case OP_F: {
                const int w = 1 << (lev - 1);
                if (lev == n && n >= 2) {
                    right_half = false;  // level n-1 is virtual
                    break;
                }
                T *dst = tree(grp) + w;
                if (lev == n - 1) {
                    if (act)
                        for (int k = sub; k < w; k += G) dst[k] = f_minsum(top(k), top(k + w));
                } else {
                    const T *src = (lev == n) ? chan : tree(ptr_get(ptrL, lev)) + (1 << lev);
                    if (act)
                        for (int k = sub; k < w; k += G) dst[k] = f_minsum(src[k], src[k + w]);
                }
                ptrL = ptr_set(ptrL, lev - 1, grp);
                break;
            }
            case OP_G: {
                const int w = 1 << (lev - 1);
                if (lev == n && n >= 2) {
                    right_half = true;  // level n-1 is virtual
                    break;
                }
                T *dst = tree(grp) + w;
                if (lev == n - 1) {
                    if (act) {
                        if (lev - 1 >= 5) {
                            const uint32_t *bits = pl0(ptr_get(ptrP, lev - 1)) + (1 << (lev - 6));
                            for (int k = sub; k < w; k += G) {
                                T a = top(k), b = top(k + w);
                                uint32_t bit = (bits[k >> 5] >> (k & 31)) & 1u;
                                dst[k] = bit ? b - a : b + a;
                            }
                        } else {
                            for (int k = sub; k < w; k += G) {
                                T a = top(k), b = top(k + w);
                                uint32_t bit = (nar0 >> (w + k)) & 1u;
                                dst[k] = bit ? b - a : b + a;
                            }
                        }
                    }
                    ptrL = ptr_set(ptrL, lev - 1, grp);
                    break;
                }
                const T *src = (lev == n) ? chan : tree(ptr_get(ptrL, lev)) + (1 << lev);
                if (act) {
                    if (lev - 1 >= 5) {
                        const uint32_t *bits = pl0(ptr_get(ptrP, lev - 1)) + (1 << (lev - 6));
                        for (int k = sub; k < w; k += G) {
                            T a = src[k], b = src[k + w];
                            uint32_t bit = (bits[k >> 5] >> (k & 31)) & 1u;
                            dst[k] = bit ? b - a : b + a;
                        }
                    } else {
                        for (int k = sub; k < w; k += G) {
                            T a = src[k], b = src[k + w];
                            uint32_t bit = (nar0 >> (w + k)) & 1u;
                            dst[k] = bit ? b - a : b + a;
                        }
                    }
                }
                ptrL = ptr_set(ptrL, lev - 1, grp);
                break;
            }
            case OP_LEAF_FROZEN: {
                T v = tree(ptr_get(ptrL, 0))[1];
                pm += branch_metric(v, 0);
                if (plane) nar1 &= ~2u;
                else nar0 &= ~2u;
                break;
            }
            case OP_LEAF_INFO: {
                T v = tree(ptr_get(ptrL, 0))[1];
                uint32_t bit;
                bool decided = false;
                if (2 * Lcur > L) {
                    // full list: a leaf every path is sure of is decided before any metric is evaluated or ranked
                    // (sure_exponent in decode_common.h has the argument)
                    T ae_lo;
                    const bool big = sure_exponent(v, ae_lo);
                    const T cpm_a = pm + (T)1;
                    const T next = shfl_real(cpm_a, lane + G < 64 ? lane + G : lane);  // the agreeing extension of the next list position
                    const T last = bcast_real(cpm_a, (L - 1) * G);
                    const bool sure = (bool)((int)big & (int)(pm + ((T)2 - ae_lo) < last) & ((int)(grp == L - 1) | (int)(cpm_a > next)));
                    if (__ballot(sure) == ~0ull) {
                        pm = cpm_a;
                        bit = v < (T)0 ? 1u : 0u;
                        decided = true;
                    }
                }
                // lane `sub` evaluates the metric of deciding (sub & 1); even/odd lanes of a group swap
                T bm_mine = (T)0, bm0 = (T)0, bm1 = (T)0;
                if (!decided) {
                    bm_mine = branch_metric(v, sub & 1);
                    bm0 = shfl_real(bm_mine, grp * G), bm1 = shfl_real(bm_mine, grp * G + 1);
                }
                if (decided) {
                } else if (2 * Lcur <= L) {
                    // list = originals ++ copies
                    const bool is_copy = grp >= Lcur && grp < 2 * Lcur;
                    const int src = is_copy ? (grp - Lcur) * G + sub : lane;
                    uint64_t sL = shfl_u64(ptrL, src), sP = shfl_u64(ptrP, src);
                    uint32_t sn = shfl_u32(nar0, src);
                    T spm = shfl_real(pm, src), sb1 = shfl_real(bm1, src);
                    if (is_copy) {
                        ptrL = sL;
                        ptrP = sP;
                        nar0 = sn;
                        pm = spm + sb1;
                        bit = 1u;
                    } else {
                        pm = pm + bm0;
                        bit = 0u;
                    }
                    Lcur *= 2;
                } else {
                    // candidates: lane sub==1 of group j is the copy (u=1), lane sub==0 the original (u=0)
                    const bool is_cand = act && sub < 2;
                    T cpm = pm + ((sub & 1) ? bm1 : bm0);
                    int cpos = (sub & 1) ? grp : Lcur + grp;  // copies first, then originals
                    if (!is_cand) cpos = 1 << 20;
                    int rank = 0;
                    for (int c = 0; c < 2 * Lcur; c++) {
                        const int src = (c >> 1) * G + (c & 1);
                        T opm = bcast_real(cpm, src);
                        int opos = __builtin_amdgcn_readlane(cpos, src);
                        rank += (opm > cpm || (opm == cpm && opos < cpos)) ? 1 : 0;
                    }
                    if (is_cand && rank < L) surv[rank] = (uint32_t)lane;
                    wave_sync();
                    const int c = (int)surv[grp < L ? grp : 0];
                    const int parent = (c / G) * G + sub;
                    wave_sync();
                    ptrL = shfl_u64(ptrL, parent);
                    ptrP = shfl_u64(ptrP, parent);
                    nar0 = shfl_u32(nar0, parent);
                    pm = shfl_real(cpm, c);
                    bit = (uint32_t)(c % G) & 1u;
                    Lcur = L;
                }
                if (plane) nar1 = (nar1 & ~2u) | (bit << 1);
                else nar0 = (nar0 & ~2u) | (bit << 1);
                break;
            }
            case OP_COMBINE: {
                const int cl = lev - 1;
                if (cl >= 5) {
                    const int nwc = 1 << (cl - 5);
                    const uint32_t *s0 = pl0(ptr_get(ptrP, cl)) + nwc, *s1 = my_p1 + nwc;
                    uint32_t *dst = (plane ? my_p1 : pl0(grp)) + 2 * nwc;
                    if (act)
                        for (int j = sub; j < nwc; j += G) {
                            uint32_t a = s0[j], b = s1[j];
                            dst[j] = a ^ b;
                            dst[nwc + j] = b;
                        }
                    if (!plane) ptrP = ptr_set(ptrP, lev, grp);
                } else if (cl == 4) {
                    uint32_t a = nar0 >> 16, b = nar1 >> 16;
                    uint32_t v = (a ^ b) | (b << 16);
                    if (sub == 0 && act) (plane ? my_p1 : pl0(grp))[1] = v;
                    if (!plane) ptrP = ptr_set(ptrP, lev, grp);
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

        // final stable descending sort of the list: rank of every path = its output index
        const bool act = grp < Lcur;
        T spm = act ? pm : (T)-INFINITY;
        int spos = act ? grp : (1 << 20);
        int rank = 0;
        for (int c = 0; c < Lcur; c++) {
            const int src = c * G;
            T opm = bcast_real(spm, src);
            int opos = __builtin_amdgcn_readlane(spos, src);
            rank += (opm > spm || (opm == spm && opos < spos)) ? 1 : 0;
        }
        if (!act) rank = grp;  // padding rows keep their index (all >= Lcur)

        // u_hat = x_hat F^{(x)n}; x_hat = root partial sums (always written into the own slot)
        uint32_t *uw = my_p1;
        if (n >= 5) {
            const uint32_t *xr = pl0(grp) + Nw;
            for (int j = sub; j < Nw; j += G) uw[j] = act ? butterfly_word(xr[j], n) : 0u;
            for (int s = 5; s < n; s++) {
                wave_sync();
                const int d = 1 << (s - 5);
                for (int j = sub; j < Nw; j += G)
                    if (!(j & d)) uw[j] ^= uw[j + d];
            }
        } else {
            uint32_t x = (nar0 >> N) & ((1u << N) - 1u);
            if (sub == 0) uw[0] = act ? butterfly_word(x, n) : 0u;
        }
        wave_sync();
        if (p.u_hat_all)
            for (int j = sub; j < Nw; j += G) p.u_hat_all[(f * L + rank) * Nw + j] = uw[j];
        if (p.pm && sub == 0) p.pm[f * L + rank] = (double)spm;
        if (p.info_hat) {
            // every group packs its own bits (the ballots are wave-wide); only the best path stores
            uint32_t *dst = p.info_hat + f * p.Kw;
            int sel = 0;
            if (p.crcS || p.truth) {
                // aided choice: the first path in sorted order that passes, the best one if none does
                bool ok;
                if (p.truth) {
                    // genie: the path's information word equals the transmitted one
                    const uint32_t *tw = p.truth + f * p.Kw;
                    bool differs = false;
                    pack_bits_group<G>(
                        p.K, sub, grp,
                        [&](int k) {
                            uint32_t pos = p.A[k];
                            return ((uw[pos >> 5] >> (pos & 31)) & 1u) != 0u;
                        },
                        [&](int wi, uint32_t v) { differs |= (v != tw[wi]); });
                    ok = !differs;
                } else {
                    // CRC: the information word leaves a zero remainder
                    uint32_t syn = 0;
                    for (int k = sub; k < p.K; k += G) {
                        uint32_t pos = p.A[k];
                        if ((uw[pos >> 5] >> (pos & 31)) & 1u) syn ^= p.crcS[k];
                    }
                    for (int off = 1; off < G; off <<= 1) syn ^= shfl_u32(syn, lane ^ off);
                    ok = syn == 0u;
                }
                if (lane == 0) surv[0] = 0u;
                wave_sync();
                if (sub == 0 && act && ok) atomicOr(&surv[0], 1u << rank);
                wave_sync();
                const uint32_t pass = surv[0];
                sel = pass ? __ffs(pass) - 1 : 0;
            }
            const bool best = act && rank == sel;
            pack_bits_group<G>(
                p.K, sub, grp,
                [&](int k) {
                    uint32_t pos = p.A[k];
                    return ((uw[pos >> 5] >> (pos & 31)) & 1u) != 0u;
                },
                [&](int wi, uint32_t v) {
                    if (sub == 0 && best) dst[wi] = v;
                });
            if (p.K == 0 && lane == 0) dst[0] = 0u;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------ launch
template <typename T, int L>
static int launch_scl_t(polar_code *c, const void *llr, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                       const uint32_t *truth)
{
    polar_ctx *ctx = c->ctx;
    constexpr int G = 64 / L;
    SCLParams<T> p;
    p.llr = (const T *)llr;
    p.ops = c->d_sched[0];
    p.A = c->d_A;
    p.crcS = crc ? c->d_crcS : nullptr;
    p.truth = truth;
    p.info_hat = info_hat;
    p.pm = pm;
    p.u_hat_all = u_hat_all;
    p.B = B;
    p.N = c->N;
    p.n = c->n;
    p.Nw = c->Nw;
    p.K = c->K;
    p.Kw = c->Kw;
    // slot stride: the tree up to level n-2 (N/2 values), rounded to 256 B, skewed by one group's worth of banks
    size_t stride = (((size_t)std::max(c->N / 2, 2) * sizeof(T)) + 255) & ~(size_t)255;
    if (L > 1) stride += ((size_t)G * sizeof(T)) % 256;
    p.slot_stride = (int)stride;
    size_t lds = stride * L + (size_t)2 * L * 2 * c->Nw * 4 + 32 * 4;
    if (lds > 160 * 1024) {
        polar_set_error("SCL decoder: N=%d L=%d needs %zu B of LDS per wavefront (limit 163840)", c->N, L, lds);
        return POLAR_EUNSUPPORTED;
    }
    auto kern = scl_decode_kernel<T, L>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds));
    if (occ < 1) occ = 1;
    int64_t grid = std::min<int64_t>(B, (int64_t)ctx->num_cus * occ * 4);
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, ctx->stream, p, (const uint32_t *)c->d_sched[0]);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "scl_decode_kernel<%s,%d>", sizeof(T) == 4 ? "float" : "double", L);
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = (int)lds;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <typename T>
static int launch_scl_l(polar_code *c, const void *llr, int L, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B,
                       bool crc, const uint32_t *truth)
{
    switch (L) {
    case 1: return launch_scl_t<T, 1>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 2: return launch_scl_t<T, 2>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 4: return launch_scl_t<T, 4>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 8: return launch_scl_t<T, 8>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 16: return launch_scl_t<T, 16>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 32: return launch_scl_t<T, 32>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    default: polar_set_error("SCL decoder: unsupported list size %d", L); return POLAR_EINVAL;
    }
}

int launch_scl_decode(polar_code *c, const void *llr, int dtype, int L, uint32_t *info_hat, double *pm, uint32_t *u_hat_all,
                      int64_t B, bool crc, const uint32_t *truth)
{
    if (B == 0) return POLAR_OK;
    if (c->n > 12) {
        polar_set_error("SCL decoder: N=%d exceeds the supported maximum of 4096", c->N);
        return POLAR_EUNSUPPORTED;
    }
    switch (dispatch_scl(c, dtype, L, pm, u_hat_all)) {
    case KF_SCL_REG: return launch_scl_reg(c, llr, dtype, info_hat, pm, u_hat_all, B, crc, truth, L);
    case KF_SCL_WIDE: return launch_scl_wide(c, llr, dtype, info_hat, pm, u_hat_all, B, L, crc, truth);
    case KF_SCL_LDS:
        if (dtype == POLAR_F32) return launch_scl_l<float>(c, llr, L, info_hat, pm, u_hat_all, B, crc, truth);
        return launch_scl_l<double>(c, llr, L, info_hat, pm, u_hat_all, B, crc, truth);
    default:
        polar_set_error("list decoder: no kernel serves N=%d, K=%d, L=%d, %s (polar_dispatch_table lists what is served)", c->N, c->K, L,
                        dtype == POLAR_F32 ? "fp32" : "fp64");
        return POLAR_EUNSUPPORTED;
    }
}
