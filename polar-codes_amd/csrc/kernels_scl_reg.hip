// Register-resident list decoder, up to L = 8 paths (L = 2 and 4 leave positions empty), N = 64 .. 1024, fp32 or fp64: one
// wavefront per frame, eight lanes per path.
//
// Lane 8p + q belongs to list position p.  Inside a path, element k of every tree level lives in the lane
// whose element index e(q) equals k mod 8 (e = q for q < 4, 11 - q otherwise, so that the three bottom
// exchanges are the DPP patterns row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]).  Levels of
// width >= 16 are per-lane register arithmetic on a slice of N/8 values; partial sums are bit slices.
//
// Forking never copies a wide slice.  Every path keeps, in one 32-bit register, the list position ("owner")
// whose registers hold each of its LLR levels of 32 elements and more and each of its left-child partial-sum levels
// of 256 bits and more.  All paths rewrite a level in lockstep, each into its own registers, so an owner's registers
// stay valid until every path has replaced that level.  Right-child g operations and combines of those levels fetch
// their operands from the owner's lanes with ds_bpermute (or, for a level kept in memory, from the owner's lanes'
// slots).  What is narrow travels with the path instead: a fork or prune moves the metric, the owner row, the live
// values and bits of the 8-leaf subtree being decoded, the two LLRs per lane of the enclosing 16-leaf node and the
// one word that holds the left-child partial sums of every node up to 128 leaves -- so the narrow levels, which are
// walked most often (64 nodes of 16 leaves per 1024), run on a lane's own registers without any fetch or owner entry.
//
// Path metric and ordering follow the reference's sclDecoder / polarSCLFork: PM += 1 - log2(1 + 2^(-+Li));
// below saturation list = originals ++ copies; at saturation the 16 candidates, ordered copies(u=1) ++
// originals(u=0), are ranked by (metric descending, candidate position ascending) and rank r < 8 becomes
// position r.  In fp32 (the throughput mode, statistical parity) the metric uses the overflow-free form with hardware
// exp2/log2; in fp64 (the parity mode) it is evaluated as the reference writes it, operation by operation, as in the
// LDS-staged kernel: the two fp64 kernels give the same bits and the same metrics.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "polar_internal.h"
#include "reg_common.h"

__device__ __forceinline__ float bperm_f(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ double bperm_f(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t bperm_u(uint32_t v, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ uint64_t bperm_u64(uint64_t v, int src_lane)
{
    return ((uint64_t)bperm_u((uint32_t)(v >> 32), src_lane) << 32) | bperm_u((uint32_t)v, src_lane);
}

// partner exchange inside an 8-lane path for a period-P replicated vector
template <int P, typename T> __device__ __forceinline__ T partner8(T v)
{
    if constexpr (P == 8) return dpp_mov<0x141>(v);       // row_half_mirror: lane i <-> 7 - i of each 8
    else if constexpr (P == 4) return dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
    else return dpp_mov<0xB1>(v);                         // quad_perm:[1,0,3,2]
}
template <int P> __device__ __forceinline__ uint32_t partner8_u(uint32_t v)
{
    if constexpr (P == 8) return dpp_mov_u32<0x141>(v);
    else if constexpr (P == 4) return dpp_mov_u32<0x4E>(v);
    else return dpp_mov_u32<0xB1>(v);
}

// value of lane `src` (wave-uniform)
__device__ __forceinline__ float lane_value(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double lane_value(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// integer with the order of the metric (no NaN and no -0 occurs among metrics)
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
__device__ __forceinline__ uint32_t bperm_key(uint32_t k, int byte_addr) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)k); }
__device__ __forceinline__ uint64_t bperm_key(uint64_t k, int byte_addr)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)k);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(k >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// a real number in the word-major exchange records of the pruning step (word w of a record at rec[RS * w]); a word row holds
// the records of the eight list positions and one more that takes the writes of the candidates that do not survive
constexpr int RS = 72;
__device__ __forceinline__ void rec_put(uint32_t *rec, int &w, float v) { rec[RS * w++] = __float_as_uint(v); }
__device__ __forceinline__ void rec_put(uint32_t *rec, int &w, double v)
{
    rec[RS * w++] = (uint32_t)__double2loint(v);
    rec[RS * w++] = (uint32_t)__double2hiint(v);
}
__device__ __forceinline__ void rec_get(const uint32_t *rec, int &w, float &v) { v = __uint_as_float(rec[RS * w++]); }
__device__ __forceinline__ void rec_get(const uint32_t *rec, int &w, double &v)
{
    const uint32_t lo = rec[RS * w++];
    const uint32_t hi = rec[RS * w++];
    v = __hiloint2double((int)hi, (int)lo);
}

// Word w of the record at LDS byte address `addr`, read INTO the register that holds the old value of the same quantity:
// the three exits of a leaf (sure, ranked without a move, moved) then leave every piece of path state in one register and
// meet without copies (as plain loads the new values get registers of their own -- pairs, when two reads are merged -- and
// the other two exits copy theirs over: five moves per leaf).  The caller waits with rec_wait() before using the values.
template <int W> __device__ __forceinline__ void rec_get_into(uint32_t &dst, uint32_t addr)
{
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(4 * RS * W) : "memory");
}
template <int W> __device__ __forceinline__ void rec_get_into(float &dst, uint32_t addr)
{
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(4 * RS * W) : "memory");
}
__device__ __forceinline__ void rec_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// fp64 branch metric: the reference's expression, operation by operation (cn/polarCode.py:982-988 and :2066-2067, as
// scl_decode_kernel<double, L> evaluates it)
__device__ __forceinline__ double branch_metric_fast(double v, uint32_t u)
{
    const double Li = v / 0.693147180559945309417232121458;
    const double e = u ? Li : -Li;
    return 1.0 - log2(1.0 + exp2(e));
}
// fp32 branch metric of deciding u at decision LLR v (overflow-free; -inf only where the reference overflows)
__device__ __forceinline__ float branch_metric_fast(float v, uint32_t u)
{
    float Li = v * 1.44269504088896340736f;  // 1/ln2 (the fp64 parity path divides by ln2 like the reference)
    float e = u ? Li : -Li;
    float m = fmaxf(e, 0.0f);
    float r = 1.0f - (m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(e))));
    return e >= 1024.0f ? -INFINITY : r;
}

__device__ __forceinline__ float fabs_t(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double fabs_t(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float fmin_t(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmin_t(double a, double b) { return __builtin_fmin(a, b); }

// up ? hi + lo : f(hi, lo) without control flow: both values are computed and one is selected (the compiler turns the
// conditional expression around f's inline assembly into a branch on the lanes' own condition, which costs more than f)
__device__ __forceinline__ float sum_or_f(bool up, float x, float o)
{
    const float s = x + o, f = f_fast(x, o);
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(s), "s"(__builtin_amdgcn_ballot_w64(up)));
    return r;
}
__device__ __forceinline__ double sum_or_f(bool up, double x, double o) { return up ? x + o : f_fast(x, o); }

constexpr int SCL_SLAB_SLOTS = 8192;  // slabs for the level in global memory: at least twice the resident workgroups

// Small per-path state that moves between lane groups at a fork.
template <typename T> struct PathState {
    T pm;
    uint32_t ownL;   // 3 bits per level l (index l - 3) for l >= 5: position whose registers hold my level-l LLRs; bits 24..26: see XB;
                     // bits OPB8.. / OPB9..: position that holds my left-child partial sums of level 8 / 9 (the narrower ones travel)
    T v[3];          // my element of the LLRs of the enclosing 8-, 4- and 2-leaf nodes, until their g has run
    // bit XB + I of ownL: partial-sum bit of the finished left child of the enclosing node I (one word less to move at a fork)
};

constexpr int XB = 24;
constexpr int OPB8 = 21, OPB9 = 27;  // bit positions in ownL of the partial-sum owners of levels 8 and 9
constexpr int EAGER_P_TOP = 7;       // left-child partial sums of levels 3 .. 7 (bits 1 .. 31 of a0) travel with the path
#ifndef POLAR_SCL_EAGER_L_TOP
#define POLAR_SCL_EAGER_L_TOP 3
#endif
constexpr int EAGER_L_TOP = POLAR_SCL_EAGER_L_TOP;  // so do the LLRs of levels 3 (as v[0]) and, with 4, of level 4

// Largest subtrees (log2 of their leaves) decoded as one unit: all-frozen ones, and all-information ones every path is sure
// of.  The register allocation of the N = 1024 kernel sits at its limit of 128; larger units (up to 6 is written) spill.
#ifndef POLAR_SCL_FROZEN_TOP
#define POLAR_SCL_FROZEN_TOP 4
#endif
#ifndef POLAR_SCL_SURE_TOP
#define POLAR_SCL_SURE_TOP 3
#endif


// -DPOLAR_SCL_PROFILE: per-category wall-clock cycles of the register list decoder, summed over wavefronts
// (development aid; scripts/gpu_scl_prof.sh builds a second library with it)
#ifdef POLAR_SCL_PROFILE
__device__ unsigned long long g_scl_prof[64];
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[24] = {}
#define PROF_ADD(cat)                                              \
    do {                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof_acc[cat] += now_ - prof_t;                            \
        prof_t = now_;                                             \
    } while (0)
#define PROF_ARGS , unsigned long long &prof_t, unsigned long long *prof_acc
#define PROF_PASS , prof_t, prof_acc
#define PROF_COUNT(cat)                                                    \
    do {                                                                   \
        if (lc.lane == 0) atomicAdd(&g_scl_prof[cat], 1ull);               \
    } while (0)
#else
#define PROF_COUNT(cat)
#define PROF_DECL
#define PROF_ADD(cat)
#define PROF_ARGS
#define PROF_PASS
#endif

struct LaneConst {
    int lane, p, q, e;
    bool up[3];          // P = 8, 4, 2: my element is in the upper half
    uint32_t lowmask[3];
    // ranking of the 16 candidates of a saturated list: my candidate is (p, u = q >> 2); quad lane k = q & 3 compares
    // it with candidates (2k, 0), (2k, 1), (2k+1, 0), (2k+1, 1)
    int rsrc0;           // byte address (lane * 4) of the lane that holds the first of those four metrics; the others follow 16 bytes apart
    uint32_t rsub[4];    // 1 if the other candidate precedes mine in the candidate order copies(u=1) ++ originals(u=0)
};

// At N = 1024 a path's slice is 128 LLRs per lane; with all of them in registers two wavefronts fit on a SIMD, and the
// kernel is bound by the latency of one wavefront's instruction stream (its rate is proportional to the wavefronts
// per CU).  The two widest levels therefore live elsewhere: level 9 (64 values per lane, touched by four steps per
// frame) in a per-workgroup slab of global memory, level 8 (32 values) in LDS, each lane-major ([j][lane]) so that a
// step is 64 coalesced lanes; the registers keep levels 3 .. 7 and a third wavefront fits.  fp64 values take two
// registers each, so that mode splits one level earlier (from N = 512 on): levels 3 .. 6 in registers, level 7 in
// LDS, levels 8 and 9 in the slab.
template <int NLOG, typename T> struct SclStore {
    static constexpr int TOP = NLOG - 1;  // widest stored level
    static constexpr bool SPLIT = sizeof(T) == 4 ? NLOG == 10 : NLOG >= 9;
    static constexpr int REG_TOP = !SPLIT ? TOP : (sizeof(T) == 4 ? 7 : 6);
    static constexpr int LDS_LEVEL = SPLIT ? REG_TOP + 1 : 99;
    static constexpr int GLOBAL_MIN = SPLIT ? REG_TOP + 2 : 99;   // levels from here up live in the slab
    static constexpr int REG_VALUES = 1 << (REG_TOP - 2);         // level l at offset 2^(l-3)
    static constexpr int LDS_VALUES = SPLIT ? (1 << (LDS_LEVEL - 3)) * 64 : 1;  // values per wavefront
    static constexpr int glb_row(int l)                           // first slab row (64 lanes) of level l
    {
        int o = 0;
        for (int i = GLOBAL_MIN; i < l; i++) o += 1 << (i - 3);
        return o;
    }
    static constexpr int GLOBAL_VALUES = SPLIT ? glb_row(TOP + 1) * 64 : 0;     // values per wavefront
};

template <int NLOG, int LMAX, typename T> struct SclReg {
    using St = SclStore<NLOG, T>;
    static constexpr int S = St::REG_VALUES;
    T R[S];               // level l (3 <= l <= REG_TOP) kept in registers: 2^(l-3) values at offset 2^(l-3)
    T *lv_lds;            // the LDS level of this wavefront (wave-uniform)
    T *lv_glb;            // the slab of this wavefront (wave-uniform)
    int my_lane;

    // element j of my level L / of the level-L values that the lanes `src` (owner position * 8 + q) hold / store.
    // Accesses to the levels in memory take the lane number through an opaque copy made inside the step (`lq`): their
    // addresses are then formed there, base plus instruction offset, instead of being hoisted out of the frame loop
    // as one 64-bit register pair per row.
    static __device__ __forceinline__ int opaque(int v)
    {
        asm volatile("" : "+v"(v));
        return v;
    }
    // the same for the frame's channel pointer in fp64, whose strided elements reach beyond the offset field of a load
    static __device__ __forceinline__ const T *opaque(const T *ptr)
    {
        if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(ptr));
        return ptr;
    }
    template <int L> __device__ __forceinline__ T ld_own(int j, int lq) const
    {
        if constexpr (L >= St::GLOBAL_MIN) return lv_glb[(St::glb_row(L) + j) * 64 + lq];
        else if constexpr (L == St::LDS_LEVEL) return lv_lds[j * 64 + lq];
        else return R[(1 << (L - 3)) + j];
    }
    template <int L> __device__ __forceinline__ T ld_from(int j, int src) const
    {
        if constexpr (L >= St::GLOBAL_MIN) return lv_glb[(St::glb_row(L) + j) * 64 + src];
        else if constexpr (L == St::LDS_LEVEL) return lv_lds[j * 64 + src];
        else return bperm_f(R[(1 << (L - 3)) + j], src);
    }
    template <int L> __device__ __forceinline__ void st_own(int j, T v, int lq)
    {
        if constexpr (L >= St::GLOBAL_MIN) lv_glb[(St::glb_row(L) + j) * 64 + lq] = v;
        else if constexpr (L == St::LDS_LEVEL) lv_lds[j * 64 + lq] = v;
        else R[(1 << (L - 3)) + j] = v;
    }
    // after a level held in memory was written and before other lanes read it
    template <int L> static __device__ __forceinline__ void level_written()
    {
        if constexpr (L >= St::GLOBAL_MIN) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else if constexpr (L == St::LDS_LEVEL) {
            wave_sync();
        }
    }
    // plane 0 partial sums: levels with <= 16 bits per lane in a0 (a level of C bits at bit offset C), the 32-bit level in
    // a0h, the 64-bit level in a1; 32-bit words where the levels allow it: a word of its own moves with a path at a prune,
    // and 64-bit shifts cost several instructions
    uint32_t a0, a0h;
    uint64_t a1;
    uint32_t b0, b0h;     // plane 1
    uint64_t b1;
    uint64_t r0, r1;      // root (S bits)
    PathState<T> st;
    int Lcur;
    static constexpr int Lmax = LMAX;   // list size 2, 4 or 8: positions >= Lmax stay empty

    template <int C> static __device__ __forceinline__ uint64_t get_bits(uint32_t w0, uint32_t w0h, uint64_t w1)
    {
        if constexpr (C == 64) return w1;
        else if constexpr (C == 32) return (uint64_t)w0h;
        else return (uint64_t)((w0 >> C) & ((1u << C) - 1u));
    }
    template <int C> static __device__ __forceinline__ void set_bits(uint32_t &w0, uint32_t &w0h, uint64_t &w1, uint64_t x)
    {
        if constexpr (C == 64) {
            w1 = x;
        } else if constexpr (C == 32) {
            w0h = (uint32_t)x;
        } else {
            const uint32_t m = ((1u << C) - 1u) << C;
            w0 = (w0 & ~m) | ((uint32_t)x << C);
        }
    }
    static __device__ __forceinline__ uint32_t own_get(uint32_t row, int lev) { return (row >> (3 * (lev - 3))) & 7u; }
    static __device__ __forceinline__ uint32_t own_set(uint32_t row, int lev, uint32_t p)
    {
        return (row & ~(7u << (3 * (lev - 3)))) | (p << (3 * (lev - 3)));
    }
    // owner of the left-child partial sums of level lev (8 or 9: the narrower levels are the lane's own)
    static __device__ __forceinline__ uint32_t ownp_get(uint32_t row, int lev) { return (row >> (lev == 8 ? OPB8 : OPB9)) & 7u; }
    static __device__ __forceinline__ uint32_t ownp_set(uint32_t row, int lev, uint32_t p)
    {
        const int sh = lev == 8 ? OPB8 : OPB9;
        return (row & ~(7u << sh)) | (p << sh);
    }

    // Both steps read their operands in batches of up to 16 pairs before the first result of the batch is stored: loads
    // that follow a store to memory the compiler cannot tell apart would otherwise wait for it one by one (64 dependent
    // round trips at the top level instead of 4).
    // f: level LEV -> LEV-1 from my own values (or the channel at the top)
    static constexpr int BATCH = sizeof(T) == 4 ? 16 : 8;  // pairs in flight: 32 registers of operands either way
    template <int LEV> __device__ __forceinline__ void step_f(const T *chan_lane, const LaneConst &lc)
    {
        constexpr int c = 1 << (LEV - 4);
        constexpr bool MEM = LEV == NLOG || LEV >= St::GLOBAL_MIN || LEV == St::LDS_LEVEL;
        constexpr int BT = !MEM ? 1 : c < BATCH ? c : BATCH;
        const int lq = opaque(my_lane);
        if constexpr (LEV == NLOG) chan_lane = opaque(chan_lane);
#pragma unroll
        for (int j0 = 0; j0 < c; j0 += BT) {
            T a[BT], b[BT];
#pragma unroll
            for (int t = 0; t < BT; t++) {
                if constexpr (LEV == NLOG) {
                    a[t] = chan_lane[8 * (j0 + t)];
                    b[t] = chan_lane[8 * (j0 + t + c)];
                } else {
                    a[t] = ld_own<LEV>(j0 + t, lq);
                    b[t] = ld_own<LEV>(c + j0 + t, lq);
                }
            }
#pragma unroll
            for (int t = 0; t < BT; t++) st_own<LEV - 1>(j0 + t, f_med(a[t], b[t]), lq);
            if constexpr (MEM && sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);  // one batch of operands in registers at a time
        }
        level_written<LEV - 1>();
        if constexpr (LEV - 1 > EAGER_L_TOP) st.ownL = own_set(st.ownL, LEV - 1, (uint32_t)lc.p);
    }
    // g: operands and the left child's partial sums come from their owners' lanes (levels that travel with the path: from mine)
    template <int LEV> __device__ __forceinline__ void step_g(const T *chan_lane, const LaneConst &lc)
    {
        constexpr int c = 1 << (LEV - 4);
        constexpr bool MEM = LEV == NLOG || LEV >= St::GLOBAL_MIN || LEV == St::LDS_LEVEL;
        constexpr int BT = MEM ? (c < BATCH ? c : BATCH) : (c < 4 ? c : 4);  // register levels: four pairs of lane gathers in flight
        constexpr bool OWN_L = LEV <= EAGER_L_TOP && LEV != NLOG;  // the operands are in my own registers
        const int srcL = opaque((LEV == NLOG || OWN_L) ? 0 : (int)own_get(st.ownL, LEV) * 8 + lc.q);
        const int lq = opaque(my_lane);
        if constexpr (LEV == NLOG) chan_lane = opaque(chan_lane);
        uint64_t bits;
        if constexpr (LEV - 1 <= EAGER_P_TOP) {
            bits = (uint64_t)((a0 >> c) & ((1u << c) - 1u));
        } else {
            const int srcP = (int)ownp_get(st.ownL, LEV - 1) * 8 + lc.q;
            if constexpr (c == 64) bits = bperm_u64(a1, srcP);
            else bits = (uint64_t)bperm_u(a0h, srcP);
        }
        // operands are level LEV, results level LEV-1: no lane overwrites what another still reads
#pragma unroll
        for (int j0 = 0; j0 < c; j0 += BT) {
            T a[BT], b[BT];
#pragma unroll
            for (int t = 0; t < BT; t++) {
                if constexpr (LEV == NLOG) {
                    a[t] = chan_lane[8 * (j0 + t)];
                    b[t] = chan_lane[8 * (j0 + t + c)];
                } else if constexpr (OWN_L) {
                    a[t] = ld_own<LEV>(j0 + t, lq);
                    b[t] = ld_own<LEV>(c + j0 + t, lq);
                } else {
                    a[t] = ld_from<LEV>(j0 + t, srcL);
                    b[t] = ld_from<LEV>(c + j0 + t, srcL);
                }
            }
#pragma unroll
            for (int t = 0; t < BT; t++) st_own<LEV - 1>(j0 + t, g_fast(a[t], b[t], (uint32_t)(bits >> (j0 + t)) & 1u), lq);
            if constexpr (MEM && sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0);  // one batch of operands in registers at a time
        }
        level_written<LEV - 1>();
        if constexpr (LEV - 1 > EAGER_L_TOP) st.ownL = own_set(st.ownL, LEV - 1, (uint32_t)lc.p);
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane, const LaneConst &lc)
    {
        constexpr int c = 1 << (LEV - 4);  // bits per lane of each child
        uint64_t xl;
        if constexpr (LEV - 1 <= EAGER_P_TOP) {
            xl = (uint64_t)((a0 >> c) & ((1u << c) - 1u));
        } else {
            const int srcP = (int)ownp_get(st.ownL, LEV - 1) * 8 + lc.q;
            if constexpr (c == 64) xl = bperm_u64(a1, srcP);
            else xl = (uint64_t)bperm_u(a0h, srcP);
        }
        const uint64_t xr = get_bits<c>(b0, b0h, b1);
        if constexpr (LEV == NLOG) {
            if constexpr (c == 64) {
                r0 = xl ^ xr;
                r1 = xr;
            } else {
                r0 = (xl ^ xr) | (xr << c);
                r1 = 0;
            }
        } else {
            const uint64_t x = (c == 64) ? 0 : ((xl ^ xr) | (xr << c));
            if (plane) {
                set_bits<2 * c>(b0, b0h, b1, x);
            } else {
                set_bits<2 * c>(a0, a0h, a1, x);
                if constexpr (LEV > EAGER_P_TOP) st.ownL = ownp_set(st.ownL, LEV, (uint32_t)lc.p);
            }
        }
    }

    // ---- fork / prune: position lc.p becomes (parent position `par`, bit b) ------------------------------
    // LIVE: bit I set = the g of the enclosing node I is still to come, so its saved LLR travels with the path
    template <uint32_t LIVE> __device__ __forceinline__ void take_over(T cpm, int par, uint32_t b, const LaneConst &lc)
    {
        const int src = par * 8 + lc.q;
        st.pm = bperm_f(cpm, par * 8 + 4 * (int)b + (lc.q & 3));  // lanes q >= 4 hold the u=1 candidate metric
        st.ownL = bperm_u(st.ownL, src);
        a0 = bperm_u(a0, src);
        if constexpr (EAGER_L_TOP >= 4) {
            R[2] = bperm_f(R[2], src);
            R[3] = bperm_f(R[3], src);
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
            if ((LIVE >> i) & 1u) st.v[i] = bperm_f(st.v[i], src);
    }

    // One leaf.  v is the decision LLR, replicated over the 8 lanes of each path.  Returns the decided bit of
    // the path that occupies this list position afterwards.
    template <uint32_t LIVE> __device__ __forceinline__ uint32_t leaf(T v, uint32_t is_info, const LaneConst &lc, uint32_t *surv)
    {
        if (!is_info) {
            st.pm += branch_metric_fast(v, 0u);
            return 0u;
        }
        const uint32_t bme = (uint32_t)lc.q >> 2;  // lanes 0..3 evaluate u = 0, lanes 4..7 evaluate u = 1
        const T cpm = st.pm + branch_metric_fast(v, bme);
        int par;
        uint32_t b;
        if (2 * Lcur <= Lmax) {
            const bool is_copy = lc.p >= Lcur && lc.p < 2 * Lcur;
            par = is_copy ? lc.p - Lcur : lc.p;
            b = is_copy ? 1u : 0u;
            Lcur *= 2;
        } else {
            // 16 candidates (p, u); my candidate is (lc.p, bme); my quad lane k compares it with 4 of the 16
            // the candidates of empty positions (list sizes below 8) rank behind every real one
            const bool alive = LMAX == 8 || lc.p < Lcur;
            const T rpm = alive ? cpm : (T)-INFINITY;
            const int cpos = (bme ? lc.p : 8 + lc.p) + (alive ? 0 : 16);  // copies (u=1) come first in the candidate order
            const int k = lc.q & 3;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int op = 2 * k + (i >> 1), ob = i & 1;
                const T opm = bperm_f(rpm, op * 8 + 4 * ob);
                const int opos = (ob ? op : 8 + op) + ((LMAX == 8 || op < Lcur) ? 0 : 16);
                cnt += (opm > rpm || (opm == rpm && opos < cpos)) ? 1 : 0;
            }
            cnt += (int)dpp_mov_u32<0xB1>((uint32_t)cnt);
            cnt += (int)dpp_mov_u32<0x4E>((uint32_t)cnt);
            if (k == 0 && cnt < Lmax) surv[cnt] = (uint32_t)(lc.p * 8 + 4 * (int)bme);
            wave_sync();
            const uint32_t s = (LMAX == 8 || lc.p < Lmax) ? surv[lc.p] : (uint32_t)(lc.p * 8);
            wave_sync();
            par = (int)(s >> 3);
            b = (s >> 2) & 1u;
        }
        take_over<LIVE>(cpm, par, b, lc);
        return b;
    }

    // The same leaf once the list is full (every later information bit prunes 2 Lmax candidates to Lmax), as
    // straight-line code with as few dependent trips through the LDS crossbar as the data flow allows:
    //  0. leaves that every path is sure of are recognised from the LLRs and the metrics alone (see the block below);
    //  1. the four metrics a lane ranks its candidate against arrive together;
    //  2. when every path's better candidate keeps the path's own list position (the common case at reliable bits)
    //     nothing moves between lane groups: a wave-uniform test on the ranks ends the leaf there;
    //  3. otherwise every lane writes its per-path state, once per surviving child, into the record of the child's new
    //     position (xch[word][position * 8 + q]) and reads the record of its own position back: one write and one read
    //     instead of a survivor table followed by gathers from the parent's lanes.
    template <uint32_t LIVE> __device__ __forceinline__ uint32_t leaf_sat(T v, const LaneConst &lc, uint32_t *xch)
    {
        const uint32_t bme = (uint32_t)lc.q >> 2;
        {
            // 0. A leaf every path is sure of ends before any metric is evaluated (sure_exponent in decode_common.h has the
            //    argument): the ranking below would find each agreeing extension at its path's own position and take the exit
            //    of step 2 with these very metrics.
            T ae_lo;
            const bool big = sure_exponent(v, ae_lo);
            const T cpm_a = st.pm + (T)1;
            const T next = bperm_f(cpm_a, lc.lane + 8);  // the agreeing extension of the next list position
            const T other = st.pm + ((T)2 - ae_lo);
            const T last = lane_value(cpm_a, 8 * (LMAX - 1));
            // the three conditions as one expression without short-circuit evaluation: as nested branches on the lanes' own
            // outcomes (what && compiles to) the test cost 4 % of the kernel's time
            const bool sure = (bool)((int)big & (int)(other < last) & ((int)(lc.p == LMAX - 1) | (int)(cpm_a > next)));
#ifdef POLAR_SCL_PROFILE
            {   // why a leaf goes on to the ranking: some LLR small (1), else another extension could survive (2), else order (3)
                const bool dead = LMAX < 8 && lc.p >= LMAX;
                if (__ballot(big || dead) != ~0ull) PROF_COUNT(1);
                else if (__ballot(other < last || dead) != ~0ull) PROF_COUNT(2);
                else if (__ballot(sure || dead) != ~0ull) PROF_COUNT(3);
            }
#endif
            if (__ballot(sure || (LMAX < 8 && lc.p >= LMAX)) == ~0ull) {
                PROF_COUNT(0);
                st.pm = cpm_a;
                return v < (T)0 ? 1u : 0u;
            }
        }
        const T cpm = st.pm + branch_metric_fast(v, bme);
        const bool alive = LMAX == 8 || lc.p < LMAX;
        const T rpm = alive ? cpm : (T)-INFINITY;
        // ranks on integer keys with the order of the metrics: "the other candidate wins" = its key is larger, or
        // equal while it comes first in the candidate order, i.e. key_other > key_mine - before
        const auto key = order_key(rpm);
        decltype(order_key(rpm)) o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = bperm_key(key, lc.rsrc0 + 16 * i);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) cnt += (o[i] > key - lc.rsub[i]) ? 1 : 0;
        cnt += (int)dpp_mov_u32<0xB1>((uint32_t)cnt);
        cnt += (int)dpp_mov_u32<0x4E>((uint32_t)cnt);
        const bool keep = cnt == lc.p;
        constexpr uint64_t EVERY = LMAX == 8 ? 0x0101010101010101ull : LMAX == 4 ? 0x01010101ull : 0x0101ull;
        const uint64_t km = __ballot(keep);
        const T cpm_o = dpp_mov<0x141>(cpm);  // the metric of my path's other candidate
        PROF_COUNT(22);
        if (((km | (km >> 4)) & EVERY) == EVERY) {
            PROF_COUNT(23);
            st.pm = keep ? cpm : cpm_o;
            return (keep ? 1u : 0u) ^ (bme ^ 1u);  // u = 1 survived iff the u = 1 lanes keep / the u = 0 lanes do not
        }
        const int cnt_o = (int)dpp_mov_u32<0x141>((uint32_t)cnt);  // the rank of my path's other candidate
        const int rk[2] = {cnt, cnt_o};
        const T mt[2] = {cpm, cpm_o};
#pragma unroll
        for (int u = 0; u < 2; u++) {  // my own candidate, then my path's other one
            {
                // candidates that do not survive write as well, into the spare record: cheaper than a branch on the lane's rank
                const int slot = rk[u] < Lmax ? rk[u] : 8;
                uint32_t *rec = xch + slot * 8 + lc.q;
                rec[0] = st.ownL;
                rec[RS] = (a0 & ~1u) | (bme ^ (uint32_t)u);  // bit 0 of a0 belongs to no level; u = 0: my own candidate
                int w = 2;
                rec_put(rec, w, mt[u]);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if ((LIVE >> i) & 1u) rec_put(rec, w, st.v[i]);
                if constexpr (EAGER_L_TOP >= 4) {
                    rec_put(rec, w, R[2]);
                    rec_put(rec, w, R[3]);
                }
            }
        }
        wave_sync();
        uint32_t op;
        if constexpr (sizeof(T) == 4) {
            // single reads into the registers that held the old values (see rec_get_into)
            const uint32_t addr = (uint32_t)(uintptr_t)(xch + lc.lane);
            op = a0;
            rec_get_into<0>(st.ownL, addr);
            rec_get_into<1>(op, addr);
            rec_get_into<2>(st.pm, addr);
            constexpr int W0 = 3, W1 = W0 + (int)(LIVE & 1u), W2 = W1 + (int)((LIVE >> 1) & 1u), W3 = W2 + (int)((LIVE >> 2) & 1u);
            if constexpr (LIVE & 1u) rec_get_into<W0>(st.v[0], addr);
            if constexpr (LIVE & 2u) rec_get_into<W1>(st.v[1], addr);
            if constexpr (LIVE & 4u) rec_get_into<W2>(st.v[2], addr);
            if constexpr (EAGER_L_TOP >= 4) {
                rec_get_into<W3>(R[2], addr);
                rec_get_into<W3 + 1>(R[3], addr);
            }
            rec_wait();
        } else {
            const uint32_t *mine = xch + lc.lane;
            st.ownL = mine[0];
            op = mine[RS];
            int w = 2;
            rec_get(mine, w, st.pm);
#pragma unroll
            for (int i = 0; i < 3; i++)
                if ((LIVE >> i) & 1u) rec_get(mine, w, st.v[i]);
            if constexpr (EAGER_L_TOP >= 4) {
                rec_get(mine, w, R[2]);
                rec_get(mine, w, R[3]);
            }
        }
        wave_sync();
        a0 = op & ~1u;
        return op & 1u;
    }

    // A node of P leaves (2, 4 or 8), all frozen: nothing forks and every partial sum is zero, so the P decision LLRs
    // are computed side by side (after the exchange for period Q the lanes of the lower half hold the left child's
    // element, the others the right child's: lane element e ends with the LLR of leaf e mod P), the P metric
    // terms in one go, and the path metric takes them in leaf order, as the leaf-by-leaf walk would.
    template <int P> __device__ __forceinline__ void frozen_node(T v, const LaneConst &lc)
    {
        T x = v;
        if constexpr (P >= 8) {
            const T o = partner8<8>(x);
            x = sum_or_f(lc.up[0], x, o);
        }
        if constexpr (P >= 4) {
            const T o = partner8<4>(x);
            x = sum_or_f(lc.up[1], x, o);
        }
        {
            const T o = partner8<2>(x);
            x = sum_or_f(lc.up[2], x, o);
        }
        const T m = branch_metric_fast(x, 0u);
        T t[P];
#pragma unroll
        for (int i = 0; i < P; i++) t[i] = bperm_f(m, (lc.lane & ~7) + (i < 4 ? i : 11 - i));  // the lane whose element is i
        T pm = st.pm;
#pragma unroll
        for (int i = 0; i < P; i++) pm += t[i];
        st.pm = pm;
    }

    // ---- whole subtrees of 16, 32 or 64 leaves decoded as one unit --------------------------------------------------
    // The partial sums of a finished subtree of level LV (2^(LV-3) bits per lane) go to plane `plane` like a combine's.
    template <int LV> __device__ __forceinline__ void put_sums(int plane, uint64_t x, const LaneConst &lc)
    {
        constexpr int CB = 1 << (LV - 3);
        if (plane) {
            set_bits<CB>(b0, b0h, b1, x);
        } else {
            set_bits<CB>(a0, a0h, a1, x);  // LV <= 6: these sums travel with the path, no owner entry
        }
        (void)lc;
    }
    // All 2^LV leaves frozen: every decision is 0, so nothing forks, no owner changes and no partial sum is ever non-zero:
    // g is hi + lo everywhere.  The decision LLRs of all leaves are computed side by side (register j of the lane ends as
    // the 8-leaf node j of the subtree, whose three bottom stages are frozen_node's exchanges) with the operations the
    // leaf-by-leaf walk performs, and the metric takes the 2^LV terms in leaf order.
    template <int LV> __device__ __forceinline__ void frozen_subtree(int plane, const LaneConst &lc)
    {
        constexpr int C = 1 << (LV - 3);
        T *const x = R + C;  // the subtree's own level: dead once its leaves are decided, so the work happens in place
#pragma unroll
        for (int m = LV; m > 3; m--) {
            const int h = 1 << (m - 4);
#pragma unroll
            for (int blk = 0; blk < C; blk += 2 * h)
#pragma unroll
                for (int t = 0; t < h; t++) {
                    const T a = x[blk + t], b = x[blk + h + t];
                    x[blk + t] = f_med(a, b);
                    x[blk + h + t] = b + a;
                }
        }
#pragma unroll
        for (int j = 0; j < C; j++) {
            T o = partner8<8>(x[j]);
            x[j] = sum_or_f(lc.up[0], x[j], o);
            o = partner8<4>(x[j]);
            x[j] = sum_or_f(lc.up[1], x[j], o);
            o = partner8<2>(x[j]);
            x[j] = sum_or_f(lc.up[2], x[j], o);
            x[j] = branch_metric_fast(x[j], 0u);
        }
        T pm = st.pm;
#pragma unroll
        for (int j = 0; j < C; j++) {
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 4) {
                T t[4];
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = bperm_f(x[j], (lc.lane & ~7) + (i0 + i < 4 ? i0 + i : 11 - i0 - i));
#pragma unroll
                for (int i = 0; i < 4; i++) pm += t[i];
            }
        }
        st.pm = pm;
        put_sums<LV>(plane, 0, lc);
    }
    // All 2^LV leaves carry information and the list is full.  With hard decisions on its LLRs (the decisions SC makes in
    // such a node) every LLR below has at least the magnitude of the smallest of them: f takes a minimum, g adds two values
    // of equal sign.  If that smallest magnitude passes sure_exponent's test in every path, the paths' metrics are apart
    // by more than DELTA in list order, and every path's metric plus (2 - |e|) stays more than DELTA + 1 below the last
    // path's, then the three conditions of leaf_sat's first step hold at each of the 2^LV leaves in turn (the metrics all
    // grow by 1 per leaf; DELTA is far above what 64 rounded additions to numbers below 4096 can drift): no path moves,
    // each metric takes 2^LV additions of 1, and the node's partial sums are the signs of its LLRs.  Exact, like the
    // per-leaf test; returns false (and changes nothing) when the conditions do not hold.
    template <int LV> __device__ __forceinline__ bool rate1_sure(int plane, const LaneConst &lc, uint64_t &xs)
    {
        constexpr int C = 1 << (LV - 3);
        T mn = fabs_t(R[C]);
#pragma unroll
        for (int j = 1; j < C; j++) mn = fmin_t(mn, fabs_t(R[C + j]));
        mn = fmin_t(mn, partner8<8>(mn));
        mn = fmin_t(mn, partner8<4>(mn));
        mn = fmin_t(mn, partner8<2>(mn));
        T ae;
        const bool big = sure_exponent(mn, ae);
        const T DELTA = (T)0.125;
        const T pm = st.pm;
        const T next = bperm_f(pm, lc.lane + 8);
        const T last = lane_value(pm, 8 * (LMAX - 1));
        const bool ok = (bool)((int)big & (int)(fabs_t(pm) < (T)4000) & (int)(pm + ((T)2 - ae) + DELTA < last + (T)1) &
                               ((int)(lc.p == LMAX - 1) | (int)(pm - next > DELTA)));  // no short-circuit: see leaf_sat
        PROF_COUNT(24 + 2 * (LV - 3));
#ifdef POLAR_SCL_PROFILE
        if (LV == 3) {  // why not: smallest LLR too small (32), metric out of range (33), another extension could survive (34), order (35)
            const bool dead = LMAX < 8 && lc.p >= LMAX;
            if (__ballot(big || dead) != ~0ull) PROF_COUNT(32);
            else if (__ballot(fabs_t(pm) < (T)4000 || dead) != ~0ull) PROF_COUNT(33);
            else if (__ballot(pm + ((T)2 - ae) + DELTA < last + (T)1 || dead) != ~0ull) PROF_COUNT(34);
            else if (__ballot(ok || dead) != ~0ull) PROF_COUNT(35);
        }
#endif
        if (__ballot(ok || (LMAX < 8 && lc.p >= LMAX)) != ~0ull) return false;
        PROF_COUNT(25 + 2 * (LV - 3));
        T acc = pm;
#pragma unroll 8
        for (int i = 0; i < (1 << LV); i++) acc += (T)1;
        st.pm = acc;
        uint64_t x = 0;
#pragma unroll
        for (int j = 0; j < C; j++) x |= (uint64_t)(R[C + j] < (T)0 ? 1u : 0u) << j;
        xs = x;
        (void)plane;
        return true;
    }
    // 1 = all frozen, 2 = all information, 0 = mixed, for the node `idx` of level LV (16, 32 or 64 leaves); wave-uniform
    template <int LV> static __device__ __forceinline__ int node_class(uint32_t maskv, uint32_t idx)
    {
        uint32_t lo, hi;
        if constexpr (LV == 4) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(idx >> 1));
            lo = hi = (w >> (16 * (idx & 1))) & 0xFFFFu;
            hi |= 0xFFFF0000u;
        } else if constexpr (LV == 5) {
            lo = hi = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)idx);
        } else {
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(2 * idx));
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(2 * idx + 1));
            lo = w0 | w1;
            hi = w0 & w1;
        }
        return lo == 0u ? 1 : hi == 0xFFFFFFFFu ? 2 : 0;
    }

    // A node of P leaves inside the 8-leaf subtree; v replicated with period P over the path's lanes.
    template <int P, uint32_t LIVE = 0u>
    __device__ __forceinline__ uint32_t sub(T v, uint32_t mask, const LaneConst &lc, uint32_t *surv)
    {
        if constexpr (P == 1) {
            return leaf<LIVE>(v, mask & 1u, lc, surv);
        } else {
            if (mask == 0u) {  // wave-uniform: the rate profile is the same for every path
                frozen_node<P>(v, lc);
                return 0u;
            }
            constexpr int I = P == 8 ? 0 : P == 4 ? 1 : 2;
            constexpr uint32_t HALF = (1u << (P / 2)) - 1u;
            const T o = partner8<P>(v);
            st.v[I] = v;  // forks inside the left child move it along with the path
            const T vl = f_fast(v, o);
            const uint32_t xl = sub<P / 2, LIVE | (1u << I)>(vl, mask & HALF, lc, surv);
            st.ownL = (st.ownL & ~(1u << (XB + I))) | (xl << (XB + I));
            // g: the lanes of the lower half hold lo and apply the sign, then every lane adds its partner's value
            const T t = flip_sign(st.v[I], xl & lc.lowmask[I]);
            const T vr = t + partner8<P>(t);
            const uint32_t xr = sub<P / 2, LIVE>(vr, (mask >> (P / 2)) & HALF, lc, surv);
            const uint32_t xl_now = (st.ownL >> (XB + I)) & 1u;  // the left result of whoever sits here now
            return xr ^ (xl_now & lc.lowmask[I]);
        }
    }

    // The same node with its rate profile known at compile time and the list full: no branch on the profile is left,
    // an all-frozen left half skips the save / restore of the node's LLR (nothing forks inside it).
    template <int P, uint32_t MASK, uint32_t LIVE = 0u>
    __device__ __forceinline__ uint32_t sub_s(T v, const LaneConst &lc, uint32_t *surv)
    {
        if constexpr (MASK == 0u) {
            if constexpr (P == 1) st.pm += branch_metric_fast(v, 0u);
            else frozen_node<P>(v, lc);
            return 0u;
        } else if constexpr (P == 1) {
            return leaf_sat<LIVE>(v, lc, surv);
        } else {
            constexpr int I = P == 8 ? 0 : P == 4 ? 1 : 2;
            constexpr uint32_t HALF = (1u << (P / 2)) - 1u;
            constexpr uint32_t ML = MASK & HALF, MR = (MASK >> (P / 2)) & HALF;
            const T o = partner8<P>(v);
            const T vl = f_fast(v, o);
            if constexpr (ML == 0u) {
                if constexpr (P == 2) st.pm += branch_metric_fast(vl, 0u);
                else frozen_node<P / 2>(vl, lc);
                return sub_s<P / 2, MR, LIVE>(v + o, lc, surv);
            } else {
                st.v[I] = v;
                const uint32_t xl = sub_s<P / 2, ML, LIVE | (1u << I)>(vl, lc, surv);
                st.ownL = (st.ownL & ~(1u << (XB + I))) | (xl << (XB + I));
                const T t = flip_sign(st.v[I], xl & lc.lowmask[I]);
                const T vr = t + partner8<P>(t);
                const uint32_t xr = sub_s<P / 2, MR, LIVE>(vr, lc, surv);
                const uint32_t xl_now = (st.ownL >> (XB + I)) & 1u;
                return xr ^ (xl_now & lc.lowmask[I]);
            }
        }
    }

    // An 8-leaf node: the nine rate profiles a code built from a reliability order can have here (beside all-frozen)
    // run as compiled straight-line code once the list is full, anything else through the generic form.
    __device__ __forceinline__ uint32_t node8(T v, uint32_t m8, const LaneConst &lc, uint32_t *surv)
    {
        if (__builtin_amdgcn_readfirstlane(Lcur) == Lmax) {
            switch (m8) {
            case 0x00u: frozen_node<8>(v, lc); return 0u;
            case 0x80u: return sub_s<8, 0x80u>(v, lc, surv);
            case 0xC0u: return sub_s<8, 0xC0u>(v, lc, surv);
            case 0xE0u: return sub_s<8, 0xE0u>(v, lc, surv);
            case 0xE8u: return sub_s<8, 0xE8u>(v, lc, surv);
            case 0xF0u: return sub_s<8, 0xF0u>(v, lc, surv);
            case 0xF8u: return sub_s<8, 0xF8u>(v, lc, surv);
            case 0xFCu: return sub_s<8, 0xFCu>(v, lc, surv);
            case 0xFEu: return sub_s<8, 0xFEu>(v, lc, surv);
            case 0xFFu: {
                uint64_t xs;
                if (POLAR_SCL_SURE_TOP >= 3 && rate1_sure<3>(0, lc, xs)) return (uint32_t)xs;
                return sub_s<8, 0xFFu>(v, lc, surv);
            }
            default: break;
            }
        }
        return sub<8>(v, m8, lc, surv);
    }
};

template <int LEV, int NLOG, int LMAX, typename T>
__device__ __forceinline__ void scl_node(SclReg<NLOG, LMAX, T> &d, uint32_t maskv, const T *chan_lane, uint32_t idx, int plane,
                                         const LaneConst &lc, uint32_t *surv PROF_ARGS)
{
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
        if (h == 0) d.template step_f<LEV>(chan_lane, lc);
        else d.template step_g<LEV>(chan_lane, lc);
        PROF_ADD(LEV);  // categories 4 .. 10: the f / g steps of that level
        const uint32_t cidx = 2 * idx + h;
        if constexpr (LEV - 1 == 3) {
            const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(cidx >> 2));
            const uint32_t m8 = (mw >> (8 * (cidx & 3))) & 0xFFu;
            const uint32_t x = d.node8(d.R[1], m8, lc, surv);
            PROF_ADD(12 + __builtin_popcount(m8));  // categories 12 .. 20: 8-leaf nodes by their number of information bits
            if (h) {
                d.b0 = (d.b0 & ~2u) | (x << 1);
            } else {
                d.a0 = (d.a0 & ~2u) | (x << 1);
            }
        } else {
            if constexpr (LEV - 1 <= 6 && (LEV - 1 <= POLAR_SCL_FROZEN_TOP || LEV - 1 <= POLAR_SCL_SURE_TOP)) {
                const int cls = SclReg<NLOG, LMAX, T>::template node_class<LEV - 1>(maskv, cidx);
                if (LEV - 1 <= POLAR_SCL_FROZEN_TOP && cls == 1) {
                    d.template frozen_subtree<LEV - 1>((int)h, lc);
                    PROF_ADD(12);
                    continue;
                }
                if (LEV - 1 <= POLAR_SCL_SURE_TOP && cls == 2 && __builtin_amdgcn_readfirstlane(d.Lcur) == LMAX) {
                    uint64_t xs;
                    if (d.template rate1_sure<LEV - 1>((int)h, lc, xs)) {
                        d.template put_sums<LEV - 1>((int)h, xs, lc);
                        PROF_ADD(20);
                        continue;
                    }
                }
            }
            scl_node<LEV - 1, NLOG, LMAX, T>(d, maskv, chan_lane, cidx, (int)h, lc, surv PROF_PASS);
        }
    }
    d.template combine<LEV>(plane, lc);
    PROF_ADD(11);  // combines
}

#ifndef POLAR_SCL64_TWO
#define POLAR_SCL64_TWO 99  // fp64: the length (log2) that runs two wavefronts per SIMD without spilling rather than three with
#endif
// List outputs (pm_out, u_hat_all; either may be NULL) are those of scl_decode_kernel: row f * LMAX + r holds the path of
// rank r in the stable descending order of the metrics, rows of empty positions keep their index, -inf and zeros.
template <int NLOG, int LMAX, typename T>
__global__ void __launch_bounds__(64, sizeof(T) == 4 ? 4 : (NLOG == 8 || NLOG == POLAR_SCL64_TWO) ? 2 : 3)
scl_reg_kernel(const T *__restrict__ llr, const uint32_t *__restrict__ maskw, const uint16_t *__restrict__ Asl8,
               const uint32_t *__restrict__ crcS, const uint32_t *__restrict__ truth, uint32_t *__restrict__ info_hat, int64_t B, int K,
               int Kw, void *__restrict__ slab, double *__restrict__ pm_out, uint32_t *__restrict__ u_hat_all, uint32_t slot_words,
               uint32_t *__restrict__ fault)
{
    // exchange records of the pruning step (8 words per lane in fp32, 14 in fp64, word-major); its first words double as the
    // survivor table of the general leaf, and the whole area as the staging buffer of the output stage (8 paths x 8 lanes x
    // 128-bit slice of u_hat)
    constexpr int XW = 2 + (EAGER_L_TOP >= 4 ? 6 : 4) * (int)(sizeof(T) / 4);
    __shared__ uint32_t xchg[RS * XW];
    uint32_t *const surv = xchg;
    uint32_t *const stage = xchg;
    constexpr int N = 1 << NLOG;
    // the gather table of the output stage (the CRC remainders stay in global memory); at N = 1024, where LDS decides how
    // many wavefronts a CU holds, it is read from global memory instead
    constexpr bool ASL_LDS = NLOG < 10;
    __shared__ uint16_t asl_s[ASL_LDS ? N : 1];
    using St = SclStore<NLOG, T>;
    __shared__ T lvl_s[St::LDS_VALUES];
    if constexpr (ASL_LDS) {
        for (int k = threadIdx.x; k < K; k += 64) asl_s[k] = Asl8[k];
        wave_sync();
    }
    LaneConst lc;
    lc.lane = threadIdx.x;
    lc.p = lc.lane >> 3;
    lc.q = lc.lane & 7;
    lc.e = lc.q < 4 ? lc.q : 11 - lc.q;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        lc.up[i] = (lc.e & (4 >> i)) != 0;
        lc.lowmask[i] = lc.up[i] ? 0u : 1u;
    }
    {
        const int bme = lc.q >> 2, k = lc.q & 3, cpos = bme ? lc.p : 8 + lc.p;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int op = 2 * k + (i >> 1), ob = i & 1;
            if (i == 0) lc.rsrc0 = (op * 8 + 4 * ob) << 2;  // candidate i sits 4 lanes (16 bytes) further: (2k + i / 2) * 8 + 4 (i & 1)
            // candidates of empty positions (list sizes below 8) come after every real one
            const int opos = (ob ? op : 8 + op) + ((LMAX == 8 || op < LMAX) ? 0 : 16);
            lc.rsub[i] = opos < cpos + ((LMAX == 8 || lc.p < LMAX) ? 0 : 16) ? 1u : 0u;
        }
    }
    const uint32_t maskv = maskw[(lc.lane & 31) < (N / 32) ? (lc.lane & 31) : 0];
    SclReg<NLOG, LMAX, T> d;
    d.my_lane = lc.lane;
    d.lv_lds = lvl_s;
    // The level in global memory (N = 1024) lives in one of the launch's slabs (a power of two, at least twice the
    // workgroups that can be resident: slot_words = slabs / 32), claimed here and given back at the end: the slabs in use are
    // as many as there are resident workgroups, however long the grid is.  slab[0 .. SCL_SLAB_SLOTS / 32) is the bitmap of
    // claimed slabs (cleared before the launch).  With that many slabs the search ends within one sweep; a search that has
    // looked at every word eight times without a claim (a bitmap that was not cleared, a grid beyond what the host sized)
    // gives up: the workgroup reports through *fault and decodes nothing, instead of spinning for ever.
    uint32_t slab_word = 0, slab_bit = 0;
    if constexpr (St::GLOBAL_VALUES > 0) {
        uint32_t *bm = reinterpret_cast<uint32_t *>(slab);
        if (lc.lane == 0) {
            uint32_t w = (blockIdx.x * 5u) & (slot_words - 1u);
            xchg[1] = 0xFFFFFFFFu;
            for (uint32_t tries = 0; tries < 8u * slot_words; tries++) {
                const uint32_t cur = __hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur != 0xFFFFFFFFu) {
                    const uint32_t b = (uint32_t)__ffs((int)~cur) - 1u;
                    if (!(atomicOr(bm + w, 1u << b) & (1u << b))) {
                        xchg[0] = w;
                        xchg[1] = b;
                        break;
                    }
                } else {
                    w = (w + 1u) & (slot_words - 1u);
                }
            }
            if (xchg[1] == 0xFFFFFFFFu) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        wave_sync();
        slab_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)xchg[0]);
        slab_bit = (uint32_t)__builtin_amdgcn_readfirstlane((int)xchg[1]);
        wave_sync();
        if (slab_bit == 0xFFFFFFFFu) return;  // wave-uniform
        d.lv_glb = reinterpret_cast<T *>(reinterpret_cast<char *>(slab) + SCL_SLAB_SLOTS / 8) + (size_t)(slab_word * 32u + slab_bit) * St::GLOBAL_VALUES;
    } else {
        d.lv_glb = reinterpret_cast<T *>(slab);
    }

    for (int64_t f = blockIdx.x; f < B; f += gridDim.x) {
        const T *chan_lane = llr + f * N + lc.e;
        d.a0 = d.a0h = d.b0 = d.b0h = 0u;
        d.a1 = d.b1 = d.r0 = d.r1 = 0;
        d.st.pm = (T)0;
        d.st.ownL = 0u;
#pragma unroll
        for (int i = 0; i < 3; i++) d.st.v[i] = (T)0;
        d.Lcur = 1;
        PROF_DECL;
        scl_node<NLOG, NLOG, LMAX, T>(d, maskv, chan_lane, 0u, 0, lc, surv PROF_PASS);

        // final stable descending order: rank of every position; the best path is rank 0
        const bool act = lc.p < d.Lcur;
        const T spm = act ? d.st.pm : (T)-INFINITY;
        const int spos = act ? lc.p : 64;
        int rank = 0;
        for (int c = 0; c < 8; c++) {
            const T opm = lane_value(spm, c * 8);
            const int opos = __builtin_amdgcn_readlane(spos, c * 8);
            rank += (opm > spm || (opm == spm && opos < spos)) ? 1 : 0;
        }
        if (!act) rank = lc.p;  // rows of empty positions keep their index (all >= Lcur)

        // u_hat = x_hat F^{(x)n} on the sliced layout: element e + 8 j is bit j of my 128-bit root slice
        uint64_t u0 = d.r0, u1 = d.r1;
        if constexpr (NLOG > 3) { u0 ^= (u0 >> 1) & 0x5555555555555555ull; u1 ^= (u1 >> 1) & 0x5555555555555555ull; }
        if constexpr (NLOG > 4) { u0 ^= (u0 >> 2) & 0x3333333333333333ull; u1 ^= (u1 >> 2) & 0x3333333333333333ull; }
        if constexpr (NLOG > 5) { u0 ^= (u0 >> 4) & 0x0F0F0F0F0F0F0F0Full; u1 ^= (u1 >> 4) & 0x0F0F0F0F0F0F0F0Full; }
        if constexpr (NLOG > 6) { u0 ^= (u0 >> 8) & 0x00FF00FF00FF00FFull; u1 ^= (u1 >> 8) & 0x00FF00FF00FF00FFull; }
        if constexpr (NLOG > 7) { u0 ^= (u0 >> 16) & 0x0000FFFF0000FFFFull; u1 ^= (u1 >> 16) & 0x0000FFFF0000FFFFull; }
        if constexpr (NLOG > 8) { u0 ^= (u0 >> 32) & 0x00000000FFFFFFFFull; u1 ^= (u1 >> 32) & 0x00000000FFFFFFFFull; }
        if constexpr (NLOG > 9) u0 ^= u1;
        uint32_t w[4] = {(uint32_t)u0, (uint32_t)(u0 >> 32), (uint32_t)u1, (uint32_t)(u1 >> 32)};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t t;
            t = partner8_u<2>(w[i]); w[i] ^= (lc.e & 1) ? 0u : t;
            t = partner8_u<4>(w[i]); w[i] ^= (lc.e & 2) ? 0u : t;
            t = partner8_u<8>(w[i]); w[i] ^= (lc.e & 4) ? 0u : t;
            stage[lc.lane * 4 + i] = w[i];
        }
        wave_sync();
        if (pm_out && lc.q == 0 && lc.p < LMAX) pm_out[f * LMAX + rank] = (double)spm;
        if (u_hat_all && lc.p < LMAX) {
            // natural-order words of my path's u_hat: element 8 j + e is bit j of the slice of the lane whose element is e
            uint32_t *row = u_hat_all + ((size_t)f * LMAX + rank) * (N / 32);
            for (int wi = lc.q; wi < N / 32; wi += 8) {
                uint32_t word = 0u;
#pragma unroll 8
                for (int t = 0; t < 32; t++) {
                    const int e = t & 7, j = 4 * wi + (t >> 3);
                    const uint32_t sw = stage[(lc.p * 8 + (e < 4 ? e : 11 - e)) * 4 + (j >> 5)];
                    word |= ((sw >> (j & 31)) & 1u) << t;
                }
                row[wi] = act ? word : 0u;
            }
        }
        if (info_hat) {
            // Only one path's information bits leave.  Plain decoding: the best path's; aided decoding: those of the first
            // path in metric order that passes (CRC: the information word leaves a zero remainder; genie: it equals the
            // transmitted word), the best one if none does.  The paths are tried in that order, each gathered by all 64
            // lanes, so a frame whose best path passes -- nearly every frame above the waterfall -- costs one gather.
            uint32_t *dst = info_hat + f * Kw;
            const int tries = (crcS || truth) ? d.Lcur : 1;
            uint32_t wmine = 0u, wbest = 0u;
            for (int r = 0; r < tries; r++) {
                const uint64_t bm = __ballot(act && rank == r);
                const uint32_t *cand = stage + ((__ffsll((unsigned long long)bm) - 1) >> 3) * 32;
                uint32_t syn = 0u;  // this lane's share of the path's CRC remainder
                wmine = 0u;
                pack_bits_group<64>(
                    K, lc.lane, 0,
                    [&](int k) {
                        const uint32_t a = ASL_LDS ? asl_s[k] : Asl8[k];  // (lane-in-path << 7) | bit
                        const bool bit = ((cand[(a >> 7) * 4 + ((a >> 5) & 3u)] >> (a & 31u)) & 1u) != 0u;
                        if (crcS && bit) syn ^= crcS[k];
                        return bit;
                    },
                    [&](int wi, uint32_t v) {
                        if (wi == lc.lane) wmine = v;
                    });
                if (r == 0) wbest = wmine;
                bool ok = true;
                if (truth) {
                    ok = __ballot(lc.lane < Kw && wmine != truth[f * Kw + lc.lane]) == 0;
                } else if (crcS) {
                    syn ^= dpp_mov_u32<0xB1>(syn);   // quad_perm:[1,0,3,2]
                    syn ^= dpp_mov_u32<0x4E>(syn);   // quad_perm:[2,3,0,1]
                    syn ^= dpp_mov_u32<0x141>(syn);  // row_half_mirror
                    syn ^= dpp_mov_u32<0x140>(syn);  // row_mirror: every lane now holds its row's sum
                    const uint32_t all = (uint32_t)(__builtin_amdgcn_readlane((int)syn, 0) ^ __builtin_amdgcn_readlane((int)syn, 16) ^
                                                    __builtin_amdgcn_readlane((int)syn, 32) ^ __builtin_amdgcn_readlane((int)syn, 48));
                    ok = all == 0u;
                }
                if (ok) break;                       // wave-uniform
                if (r == tries - 1) wmine = wbest;   // none passes
            }
            if (lc.lane < Kw) dst[lc.lane] = wmine;
            if (K == 0 && lc.lane == 0) dst[0] = 0u;
        }
        wave_sync();
#ifdef POLAR_SCL_PROFILE
        PROF_ADD(21);  // output stage
        if (lc.lane == 0)
            for (int i = 0; i < 24; i++) atomicAdd(&g_scl_prof[i], prof_acc[i]);
#endif
    }
    if constexpr (St::GLOBAL_VALUES > 0) {
        // every store to the slab has landed before the next owner can claim it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        if (lc.lane == 0) atomicAnd(reinterpret_cast<uint32_t *>(slab) + slab_word, ~(1u << slab_bit));
    }
}

#ifdef POLAR_SCL_PROFILE
extern "C" __attribute__((visibility("default"))) int polar_debug_scl_profile(unsigned long long *out, int clear)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scl_prof), sizeof(unsigned long long) * 40) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[64] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scl_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------ host side
bool scl_reg_supported(const polar_code *c, int dtype, int L, const void *pm, const void *u_hat_all)
{
    return dispatch_scl(c, dtype, L, pm, u_hat_all) == KF_SCL_REG;  // the table of host_tables.cpp says what this kernel serves
}

template <int NLOG, int LMAX, typename T>
static int launch_scl_reg_t(polar_code *c, const void *llr, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                            const uint32_t *truth)
{
    polar_ctx *ctx = c->ctx;
    using St = SclStore<NLOG, T>;
    auto kern = scl_reg_kernel<NLOG, LMAX, T>;
    // POLAR_SCL_LDSPAD (bytes of unused dynamic LDS per wavefront) lowers the occupancy for experiments
    const char *pe = polar_switch("POLAR_SCL_LDSPAD");
    const size_t pad = pe ? (size_t)atoi(pe) : 0;
    if (pad > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, pad));
    if (occ < 1) occ = 1;
    int rounds = 32;  // workgroups per resident slot: 14.9 / 16.2 / 16.8 / 17.25 M frames/s at (1024,512), L = 8, fp32, with 1 / 4 / 8 / 32
    if (const char *e = polar_switch("POLAR_SCL_ROUNDS")) rounds = std::max(1, atoi(e));
    int64_t grid = std::min<int64_t>(B, (int64_t)ctx->num_cus * occ * rounds);
    // The slabs of the levels that live in global memory (fp32, N = 1024: 16 KB each) behind the bitmap of claimed ones: a
    // power of two, at least twice the workgroups that can be resident at once (the smaller of grid and what the part
    // holds), at most SCL_SLAB_SLOTS; a part that could hold more than half of that gets a grid of that many workgroups.
    int rc;
    int64_t resident = std::min<int64_t>(grid, (int64_t)ctx->num_cus * occ);
    if (St::GLOBAL_VALUES > 0 && resident * 2 > SCL_SLAB_SLOTS) {
        grid = std::min<int64_t>(grid, SCL_SLAB_SLOTS / 2);
        resident = grid;
    }
    uint32_t slots = 32;
    while ((int64_t)slots < 2 * resident) slots *= 2;
    if (polar_fault_at("scl_slab")) slots = 32, grid = std::min<int64_t>(B, 4096);  // fewer slabs than resident workgroups
    const size_t slab_bytes = St::GLOBAL_VALUES > 0 ? SCL_SLAB_SLOTS / 8 + (size_t)slots * St::GLOBAL_VALUES * sizeof(T) : 4;
    if ((rc = ctx->scl_slab.reserve(slab_bytes))) return rc;
    if (St::GLOBAL_VALUES > 0) HIP_TRY(hipMemsetAsync(ctx->scl_slab.p, 0, SCL_SLAB_SLOTS / 8, ctx->stream));
    LaunchTimer tm(ctx);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), pad, ctx->stream, (const T *)llr, (const uint32_t *)c->d_maskw,
                       (const uint16_t *)c->d_Asl8, crc ? (const uint32_t *)c->d_crcS : nullptr, truth, info_hat, B, c->K, c->Kw,
                       (void *)ctx->scl_slab.p, pm, u_hat_all, slots / 32u, ctx->fault_dev);
    HIP_TRY(hipGetLastError());
    float ms = tm.stop();
    if (ctx->timing && (rc = polar_ctx_check_fault(ctx))) return rc;  // the timer has waited for the kernel
    if (sizeof(T) == 4) snprintf(c->last.kernel, sizeof(c->last.kernel), "scl_reg_kernel<%d>/L%d", NLOG, LMAX);
    else snprintf(c->last.kernel, sizeof(c->last.kernel), "scl_reg_kernel<%d,double>/L%d", NLOG, LMAX);
    c->last.lanes_per_frame = 64;
    c->last.lds_bytes_per_block = RS * 4 * (2 + (EAGER_L_TOP >= 4 ? 6 : 4) * (int)(sizeof(T) / 4)) + (NLOG < 10 ? 2 * c->N : 2) + (int)sizeof(T) * St::LDS_VALUES;
    c->last.block_threads = 64;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

template <int NLOG, typename T>
static int launch_scl_reg_l(polar_code *c, const void *llr, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                            const uint32_t *truth, int L)
{
    switch (L) {
    case 8: return launch_scl_reg_t<NLOG, 8, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
#ifndef POLAR_SCL_DEV_ONE  // development builds compile one instantiation (N = 1024, L = 8, fp32) in seconds
    case 4: return launch_scl_reg_t<NLOG, 4, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
    case 2: return launch_scl_reg_t<NLOG, 2, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth);
#endif
    default: polar_set_error("register SCL kernel: list size must be 2, 4 or 8"); return POLAR_EUNSUPPORTED;
    }
}

template <typename T>
static int launch_scl_reg_n(polar_code *c, const void *llr, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                            const uint32_t *truth, int L)
{
    switch (c->n) {
#ifndef POLAR_SCL_DEV_ONE
    case 6: return launch_scl_reg_l<6, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
    case 7: return launch_scl_reg_l<7, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
    case 8: return launch_scl_reg_l<8, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
    case 9: return launch_scl_reg_l<9, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
#endif
    case 10: return launch_scl_reg_l<10, T>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
    default: polar_set_error("register SCL kernel: unsupported length"); return POLAR_EUNSUPPORTED;
    }
}

int launch_scl_reg(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                   const uint32_t *truth, int L)
{
    if (B == 0) return POLAR_OK;
    if (dtype == POLAR_F32) return launch_scl_reg_n<float>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
#ifdef POLAR_SCL_DEV_ONE
    return POLAR_EUNSUPPORTED;
#else
    return launch_scl_reg_n<double>(c, llr, info_hat, pm, u_hat_all, B, crc, truth, L);
#endif
}
