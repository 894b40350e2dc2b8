// Internal declarations shared by the translation units of libpolar_hip.so.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/polar_hip.h"

// ---- error plumbing -------------------------------------------------------------------------
void polar_set_error(const char *fmt, ...);
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            polar_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return POLAR_EHIP;                                                                     \
        }                                                                                          \
    } while (0)
#define ARG_TRY(cond, msg)                                                                         \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            polar_set_error("invalid argument: %s", msg);                                          \
            return POLAR_EINVAL;                                                                   \
        }                                                                                          \
    } while (0)

#include "host_tables.h"  // decoder schedule encoding and the host-only table builders

// ---- grow-only device buffer ------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

struct polar_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int timing = 0;
    bool overlapped = false;  // a second stream of this context has kernels in flight (the staged Monte-Carlo loop)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Monte-Carlo pipeline: frames of chunk i+1 are generated on gen_stream while chunk i is decoded; host-space decoder
    // calls copy chunk i+1 of the caller's LLRs on it the same way
    hipStream_t gen_stream = nullptr;
    hipEvent_t gen_done[2] = {nullptr, nullptr}, dec_done[2] = {nullptr, nullptr};
    // staging / scratch
    DevBuf in_bytes, out_bytes, in_words, out_words, llr_buf, aux_buf, aux2_buf, pm_buf, counters, mc_info, mc_hat, scl_slab;
    DevBuf mc_llr2, mc_info2;  // second buffers of the Monte-Carlo pipeline (mc_llr2 also of the host-space LLR feed)
    DevBuf flip_rows, flip_ctr, flip_small;  // failing-frame list, {work counter, row count}, per-frame side outputs
    DevBuf flip_pairs;                       // two-flip search: frames, best pairs and ranked first flips of the frames it works on
    // one word of page-locked host memory that kernels can write (fault_dev is its device address): a kernel that gives up
    // (the slab search of the register list decoder) says so here, and the next synchronising call returns POLAR_EHIP
    uint32_t *fault_host = nullptr, *fault_dev = nullptr;
};
// after the context's stream was synchronised: POLAR_EHIP (and the word cleared) if a kernel reported a fault
int polar_ctx_check_fault(polar_ctx *ctx);

struct polar_code {
    polar_ctx *ctx = nullptr;
    int N = 0, n = 0, K = 0;
    int Nw = 0, Kw = 0;  // words per frame (codeword / info)
    std::vector<uint8_t> mask;
    std::vector<uint32_t> sched[4];  // 0 = plain, 1 = fast; 2 = plain, 3 = fast with 16-leaf nodes as single operations
    // device tables
    uint32_t *d_maskw = nullptr;   // Nw words
    uint32_t *d_rank = nullptr;    // Nw words: info bits before word w
    uint32_t *d_A = nullptr;       // K info positions, ascending
    uint32_t *d_sched[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t *d_Asl8 = nullptr;    // same for the 8-lanes-per-path layout: (lane-in-path << 7) | bit
    uint32_t *d_pid16 = nullptr;   // max(64, N/16) words: compiled-profile number of each 16-leaf node (register SC kernels)
    void *big_tables = nullptr;    // tables of the long-code SC kernel, built on first use (kernels_sc_big.hip)
    uint32_t *d_ctab = nullptr;    // compress / place table of the register SC kernels' output stage (build_compress_tables)
    uint32_t *d_Asl = nullptr;     // gather table of the register SC kernels: [bit i][word w] -> (staged word << 5) | bit
    // CRC (0 = none): generator with leading term, degree, and the two linear maps of the remainder
    uint64_t crc_poly = 0;
    int crc_w = 0;
    uint32_t *d_crcE = nullptr;    // K - crc_w words: CRC bits (bit j = j-th appended bit) contributed by message bit k
    uint32_t *d_crcS = nullptr;    // K words: remainder contributed by bit k of the K-bit word (message + CRC)
    // bit flipping: level-0 candidates, candidates after a flip at each position (CSR), sqrt of the mean LLRs
    int32_t *d_S0 = nullptr, *d_crit_off = nullptr, *d_crit_idx = nullptr, *d_crit_sum = nullptr;
    double *d_sqrt_mu = nullptr;
    int nS0 = 0, maxS = 0;
    bool have_mu = false, have_crit = false;
    // set around a launch by polar_sc_decode_count: the register SC kernels then compare with the transmitted
    // words (first count_nbits bits of each) and add to count_dev[3] instead of leaving that to a second kernel
    const uint32_t *count_truth = nullptr;
    int count_nbits = 0;
    uint64_t *count_dev = nullptr;
    // decoder compiled at run time for this rate profile (polar_code_specialize), or null
    hipModule_t static_mod = nullptr;
    hipFunction_t static_fn = nullptr;
    hipFunction_t static_mc_fn = nullptr;  // generate + decode + count in one kernel (absent below N = 128)
    polar_launch_info last{};
};

// ---- kernel launchers (each in its own .hip file) ----------------------------------------------
int launch_pack_bits(polar_ctx *ctx, const uint8_t *bytes, uint32_t *words, int64_t B, int nbits);
int launch_unpack_bits(polar_ctx *ctx, const uint32_t *words, uint8_t *bytes, int64_t B, int nbits);
int launch_encode(polar_code *c, const uint32_t *info_or_u, uint32_t *x, int64_t B, bool rate_profile);
int launch_channel(polar_code *c, const uint32_t *xw, void *llr, int dtype, double N0, uint64_t seed,
                   uint64_t frame0, int64_t B);
int launch_gen_frames(polar_code *c, uint32_t *info, void *llr, int dtype, double N0, uint64_t seed,
                      uint64_t frame0, int64_t B, bool crc);
int launch_sc_decode(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat,
                     void *dec_llr, int64_t B, int flags);
uint32_t sc_reg_profile_id(uint32_t m16);
bool sc_reg_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr);
int launch_sc_reg(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, int64_t B, int flags);
bool sc_reg_full_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr, int flags);
int launch_sc_reg_full(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat, void *dec_llr, int64_t B);
bool sc_big_supported(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr, int flags);
int launch_sc_big(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B);
void sc_big_release(polar_code *c);
bool scl_reg_supported(const polar_code *c, int dtype, int L, const void *pm, const void *u_hat_all);
// truth (packed [B x Kw], or NULL): genie-aided choice of the output path
int launch_scl_reg(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, bool crc,
                   const uint32_t *truth, int L);
// the two-lanes-per-path register kernel for L = 16 / 32 (fp32: best path only; fp64: also the sorted list and its metrics)
bool scl_wide_supported(const polar_code *c, int dtype, int L, const void *pm, const void *u_hat_all);
int launch_scl_wide(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, int L,
                    bool crc, const uint32_t *truth);
int launch_scl_decode(polar_code *c, const void *llr, int dtype, int L, uint32_t *info_hat, double *pm,
                      uint32_t *u_hat_all, int64_t B, bool crc, const uint32_t *truth);
// nbits: leading bits of each K-bit word that are compared
int launch_count_errors(polar_code *c, const uint32_t *info, const uint32_t *hat, int64_t B, uint64_t *counters, int nbits);

// Workspace of the two-flip search split into (frame, first flip) work items: ctr = {work counter, number of frames
// that failed every single flip} (zero at launch), rows2 / best one entry per frame that may get there, order_rec
// nS0 entries per such frame.
struct PbfSplit {
    unsigned long long *ctr;       // work counter of the pair search, number of frames handed over, work counter of the triple search
    uint32_t *rows2;
    unsigned long long *best;
    uint16_t *order_rec;
    unsigned long long *best3;     // three flips only
};
int launch_pbf(polar_code *c, const void *llr, int dtype, const uint32_t *rows, const unsigned long long *nrows_dev,
               unsigned long long *work, const int32_t *flips_in, int max_level, bool crc, uint32_t *info_hat, uint32_t *u_hat,
               void *dec_llr, int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B, const PbfSplit *split = nullptr);
int launch_crc_filter(polar_code *c, const uint32_t *info_hat, uint32_t *rows, unsigned long long *nrows, int32_t *runs,
                      uint8_t *crc_ok, int32_t *last_seq, int64_t B);

int launch_sc_static(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B);
int launch_sc_static_mc(polar_code *c, int64_t B, double N0, uint64_t seed, uint64_t frame0, uint64_t *counters_dev);

// timing bracket around the dominant kernel of a call
// Switches for experiments and for the tests that compare kernels (POLAR_SC_KERNEL, POLAR_SCL_KERNEL, ...): read from the
// environment only when POLAR_EXPERIMENT=1 is set as well, so that no stray variable changes which kernel a product run uses.
inline const char *polar_switch(const char *name)
{
    const char *on = getenv("POLAR_EXPERIMENT");
    if (!on || on[0] != '1') return nullptr;
    return getenv(name);
}

// ---- dispatch: every launcher asks the table of host_tables.cpp which kernel family serves the call ---------------------
inline int dq_switch_flags(const char *var)
{
    const char *e = polar_switch(var);
    if (!e) return 0;
    return e[0] == 'l' ? DQ_FORCE_LDS : (e[0] == 'g' || e[0] == 's') ? DQ_FORCE_GENERIC : 0;
}
// family (KernelFamily) that serves an SC call on this code, or -1
inline int dispatch_sc(const polar_code *c, int dtype, const void *u_hat, const void *dec_llr, int flags)
{
    DispatchQuery q{DQ_OP_SC, c->n, c->K, dtype == POLAR_F32 ? 0 : dtype == POLAR_F64 ? 1 : -1, 0,
                    DQ_OUT_INFO | (u_hat ? DQ_OUT_UHAT : 0) | (dec_llr ? DQ_OUT_DECLLR : 0),
                    ((flags & POLAR_SC_FAST) ? DQ_FAST : 0) | (c->static_fn ? DQ_SPECIALISED : 0) | dq_switch_flags("POLAR_SC_KERNEL")};
    const int row = dispatch_select(q);
    return row < 0 ? -1 : dispatch_table(nullptr)[row].family;
}
inline int dispatch_scl(const polar_code *c, int dtype, int L, const void *pm, const void *u_hat_all)
{
    DispatchQuery q{DQ_OP_SCL, c->n, c->K, dtype == POLAR_F32 ? 0 : dtype == POLAR_F64 ? 1 : -1, L,
                    DQ_OUT_INFO | (u_hat_all ? DQ_OUT_UHAT : 0) | (pm ? DQ_OUT_PM : 0), dq_switch_flags("POLAR_SCL_KERNEL")};
    const int row = dispatch_select(q);
    return row < 0 ? -1 : dispatch_table(nullptr)[row].family;
}

// Failure injection for the tests of the error paths (POLAR_FAULT_AT=<site>, honoured like every switch only with
// POLAR_EXPERIMENT=1): the named site behaves as if its runtime call had failed.
inline bool polar_fault_at(const char *site)
{
    const char *e = polar_switch("POLAR_FAULT_AT");
    return e && strcmp(e, site) == 0;
}

struct LaunchTimer {
    polar_ctx *ctx;
    explicit LaunchTimer(polar_ctx *c) : ctx(c)
    {
        if (ctx->timing) (void)hipEventRecord(ctx->ev0, ctx->stream);
    }
    float stop()
    {
        if (!ctx->timing) return 0.f;
        float ms = 0.f;
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        (void)hipEventSynchronize(ctx->ev1);
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        return ms;
    }
};
