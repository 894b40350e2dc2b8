// C-ABI entry points of libpolar_hip.so: argument checking, host<->device staging, chunking.
// The arithmetic lives in the kernels_*.hip files.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>

#include "polar_internal.h"

static thread_local char g_err[512] = "";

int polar_noalloc_setup(int device);
static int ensure_flip_tables(polar_code *c, bool need_lists);

void polar_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *polar_last_error(void) { return g_err; }
extern "C" const char *polar_version(void) { return "polar_hip 0.1 (gfx950)"; }

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return POLAR_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max(bytes, (size_t)1 << 16);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        polar_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return POLAR_ENOMEM;
    }
    cap = want;
    return POLAR_OK;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// ---------------------------------------------------------------------------------------------- ctx
extern "C" int polar_ctx_create(int device, polar_ctx **out)
{
    ARG_TRY(out != nullptr, "out is NULL");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        polar_set_error("device %d out of range (%d visible)", device, ndev);
        return POLAR_EINVAL;
    }
    HIP_TRY(hipSetDevice(device));
    {
        int rc = polar_noalloc_setup(device);
        if (rc) return rc;
    }
    polar_ctx *c = new (std::nothrow) polar_ctx();
    if (!c) return POLAR_ENOMEM;
    c->device = device;
    // every failure below hands what exists so far to polar_ctx_destroy (which skips what was never made)
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) {
        c->num_cus = prop.multiProcessorCount;
        e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    }
    c->stream = c->own_stream;
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = polar_fault_at("ctx_create") ? hipErrorOutOfMemory : hipStreamCreateWithFlags(&c->gen_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->fault_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        *c->fault_host = 0u;
        e = hipHostGetDevicePointer((void **)&c->fault_dev, c->fault_host, 0);
    }
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->gen_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->dec_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        polar_set_error("polar_ctx_create: %s", hipGetErrorString(e));
        (void)polar_ctx_destroy(c);
        return POLAR_EHIP;
    }
    *out = c;
    return POLAR_OK;
}

extern "C" int polar_ctx_destroy(polar_ctx *c)
{
    if (!c) return POLAR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *bufs[] = {&c->in_bytes, &c->out_bytes, &c->in_words, &c->out_words, &c->llr_buf,   &c->aux_buf,  &c->aux2_buf,
                      &c->pm_buf,   &c->counters,  &c->mc_info,  &c->mc_hat,    &c->flip_rows, &c->flip_ctr, &c->flip_small,
                      &c->mc_llr2,  &c->mc_info2,  &c->scl_slab,  &c->flip_pairs};
    for (DevBuf *b : bufs) b->release();
    // a context whose creation failed half-way holds only some of these
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->gen_stream) {
        (void)hipStreamSynchronize(c->gen_stream);
        (void)hipStreamDestroy(c->gen_stream);
    }
    for (int i = 0; i < 2; i++) {
        if (c->gen_done[i]) (void)hipEventDestroy(c->gen_done[i]);
        if (c->dec_done[i]) (void)hipEventDestroy(c->dec_done[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->fault_host) (void)hipHostFree(c->fault_host);
    delete c;
    return POLAR_OK;
}

extern "C" int polar_ctx_set_stream(polar_ctx *c, void *s)
{
    ARG_TRY(c != nullptr, "ctx is NULL");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return POLAR_OK;
}
int polar_ctx_check_fault(polar_ctx *ctx)
{
    if (!ctx->fault_host || *(volatile uint32_t *)ctx->fault_host == 0u) return POLAR_OK;
    *(volatile uint32_t *)ctx->fault_host = 0u;
    polar_set_error("a decoder kernel gave up: the register list decoder found no free slab for its wide levels (results of the call are incomplete)");
    return POLAR_EHIP;
}
extern "C" int polar_ctx_sync(polar_ctx *c)
{
    ARG_TRY(c != nullptr, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return polar_ctx_check_fault(c);
}
extern "C" int polar_ctx_num_cus(polar_ctx *c) { return c ? c->num_cus : 0; }
extern "C" int polar_ctx_enable_timing(polar_ctx *c, int on)
{
    ARG_TRY(c != nullptr, "ctx is NULL");
    c->timing = on;
    return POLAR_OK;
}

// ---------------------------------------------------------------------------------------------- schedule
// (the compiler itself is host-only code: host_tables.cpp)
extern "C" int polar_output_tables(const uint8_t *mask, int N, uint32_t *out256)
{
    if (!mask || !out256 || N < 32 || N > 1024 || (N & (N - 1))) {
        polar_set_error("polar_output_tables: mask, out256 and a power of two 32 <= N <= 1024 are needed");
        return POLAR_EINVAL;
    }
    const std::vector<uint32_t> t = build_compress_tables(mask, N);
    memcpy(out256, t.data(), 256 * sizeof(uint32_t));
    return POLAR_OK;
}

extern "C" int polar_schedule_build(const uint8_t *mask, int N, int fast, uint32_t *out, int cap)
{
    ARG_TRY(mask != nullptr, "mask is NULL");
    ARG_TRY(N >= 2 && N <= 65536 && (N & (N - 1)) == 0, "N must be a power of two in [2, 65536]");
    std::vector<uint8_t> m(mask, mask + N);
    for (auto &v : m) v = v ? 1 : 0;
    std::vector<uint32_t> ops = build_schedule(m.data(), N, (fast & 1) != 0, (fast & 2) != 0);
    if (out)
        for (size_t i = 0; i < ops.size() && (int)i < cap; i++) out[i] = ops[i];
    return (int)ops.size();
}

// ---------------------------------------------------------------------------------------------- code
extern "C" int polar_code_create(polar_ctx *ctx, int N, const uint8_t *mask, polar_code **out)
{
    ARG_TRY(ctx && mask && out, "NULL argument");
    ARG_TRY(N >= 2 && N <= 65536 && (N & (N - 1)) == 0, "N must be a power of two in [2, 65536]");
    HIP_TRY(hipSetDevice(ctx->device));
    polar_code *c = new (std::nothrow) polar_code();
    if (!c) return POLAR_ENOMEM;
    c->ctx = ctx;
    c->N = N;
    while ((1 << c->n) < N) c->n++;
    c->mask.assign(mask, mask + N);
    for (auto &m : c->mask) m = m ? 1 : 0;
    c->Nw = (N + 31) / 32;
    std::vector<uint32_t> maskw(c->Nw, 0), rank(c->Nw, 0), A;
    for (int i = 0; i < N; i++) {
        if ((i & 31) == 0) rank[i >> 5] = (uint32_t)A.size();
        if (c->mask[i]) {
            maskw[i >> 5] |= 1u << (i & 31);
            A.push_back((uint32_t)i);
        }
    }
    c->K = (int)A.size();
    c->Kw = std::max(1, (c->K + 31) / 32);
    c->sched[0] = build_schedule(c->mask.data(), N, false);
    c->sched[1] = build_schedule(c->mask.data(), N, true);
    c->sched[2] = build_schedule(c->mask.data(), N, false, true);
    c->sched[3] = build_schedule(c->mask.data(), N, true, true);
    // Gather table of the register SC kernels' output stage: entry [i * TW + w] locates information bit 32 w + i
    // in the frame's staged u_hat (element k in lane k mod 16, bit k / 16 of the lane's 64-bit slice, stored as
    // words 2 lane, 2 lane + 1): (word << 5) | bit.  TW = words per row; unused entries stay zero.
    std::vector<uint32_t> Asl;  // 16-bit entries, two per word
    if (N <= 1024) {
        const int TW = std::max(N / 32, 16);
        Asl.assign((size_t)32 * TW / 2, 0u);
        for (size_t k = 0; k < A.size(); k++) {
            const uint32_t pos = A[k], lane = pos & 15u, j = pos >> 4;
            const uint32_t e = ((2u * lane + (j >> 5)) << 5) | (j & 31u);
            const size_t at = (k & 31) * TW + (k >> 5);
            Asl[at >> 1] |= e << (16 * (at & 1));
        }
    }
    std::vector<uint32_t> Asl8((A.size() + 1) / 2, 0u);  // 16-bit entries, two per word
    for (size_t k = 0; k < A.size(); k++) {
        const uint32_t pos = A[k], e = pos & 7u, q = e < 4 ? e : 11 - e;
        if (N <= 1024) Asl8[k >> 1] |= ((q << 7) | (pos >> 3)) << (16 * (k & 1));
    }
    std::vector<uint32_t> pid16(std::max(64, N / 16), 0);
    for (int b = 0; b < N / 16; b++) {
        uint32_t m16 = 0;
        for (int i = 0; i < 16; i++) m16 |= (uint32_t)c->mask[16 * b + i] << i;
        pid16[b] = sc_reg_profile_id(m16);
    }
    const std::vector<uint32_t> ctab = build_compress_tables(c->mask.data(), N);
    if (A.empty()) A.push_back(0);
    if (Asl.empty()) Asl.push_back(0);
    if (Asl8.empty()) Asl8.push_back(0);

    auto up = [&](uint32_t **dst, const std::vector<uint32_t> &v) -> int {
        HIP_TRY(hipMalloc((void **)dst, v.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        return POLAR_OK;
    };
    int rc;
    if ((rc = up(&c->d_maskw, maskw)) || (rc = up(&c->d_rank, rank)) || (rc = up(&c->d_A, A)) ||
        (rc = up(&c->d_sched[0], c->sched[0])) || (rc = up(&c->d_sched[1], c->sched[1])) ||
        (rc = up(&c->d_sched[2], c->sched[2])) || (rc = up(&c->d_sched[3], c->sched[3])) || (rc = up(&c->d_Asl, Asl)) || (rc = up(&c->d_Asl8, Asl8)) || (rc = up(&c->d_ctab, ctab)) ||
        (rc = up(&c->d_pid16, pid16))) {
        polar_code_destroy(c);
        return rc;
    }
    *out = c;
    return POLAR_OK;
}

extern "C" int polar_code_destroy(polar_code *c)
{
    if (!c) return POLAR_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    uint32_t *ps[] = {c->d_maskw, c->d_rank, c->d_A, c->d_sched[0], c->d_sched[1], c->d_sched[2], c->d_sched[3], c->d_Asl, c->d_Asl8, c->d_ctab,
                      c->d_crcE, c->d_crcS, c->d_pid16};
    for (uint32_t *p : ps)
        if (p) (void)hipFree(p);
    void *qs[] = {c->d_S0, c->d_crit_off, c->d_crit_idx, c->d_crit_sum, c->d_sqrt_mu};
    for (void *p : qs)
        if (p) (void)hipFree(p);
    if (c->static_mod) (void)hipModuleUnload(c->static_mod);
    sc_big_release(c);
    delete c;
    return POLAR_OK;
}
extern "C" int polar_code_n(const polar_code *c) { return c ? c->N : 0; }
extern "C" int polar_code_k(const polar_code *c) { return c ? c->K : 0; }
extern "C" int polar_code_schedule_len(const polar_code *c, int which)
{
    return (c && which >= 0 && which < 4) ? (int)c->sched[which].size() : 0;
}
extern "C" int polar_code_crc_width(const polar_code *c) { return c ? c->crc_w : 0; }

extern "C" int polar_code_set_crc(polar_code *c, uint64_t crc_poly)
{
    ARG_TRY(c != nullptr, "code is NULL");
    HIP_TRY(hipSetDevice(c->ctx->device));
    // the new tables are validated, built and uploaded first; the code keeps its old CRC if anything fails
    uint32_t *nE = nullptr, *nS = nullptr;
    int w = 0;
    if (crc_poly != 0) {
        w = 63 - __builtin_clzll(crc_poly);
        ARG_TRY(w >= 1 && w <= 32, "CRC width (degree of crc_poly) must be in [1, 32]");
        ARG_TRY(w <= c->K, "CRC is wider than the code's information length");
        std::vector<uint32_t> E, S;
        build_crc_tables(crc_poly, w, c->K, E, S);
        hipError_t e = hipMalloc((void **)&nE, E.size() * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&nS, S.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(nE, E.data(), E.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(nS, S.data(), S.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (nE) (void)hipFree(nE);
            if (nS) (void)hipFree(nS);
            polar_set_error("uploading the CRC tables failed: %s", hipGetErrorString(e));
            return POLAR_EHIP;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));  // launches that read the old tables have finished
    if (c->d_crcE) (void)hipFree(c->d_crcE);
    if (c->d_crcS) (void)hipFree(c->d_crcS);
    c->d_crcE = nE;
    c->d_crcS = nS;
    c->crc_poly = crc_poly;
    c->crc_w = w;
    return POLAR_OK;
}
// the list and flip decoders keep a frame's whole tree in one wavefront's LDS and 12 pointer levels per path
#define N_LIMIT_TRY(c, what)                                                                          \
    do {                                                                                              \
        if ((c)->n > 12) {                                                                            \
            polar_set_error("%s: N=%d exceeds the supported maximum of 4096", what, (c)->N);         \
            return POLAR_EUNSUPPORTED;                                                                \
        }                                                                                             \
    } while (0)
#define CRC_TRY(c, flags) ARG_TRY(!((flags)&POLAR_CRC) || (c)->crc_w > 0, "POLAR_CRC needs polar_code_set_crc first")

extern "C" int polar_last_launch(polar_code *c, polar_launch_info *out)
{
    ARG_TRY(c && out, "NULL argument");
    *out = c->last;
    return POLAR_OK;
}

// ---------------------------------------------------------------------------------------------- helpers
// Host-space calls are chunked so the staging buffers stay modest.
static const int64_t HOST_CHUNK = 1 << 14;

static size_t llr_size(int dtype) { return dtype == POLAR_F64 ? sizeof(double) : sizeof(float); }

// LLRs handed over in host memory: the batch crosses the link in chunks through two staging buffers, chunk i+1 on the
// context's second stream while chunk i is decoded on the first (with page-locked host memory the copies are asynchronous
// and the link stays busy; from pageable memory the copy call itself blocks, but the decode launched before it runs
// meanwhile).  Outputs go back on the decode stream behind their kernel; the caller synchronises once, after the loop.
struct HostLlrFeed {
    polar_ctx *ctx = nullptr;
    const char *host = nullptr;
    size_t frame_bytes = 0;
    int64_t B = 0, chunk = 0;
    void *buf[2] = {nullptr, nullptr};
    int64_t nchunks() const { return (B + chunk - 1) / chunk; }
    // the caller's buffers belong to the caller again when the call returns, also when it returns an error half-way:
    // nothing that reads or writes them is left in flight (after a call that succeeded both streams are idle already)
    ~HostLlrFeed()
    {
        if (!ctx) return;
        (void)hipStreamSynchronize(ctx->gen_stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    int begin(polar_ctx *x, const void *llr, size_t fb, int64_t nframes, int64_t per)
    {
        ctx = x, host = (const char *)llr, frame_bytes = fb, B = nframes, chunk = std::min(per, nframes);
        int rc;
        if ((rc = ctx->llr_buf.reserve(chunk * frame_bytes))) return rc;
        if (nchunks() > 1 && (rc = ctx->mc_llr2.reserve(chunk * frame_bytes))) return rc;
        buf[0] = ctx->llr_buf.p, buf[1] = nchunks() > 1 ? ctx->mc_llr2.p : ctx->llr_buf.p;
        // the copies must not overtake work queued earlier on the context's stream (it may still read the buffers)
        HIP_TRY(hipEventRecord(ctx->dec_done[1], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->gen_stream, ctx->dec_done[1], 0));
        return copy(0);
    }
    int copy(int64_t idx)
    {
        const int b = (int)(idx & 1);
        const int64_t b0 = idx * chunk, nb = std::min(chunk, B - b0);
        // buffer b was last read by the decode of chunk idx - 2
        if (idx >= 2) HIP_TRY(hipStreamWaitEvent(ctx->gen_stream, ctx->dec_done[b], 0));
        HIP_TRY(hipMemcpyAsync(buf[b], host + b0 * frame_bytes, nb * frame_bytes, hipMemcpyHostToDevice, ctx->gen_stream));
        HIP_TRY(hipEventRecord(ctx->gen_done[b], ctx->gen_stream));
        return POLAR_OK;
    }
    // device address of chunk idx; the decode stream waits for its copy
    int take(int64_t idx, const char **p)
    {
        const int b = (int)(idx & 1);
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->gen_done[b], 0));
        *p = (const char *)buf[b];
        return POLAR_OK;
    }
    // after the decode of chunk idx has been queued: mark its buffer's last reader and start the next copy
    int decoded(int64_t idx)
    {
        HIP_TRY(hipEventRecord(ctx->dec_done[idx & 1], ctx->stream));
        return idx + 1 < nchunks() ? copy(idx + 1) : POLAR_OK;
    }
};

#define CHECK_COMMON(code, B, space)                                                      \
    ARG_TRY((code) != nullptr, "code is NULL");                                           \
    ARG_TRY((B) >= 0, "B must be >= 0");                                                  \
    ARG_TRY((space) == POLAR_HOST || (space) == POLAR_DEVICE, "space must be POLAR_HOST or POLAR_DEVICE"); \
    HIP_TRY(hipSetDevice((code)->ctx->device));                                           \
    if ((B) == 0) return POLAR_OK

// ---------------------------------------------------------------------------------------------- encode
static int encode_bytes(polar_code *c, const uint8_t *in, uint8_t *x, int64_t B, int space, bool rate_profile)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(in && x, "NULL buffer");
    polar_ctx *ctx = c->ctx;
    const int inbits = rate_profile ? c->K : c->N;
    const int inw = rate_profile ? c->Kw : c->Nw;
    int rc;
    for (int64_t b0 = 0; b0 < B; b0 += HOST_CHUNK) {
        int64_t nb = std::min(HOST_CHUNK, B - b0);
        const uint8_t *din = in + b0 * inbits;
        uint8_t *dx = x + b0 * c->N;
        if ((rc = ctx->in_words.reserve(nb * inw * 4)) || (rc = ctx->out_words.reserve(nb * c->Nw * 4))) return rc;
        if (space == POLAR_HOST) {
            if ((rc = ctx->in_bytes.reserve(std::max<size_t>(1, nb * inbits))) || (rc = ctx->out_bytes.reserve(nb * c->N))) return rc;
            if (inbits) HIP_TRY(hipMemcpyAsync(ctx->in_bytes.p, din, nb * inbits, hipMemcpyHostToDevice, ctx->stream));
            din = ctx->in_bytes.as<uint8_t>();
            dx = ctx->out_bytes.as<uint8_t>();
        }
        if ((rc = launch_pack_bits(ctx, din, ctx->in_words.as<uint32_t>(), nb, inbits))) return rc;
        if ((rc = launch_encode(c, ctx->in_words.as<uint32_t>(), ctx->out_words.as<uint32_t>(), nb, rate_profile))) return rc;
        if ((rc = launch_unpack_bits(ctx, ctx->out_words.as<uint32_t>(), dx, nb, c->N))) return rc;
        if (space == POLAR_HOST) {
            HIP_TRY(hipMemcpyAsync(x + b0 * c->N, dx, nb * c->N, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    return POLAR_OK;
}

extern "C" int polar_encode(polar_code *c, const uint8_t *info, uint8_t *x, int64_t B, int space)
{
    return encode_bytes(c, info, x, B, space, true);
}
extern "C" int polar_mul_matrix(polar_code *c, const uint8_t *u, uint8_t *x, int64_t B, int space)
{
    return encode_bytes(c, u, x, B, space, false);
}

extern "C" int polar_encode_packed(polar_code *c, const uint32_t *info, uint32_t *x, int64_t B, int space)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(info && x, "NULL buffer");
    polar_ctx *ctx = c->ctx;
    if (space == POLAR_DEVICE) return launch_encode(c, info, x, B, true);
    int rc;
    for (int64_t b0 = 0; b0 < B; b0 += HOST_CHUNK) {
        int64_t nb = std::min(HOST_CHUNK, B - b0);
        if ((rc = ctx->in_words.reserve(nb * c->Kw * 4)) || (rc = ctx->out_words.reserve(nb * c->Nw * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->in_words.p, info + b0 * c->Kw, nb * c->Kw * 4, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_encode(c, ctx->in_words.as<uint32_t>(), ctx->out_words.as<uint32_t>(), nb, true))) return rc;
        HIP_TRY(hipMemcpyAsync(x + b0 * c->Nw, ctx->out_words.p, nb * c->Nw * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return POLAR_OK;
}

// ---------------------------------------------------------------------------------------------- channel
extern "C" int polar_channel_llr(polar_code *c, const uint8_t *x, void *llr, int dtype, double N0, uint64_t seed,
                                 uint64_t frame0, int64_t B, int space)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(x && llr, "NULL buffer");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(N0 > 0, "N0 must be positive");
    polar_ctx *ctx = c->ctx;
    const size_t es = llr_size(dtype);
    int rc;
    for (int64_t b0 = 0; b0 < B; b0 += HOST_CHUNK) {
        int64_t nb = std::min(HOST_CHUNK, B - b0);
        const uint8_t *dx = x + b0 * c->N;
        char *dl = (char *)llr + b0 * c->N * es;
        if ((rc = ctx->in_words.reserve(nb * c->Nw * 4))) return rc;
        if (space == POLAR_HOST) {
            if ((rc = ctx->in_bytes.reserve(nb * c->N)) || (rc = ctx->llr_buf.reserve(nb * c->N * es))) return rc;
            HIP_TRY(hipMemcpyAsync(ctx->in_bytes.p, dx, nb * c->N, hipMemcpyHostToDevice, ctx->stream));
            dx = ctx->in_bytes.as<uint8_t>();
            dl = ctx->llr_buf.as<char>();
        }
        if ((rc = launch_pack_bits(ctx, dx, ctx->in_words.as<uint32_t>(), nb, c->N))) return rc;
        if ((rc = launch_channel(c, ctx->in_words.as<uint32_t>(), dl, dtype, N0, seed, frame0 + b0, nb))) return rc;
        if (space == POLAR_HOST) {
            HIP_TRY(hipMemcpyAsync((char *)llr + b0 * c->N * es, dl, nb * c->N * es, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    return POLAR_OK;
}

extern "C" int polar_gen_frames(polar_code *c, uint32_t *info, void *llr, int dtype, double N0, uint64_t seed,
                                uint64_t frame0, int64_t B, int flags)
{
    CHECK_COMMON(c, B, POLAR_DEVICE);
    CRC_TRY(c, flags);
    ARG_TRY(llr != nullptr, "llr is NULL");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(N0 > 0, "N0 must be positive");
    return launch_gen_frames(c, info, llr, dtype, N0, seed, frame0, B, (flags & POLAR_CRC) != 0);
}

// ---------------------------------------------------------------------------------------------- SC
static int sc_common(polar_code *c, const void *llr, int dtype, void *info_hat, void *u_hat, void *dec_llr,
                     int64_t B, int flags, int space, bool packed)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(llr != nullptr, "llr is NULL");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(!((flags & POLAR_SC_FAST) && dec_llr), "decision LLRs are not produced with POLAR_SC_FAST");
    polar_ctx *ctx = c->ctx;
    const size_t es = llr_size(dtype);
    if (space == POLAR_DEVICE && packed)
        return launch_sc_decode(c, llr, dtype, (uint32_t *)info_hat, (uint32_t *)u_hat, dec_llr, B, flags);
    int rc;
    const int64_t chunk = (space == POLAR_HOST) ? HOST_CHUNK : (int64_t)1 << 20;
    HostLlrFeed feed;
    if (space == POLAR_HOST && (rc = feed.begin(ctx, llr, c->N * es, B, chunk))) return rc;
    int64_t idx = 0;
    for (int64_t b0 = 0; b0 < B; b0 += chunk, idx++) {
        int64_t nb = std::min(chunk, B - b0);
        const char *dl = (const char *)llr + b0 * c->N * es;
        char *ddec = dec_llr ? (char *)dec_llr + b0 * c->N * es : nullptr;
        if ((rc = ctx->out_words.reserve(nb * c->Kw * 4)) || (rc = ctx->aux_buf.reserve(nb * c->Nw * 4))) return rc;
        uint32_t *wi = ctx->out_words.as<uint32_t>(), *wu = u_hat ? ctx->aux_buf.as<uint32_t>() : nullptr;
        if (space == POLAR_HOST) {
            if (dec_llr) {
                if ((rc = ctx->aux2_buf.reserve(nb * c->N * es))) return rc;
                ddec = ctx->aux2_buf.as<char>();
            }
            if ((rc = feed.take(idx, &dl))) return rc;
        }
        if ((rc = launch_sc_decode(c, dl, dtype, wi, wu, ddec, nb, flags))) return rc;
        if (space == POLAR_HOST && (rc = feed.decoded(idx))) return rc;
        if (packed) {  // host + packed
            if (info_hat) HIP_TRY(hipMemcpyAsync((uint32_t *)info_hat + b0 * c->Kw, wi, nb * c->Kw * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (u_hat) HIP_TRY(hipMemcpyAsync((uint32_t *)u_hat + b0 * c->Nw, wu, nb * c->Nw * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            uint8_t *bi = (uint8_t *)info_hat, *bu = (uint8_t *)u_hat;
            if (space == POLAR_HOST) {
                if ((rc = ctx->out_bytes.reserve(nb * ((size_t)c->K + c->N) + 1))) return rc;
                uint8_t *tb = ctx->out_bytes.as<uint8_t>();
                if (info_hat) {
                    if ((rc = launch_unpack_bits(ctx, wi, tb, nb, c->K))) return rc;
                    if (c->K) HIP_TRY(hipMemcpyAsync(bi + b0 * c->K, tb, nb * c->K, hipMemcpyDeviceToHost, ctx->stream));
                }
                if (u_hat) {
                    uint8_t *tu = tb + nb * c->K;
                    if ((rc = launch_unpack_bits(ctx, wu, tu, nb, c->N))) return rc;
                    HIP_TRY(hipMemcpyAsync(bu + b0 * c->N, tu, nb * c->N, hipMemcpyDeviceToHost, ctx->stream));
                }
            } else {
                if (info_hat && (rc = launch_unpack_bits(ctx, wi, bi + b0 * c->K, nb, c->K))) return rc;
                if (u_hat && (rc = launch_unpack_bits(ctx, wu, bu + b0 * c->N, nb, c->N))) return rc;
            }
        }
        if (space == POLAR_HOST && dec_llr)
            HIP_TRY(hipMemcpyAsync((char *)dec_llr + b0 * c->N * es, ddec, nb * c->N * es, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (space == POLAR_HOST) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (int frc = polar_ctx_check_fault(ctx)) return frc;
    }
    return POLAR_OK;
}

extern "C" int polar_sc_decode(polar_code *c, const void *llr, int dtype, uint8_t *info_hat, uint8_t *u_hat,
                               void *dec_llr, int64_t B, int flags, int space)
{
    return sc_common(c, llr, dtype, info_hat, u_hat, dec_llr, B, flags, space, false);
}
extern "C" int polar_sc_decode_packed(polar_code *c, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat,
                                      void *dec_llr, int64_t B, int flags, int space)
{
    return sc_common(c, llr, dtype, info_hat, u_hat, dec_llr, B, flags, space, true);
}

// ---------------------------------------------------------------------------------------------- SCL
static int scl_common(polar_code *c, const void *llr, int dtype, int L, void *info_hat, double *pm,
                      void *u_hat_all, int64_t B, int flags, int space, bool packed, const void *truth = nullptr)
{
    CHECK_COMMON(c, B, space);
    CRC_TRY(c, flags);
    const bool crc = (flags & POLAR_CRC) != 0;
    ARG_TRY(llr != nullptr, "llr is NULL");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 32, "L must be a power of two in [1, 32]");
    N_LIMIT_TRY(c, "list decoder");
    polar_ctx *ctx = c->ctx;
    const size_t es = llr_size(dtype);
    if (space == POLAR_DEVICE && packed)
        return launch_scl_decode(c, llr, dtype, L, (uint32_t *)info_hat, pm, (uint32_t *)u_hat_all, B, crc, (const uint32_t *)truth);
    int rc = POLAR_OK;
    // the whole list as an output takes L code words per frame of staging: fewer frames per chunk then
    const int64_t chunk = (space == POLAR_HOST) ? (u_hat_all ? std::max<int64_t>(1, HOST_CHUNK / L) : HOST_CHUNK) : (int64_t)1 << 18;
    HostLlrFeed feed;
    if (space == POLAR_HOST && (rc = feed.begin(ctx, llr, c->N * es, B, chunk))) return rc;
    int64_t idx = 0;
    for (int64_t b0 = 0; b0 < B; b0 += chunk, idx++) {
        int64_t nb = std::min(chunk, B - b0);
        const char *dl = (const char *)llr + b0 * c->N * es;
        if ((rc = ctx->out_words.reserve(nb * c->Kw * 4)) || (rc = ctx->aux_buf.reserve(nb * L * c->Nw * 4)) ||
            (rc = ctx->pm_buf.reserve(nb * L * sizeof(double))))
            return rc;
        uint32_t *wi = ctx->out_words.as<uint32_t>(), *wu = u_hat_all ? ctx->aux_buf.as<uint32_t>() : nullptr;
        double *dpm = pm ? (space == POLAR_HOST ? ctx->pm_buf.as<double>() : pm + b0 * L) : nullptr;
        if (space == POLAR_HOST && (rc = feed.take(idx, &dl))) return rc;
        const uint32_t *wt = nullptr;
        if (truth) {
            if (packed && space == POLAR_DEVICE) {
                wt = (const uint32_t *)truth + b0 * c->Kw;
            } else {
                if ((rc = ctx->in_words.reserve(nb * c->Kw * 4))) return rc;
                wt = ctx->in_words.as<uint32_t>();
                if (packed) {  // host words
                    HIP_TRY(hipMemcpyAsync(ctx->in_words.p, (const uint32_t *)truth + b0 * c->Kw, nb * c->Kw * 4, hipMemcpyHostToDevice, ctx->stream));
                } else {
                    const uint8_t *tb = (const uint8_t *)truth + b0 * c->K;
                    if (space == POLAR_HOST) {
                        if ((rc = ctx->in_bytes.reserve(std::max<size_t>(1, nb * c->K)))) return rc;
                        if (c->K) HIP_TRY(hipMemcpyAsync(ctx->in_bytes.p, tb, nb * c->K, hipMemcpyHostToDevice, ctx->stream));
                        tb = ctx->in_bytes.as<uint8_t>();
                    }
                    if ((rc = launch_pack_bits(ctx, tb, ctx->in_words.as<uint32_t>(), nb, c->K))) return rc;
                }
            }
        }
        if ((rc = launch_scl_decode(c, dl, dtype, L, wi, dpm, wu, nb, crc, wt))) return rc;
        if (space == POLAR_HOST && (rc = feed.decoded(idx))) return rc;
        if (packed) {
            if (info_hat) HIP_TRY(hipMemcpyAsync((uint32_t *)info_hat + b0 * c->Kw, wi, nb * c->Kw * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (u_hat_all) HIP_TRY(hipMemcpyAsync((uint32_t *)u_hat_all + b0 * L * c->Nw, wu, nb * L * c->Nw * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            uint8_t *bi = (uint8_t *)info_hat, *bu = (uint8_t *)u_hat_all;
            if (space == POLAR_HOST) {
                if ((rc = ctx->out_bytes.reserve(nb * ((size_t)c->K + (size_t)L * c->N) + 1))) return rc;
                uint8_t *tb = ctx->out_bytes.as<uint8_t>();
                if (info_hat) {
                    if ((rc = launch_unpack_bits(ctx, wi, tb, nb, c->K))) return rc;
                    if (c->K) HIP_TRY(hipMemcpyAsync(bi + b0 * c->K, tb, nb * c->K, hipMemcpyDeviceToHost, ctx->stream));
                }
                if (u_hat_all) {
                    uint8_t *tu = tb + nb * c->K;
                    if ((rc = launch_unpack_bits(ctx, wu, tu, nb * L, c->N))) return rc;
                    HIP_TRY(hipMemcpyAsync(bu + b0 * L * c->N, tu, nb * L * c->N, hipMemcpyDeviceToHost, ctx->stream));
                }
            } else {
                if (info_hat && (rc = launch_unpack_bits(ctx, wi, bi + b0 * c->K, nb, c->K))) return rc;
                if (u_hat_all && (rc = launch_unpack_bits(ctx, wu, bu + b0 * L * c->N, nb * L, c->N))) return rc;
            }
        }
        if (space == POLAR_HOST && pm) HIP_TRY(hipMemcpyAsync(pm + b0 * L, dpm, nb * L * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (space == POLAR_HOST) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (int frc = polar_ctx_check_fault(ctx)) return frc;
    }
    return POLAR_OK;
}

extern "C" int polar_scl_decode(polar_code *c, const void *llr, int dtype, int L, uint8_t *info_hat, double *pm,
                                uint8_t *u_hat_all, int64_t B, int flags, int space)
{
    return scl_common(c, llr, dtype, L, info_hat, pm, u_hat_all, B, flags, space, false);
}
extern "C" int polar_scl_decode_packed(polar_code *c, const void *llr, int dtype, int L, uint32_t *info_hat,
                                       double *pm, uint32_t *u_hat_all, int64_t B, int flags, int space)
{
    return scl_common(c, llr, dtype, L, info_hat, pm, u_hat_all, B, flags, space, true);
}

extern "C" int polar_scl_decode_genie(polar_code *c, const void *llr, int dtype, int L, const uint8_t *info_true,
                                      uint8_t *info_hat, double *pm, uint8_t *u_hat_all, int64_t B, int space)
{
    ARG_TRY(info_true != nullptr, "info_true is NULL");
    return scl_common(c, llr, dtype, L, info_hat, pm, u_hat_all, B, 0, space, false, info_true);
}
extern "C" int polar_scl_decode_genie_packed(polar_code *c, const void *llr, int dtype, int L, const uint32_t *info_true,
                                             uint32_t *info_hat, double *pm, uint32_t *u_hat_all, int64_t B, int space)
{
    ARG_TRY(info_true != nullptr, "info_true is NULL");
    return scl_common(c, llr, dtype, L, info_hat, pm, u_hat_all, B, 0, space, true, info_true);
}

// ---------------------------------------------------------------------------------------------- bit flipping
extern "C" int polar_critical_set(const uint8_t *mask, int N, int after, int32_t *out, int cap)
{
    ARG_TRY(mask != nullptr, "mask is NULL");
    ARG_TRY(N >= 2 && N <= 65536 && (N & (N - 1)) == 0, "N must be a power of two in [2, 65536]");
    std::vector<int32_t> cs = critical_set(mask, N, after < 0 ? -1 : after);
    if (out)
        for (size_t i = 0; i < cs.size() && (int)i < cap; i++) out[i] = cs[i];
    return (int)cs.size();
}

extern "C" int polar_code_set_reliability(polar_code *c, const double *mu)
{
    ARG_TRY(c && mu, "NULL argument");
    HIP_TRY(hipSetDevice(c->ctx->device));
    HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    std::vector<double> s(c->N);
    for (int i = 0; i < c->N; i++) {
        ARG_TRY(mu[i] > 0, "mu must be positive");
        s[i] = std::sqrt(mu[i]);
    }
    if (!c->d_sqrt_mu) HIP_TRY(hipMalloc((void **)&c->d_sqrt_mu, sizeof(double) * c->N));
    HIP_TRY(hipMemcpy(c->d_sqrt_mu, s.data(), sizeof(double) * c->N, hipMemcpyHostToDevice));
    c->have_mu = true;
    // the candidate tables of the flip search depend on the rate profile only; built here so that no decode call has to
    return ensure_flip_tables(c, true);
}

// Candidate tables of the flip search; the per-position lists are only needed from two simultaneous flips on.
static int ensure_flip_tables(polar_code *c, bool need_lists)
{
    if (!c->d_S0) {
        std::vector<int32_t> s0 = critical_set(c->mask.data(), c->N, -1);
        c->nS0 = (int)s0.size();
        c->maxS = std::max(c->maxS, c->nS0);
        if (s0.empty()) s0.push_back(0);
        HIP_TRY(hipMalloc((void **)&c->d_S0, s0.size() * 4));
        HIP_TRY(hipMemcpy(c->d_S0, s0.data(), s0.size() * 4, hipMemcpyHostToDevice));
    }
    if (need_lists && !c->have_crit) {
        std::vector<int32_t> off(c->N + 1, 0), idx;
        for (int i = 0; i < c->N; i++) {
            if (c->mask[i]) {
                std::vector<int32_t> cs = critical_set(c->mask.data(), c->N, i);
                idx.insert(idx.end(), cs.begin(), cs.end());
                c->maxS = std::max(c->maxS, (int)cs.size());
            }
            off[i + 1] = (int32_t)idx.size();
        }
        // candidates two levels below a flip at position i (what the three-flip search tests under it)
        std::vector<int32_t> below(c->N, 0);
        for (int i = 0; i < c->N; i++)
            for (int32_t k = off[i]; k < off[i + 1]; k++) below[i] += off[idx[k] + 1] - off[idx[k]];
        if (idx.empty()) idx.push_back(0);
        HIP_TRY(hipMalloc((void **)&c->d_crit_sum, below.size() * 4));
        HIP_TRY(hipMemcpy(c->d_crit_sum, below.data(), below.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&c->d_crit_off, off.size() * 4));
        HIP_TRY(hipMalloc((void **)&c->d_crit_idx, idx.size() * 4));
        HIP_TRY(hipMemcpy(c->d_crit_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_crit_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
        c->have_crit = true;
    }
    return POLAR_OK;
}

// Device-side body: fast SC on every frame, CRC test, search kernel on the frames that failed.
// wi: packed information words out; d_runs / d_ok / d_last: device side outputs or NULL.
// rows (one word per frame) and ctr (two 64-bit words: work counter, row count) are the scratch of the call.
// split: workspace of the two-flip search in (frame, first flip) pieces, or NULL for one wavefront per frame.
static int pbf_device_on(polar_code *c, const void *dl, int dtype, int max_level, uint32_t *wi, int32_t *d_runs, uint8_t *d_ok,
                         int32_t *d_last, int64_t nb, uint32_t *rows, unsigned long long *ctr, const PbfSplit *split = nullptr)
{
    polar_ctx *ctx = c->ctx;
    int rc;
    HIP_TRY(hipMemsetAsync(ctr, 0, 16, ctx->stream));
    if (split) HIP_TRY(hipMemsetAsync(split->ctr, 0, 24, ctx->stream));
    if ((rc = launch_sc_decode(c, dl, dtype, wi, nullptr, nullptr, nb, POLAR_SC_FAST))) return rc;
    if ((rc = launch_crc_filter(c, wi, rows, ctr + 1, d_runs, d_ok, d_last, nb))) return rc;
    if (max_level >= 1)
        rc = launch_pbf(c, dl, dtype, rows, ctr + 1, ctr, nullptr, max_level, true, wi, nullptr, nullptr, d_runs, d_ok, d_last, nb,
                        split);
    return rc;
}

// Frames per pass of the two- and three-flip searches: bounds their workspace (a frame that fails every single flip keeps
// 20 + 2 nS0 bytes) whatever the batch size.
static const int64_t PBF_SPLIT_FRAMES = (int64_t)1 << 18;

static int pbf_device(polar_code *c, const void *dl, int dtype, int max_level, uint32_t *wi, int32_t *d_runs, uint8_t *d_ok,
                      int32_t *d_last, int64_t nb)
{
    polar_ctx *ctx = c->ctx;
    int rc;
    if ((rc = ensure_flip_tables(c, max_level >= 2))) return rc;
    if ((rc = ctx->flip_rows.reserve(nb * 4)) || (rc = ctx->flip_ctr.reserve(64))) return rc;
    uint32_t *rows = ctx->flip_rows.as<uint32_t>();
    unsigned long long *ctr = ctx->flip_ctr.as<unsigned long long>();
    const char *sw = polar_switch("POLAR_PBF_SPLIT");
    const bool whole_frames = sw && atoi(sw) == 0;
    // two flips: the second level is worked on in (frame, first flip) pieces, PBF_SPLIT_FRAMES frames at a time; three flips
    // the same way (a frame that needs the third level costs 10^5 runs: one wavefront per frame would hold the whole launch
    // for seconds)
    const bool split3 = max_level == 3;
    if ((max_level != 2 && !split3) || whole_frames) return pbf_device_on(c, dl, dtype, max_level, wi, d_runs, d_ok, d_last, nb, rows, ctr);
    const int64_t cap = std::min(nb, PBF_SPLIT_FRAMES);
    const size_t rec = (size_t)std::max(c->nS0, 1) * 2;
    const size_t off_best = ((size_t)cap * 4 + 255) & ~(size_t)255, off_best3 = off_best + (size_t)cap * 8, off_rec = off_best3 + (size_t)cap * 8;
    if ((rc = ctx->flip_pairs.reserve(off_rec + (size_t)cap * rec))) return rc;
    PbfSplit sp;
    sp.ctr = ctr + 2;
    sp.rows2 = ctx->flip_pairs.as<uint32_t>();
    sp.best = reinterpret_cast<unsigned long long *>(ctx->flip_pairs.as<char>() + off_best);
    sp.best3 = split3 ? reinterpret_cast<unsigned long long *>(ctx->flip_pairs.as<char>() + off_best3) : nullptr;
    sp.order_rec = reinterpret_cast<uint16_t *>(ctx->flip_pairs.as<char>() + off_rec);
    const size_t es = llr_size(dtype);
    for (int64_t b0 = 0; b0 < nb; b0 += cap) {
        const int64_t m = std::min(cap, nb - b0);
        if ((rc = pbf_device_on(c, (const char *)dl + b0 * c->N * es, dtype, max_level, wi + b0 * c->Kw, d_runs ? d_runs + b0 : nullptr,
                                d_ok ? d_ok + b0 : nullptr, d_last ? d_last + 3 * b0 : nullptr, m, rows, ctr, &sp)))
            return rc;
    }
    return POLAR_OK;
}

static int pbf_common(polar_code *c, const void *llr, int dtype, int max_level, void *info_hat, int32_t *runs, uint8_t *crc_ok,
                      int32_t *last_seq, int64_t B, int space, bool packed)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(llr && info_hat, "NULL buffer");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(max_level >= 0 && max_level <= 3, "max_level must be in [0, 3]");
    ARG_TRY(c->crc_w > 0, "bit flipping needs polar_code_set_crc first");
    ARG_TRY(max_level == 0 || c->have_mu, "bit flipping needs polar_code_set_reliability first");
    N_LIMIT_TRY(c, "flip decoder");
    polar_ctx *ctx = c->ctx;
    const size_t es = llr_size(dtype);
    int rc;
    const int64_t chunk = (space == POLAR_HOST) ? HOST_CHUNK : (int64_t)1 << 22;
    HostLlrFeed feed;
    if (space == POLAR_HOST && (rc = feed.begin(ctx, llr, c->N * es, B, chunk))) return rc;
    int64_t idx = 0;
    for (int64_t b0 = 0; b0 < B; b0 += chunk, idx++) {
        const int64_t nb = std::min(chunk, B - b0);
        const char *dl = (const char *)llr + b0 * c->N * es;
        uint32_t *wi;
        int32_t *d_runs = runs ? runs + b0 : nullptr, *d_last = last_seq ? last_seq + b0 * 3 : nullptr;
        uint8_t *d_ok = crc_ok ? crc_ok + b0 : nullptr;
        if (space == POLAR_HOST) {
            if ((rc = ctx->flip_small.reserve(nb * 17)) || (rc = feed.take(idx, &dl))) return rc;
            d_runs = ctx->flip_small.as<int32_t>();
            d_last = d_runs + nb;
            d_ok = reinterpret_cast<uint8_t *>(d_last + 3 * nb);
        }
        if (packed && space == POLAR_DEVICE) {
            wi = (uint32_t *)info_hat + b0 * c->Kw;
        } else {
            if ((rc = ctx->out_words.reserve(nb * c->Kw * 4))) return rc;
            wi = ctx->out_words.as<uint32_t>();
        }
        if ((rc = pbf_device(c, dl, dtype, max_level, wi, d_runs, d_ok, d_last, nb))) return rc;
        if (space == POLAR_HOST && (rc = feed.decoded(idx))) return rc;
        if (packed) {
            if (space == POLAR_HOST)
                HIP_TRY(hipMemcpyAsync((uint32_t *)info_hat + b0 * c->Kw, wi, nb * c->Kw * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else if (space == POLAR_HOST) {
            if ((rc = ctx->out_bytes.reserve(nb * (size_t)c->K + 1))) return rc;
            if ((rc = launch_unpack_bits(ctx, wi, ctx->out_bytes.as<uint8_t>(), nb, c->K))) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t *)info_hat + b0 * c->K, ctx->out_bytes.p, nb * c->K, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            if ((rc = launch_unpack_bits(ctx, wi, (uint8_t *)info_hat + b0 * c->K, nb, c->K))) return rc;
        }
        if (space == POLAR_HOST) {
            if (runs) HIP_TRY(hipMemcpyAsync(runs + b0, d_runs, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (last_seq) HIP_TRY(hipMemcpyAsync(last_seq + b0 * 3, d_last, nb * 12, hipMemcpyDeviceToHost, ctx->stream));
            if (crc_ok) HIP_TRY(hipMemcpyAsync(crc_ok + b0, d_ok, nb, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (space == POLAR_HOST) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (int frc = polar_ctx_check_fault(ctx)) return frc;
    }
    return POLAR_OK;
}

extern "C" int polar_pbf_decode(polar_code *c, const void *llr, int dtype, int max_level, uint8_t *info_hat, int32_t *runs,
                                uint8_t *crc_ok, int32_t *last_seq, int64_t B, int space)
{
    return pbf_common(c, llr, dtype, max_level, info_hat, runs, crc_ok, last_seq, B, space, false);
}
extern "C" int polar_pbf_decode_packed(polar_code *c, const void *llr, int dtype, int max_level, uint32_t *info_hat,
                                       int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B, int space)
{
    return pbf_common(c, llr, dtype, max_level, info_hat, runs, crc_ok, last_seq, B, space, true);
}

// ---- the same search for callers that cannot allocate (HIP graph capture, serving loops): the list of frames that
// failed their CRC and the two counters of the search live in a file-scope device array instead of the context's
// grow-only buffers.  Every GPU holds its own copy of the array at its own address: that address is looked up once per
// device, with the device current, when the first context on it is created, and only the looked-up pointer is ever
// handed to a kernel or a memset (the array's name in host code is not a device address).
static const int64_t PBF_NOALLOC_MAX_FRAMES = (int64_t)1 << 20;
static __device__ __align__(256) unsigned char pbf_scratch[256 + 4 * ((size_t)1 << 20)];  // counters, then one row word per frame
static const int MAX_DEVICES = 64;
static void *g_pbf_scratch[MAX_DEVICES];
static std::mutex g_pbf_scratch_mu;

int polar_noalloc_setup(int device)
{
    if (device < 0 || device >= MAX_DEVICES) return POLAR_OK;  // such a device has no scratch; the entry point says so
    std::lock_guard<std::mutex> lk(g_pbf_scratch_mu);
    if (g_pbf_scratch[device]) return POLAR_OK;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(pbf_scratch)));
    g_pbf_scratch[device] = ptr;
    return POLAR_OK;
}

extern "C" int polar_pbf_decode_noalloc(polar_code *c, const void *llr, int dtype, int max_level, uint32_t *info_hat, int32_t *runs,
                                        uint8_t *crc_ok, int32_t *last_seq, int64_t B)
{
    CHECK_COMMON(c, B, POLAR_DEVICE);
    ARG_TRY(llr && info_hat, "NULL buffer");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(max_level >= 0 && max_level <= 3, "max_level must be in [0, 3]");
    ARG_TRY(c->crc_w > 0, "bit flipping needs polar_code_set_crc first");
    ARG_TRY(max_level == 0 || c->have_mu, "bit flipping needs polar_code_set_reliability first");
    // the timing bracket waits on an event after the launch, which a caller under graph capture must not do
    ARG_TRY(!c->ctx->timing, "polar_pbf_decode_noalloc: switch polar_ctx_enable_timing off (it synchronises after every launch)");
    if (B > PBF_NOALLOC_MAX_FRAMES) {
        polar_set_error("polar_pbf_decode_noalloc: %lld frames exceed the scratch array's capacity of %lld frames per call",
                        (long long)B, (long long)PBF_NOALLOC_MAX_FRAMES);
        return POLAR_EINVAL;
    }
    if (c->n > 10) {
        polar_set_error("polar_pbf_decode_noalloc: N=%d; codes longer than 1024 are decoded with library-allocated workspace", c->N);
        return POLAR_EUNSUPPORTED;
    }
    if (max_level >= 1 && (!c->d_S0 || (max_level >= 2 && !c->have_crit))) {
        polar_set_error("polar_pbf_decode_noalloc: candidate tables missing (polar_code_set_reliability builds them)");
        return POLAR_EINVAL;
    }
    const int dev = c->ctx->device;
    unsigned char *scr = dev < MAX_DEVICES ? (unsigned char *)g_pbf_scratch[dev] : nullptr;
    if (!scr) {
        polar_set_error("polar_pbf_decode_noalloc: no scratch array on device %d", dev);
        return POLAR_EUNSUPPORTED;
    }
    // the counters are cleared by pbf_device_on; the row list needs no clearing (only rows[0 .. row count) are read)
    return pbf_device_on(c, llr, dtype, max_level, info_hat, runs, crc_ok, last_seq, B, reinterpret_cast<uint32_t *>(scr + 256),
                         reinterpret_cast<unsigned long long *>(scr));
}

extern "C" int polar_sc_decode_flip(polar_code *c, const void *llr, int dtype, const int32_t *flips, uint8_t *info_hat,
                                    uint8_t *u_hat, void *dec_llr, int64_t B, int space)
{
    CHECK_COMMON(c, B, space);
    ARG_TRY(llr && flips, "NULL buffer");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    N_LIMIT_TRY(c, "flip decoder");
    polar_ctx *ctx = c->ctx;
    const size_t es = llr_size(dtype);
    int rc;
    if ((rc = ensure_flip_tables(c, false))) return rc;
    const int64_t chunk = (space == POLAR_HOST) ? HOST_CHUNK : (int64_t)1 << 20;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = std::min(chunk, B - b0);
        const char *dl = (const char *)llr + b0 * c->N * es;
        const int32_t *dfl = flips + b0 * 3;
        char *ddec = dec_llr ? (char *)dec_llr + b0 * c->N * es : nullptr;
        if ((rc = ctx->out_words.reserve(nb * c->Kw * 4)) || (rc = ctx->aux_buf.reserve(nb * c->Nw * 4)) || (rc = ctx->flip_ctr.reserve(16)))
            return rc;
        uint32_t *wi = ctx->out_words.as<uint32_t>(), *wu = u_hat ? ctx->aux_buf.as<uint32_t>() : nullptr;
        if (space == POLAR_HOST) {
            if ((rc = ctx->llr_buf.reserve(nb * c->N * es)) || (rc = ctx->flip_small.reserve(nb * 12))) return rc;
            HIP_TRY(hipMemcpyAsync(ctx->llr_buf.p, dl, nb * c->N * es, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(ctx->flip_small.p, dfl, nb * 12, hipMemcpyHostToDevice, ctx->stream));
            dl = ctx->llr_buf.as<char>();
            dfl = ctx->flip_small.as<int32_t>();
            if (dec_llr) {
                if ((rc = ctx->aux2_buf.reserve(nb * c->N * es))) return rc;
                ddec = ctx->aux2_buf.as<char>();
            }
        }
        unsigned long long *ctr = ctx->flip_ctr.as<unsigned long long>();
        HIP_TRY(hipMemsetAsync(ctr, 0, 16, ctx->stream));
        if ((rc = launch_pbf(c, dl, dtype, nullptr, nullptr, ctr, dfl, 0, false, wi, wu, ddec, nullptr, nullptr, nullptr, nb))) return rc;
        if (space == POLAR_HOST) {
            if ((rc = ctx->out_bytes.reserve(nb * ((size_t)c->K + c->N) + 1))) return rc;
            uint8_t *tb = ctx->out_bytes.as<uint8_t>();
            if (info_hat) {
                if ((rc = launch_unpack_bits(ctx, wi, tb, nb, c->K))) return rc;
                if (c->K) HIP_TRY(hipMemcpyAsync(info_hat + b0 * c->K, tb, nb * c->K, hipMemcpyDeviceToHost, ctx->stream));
            }
            if (u_hat) {
                uint8_t *tu = tb + nb * c->K;
                if ((rc = launch_unpack_bits(ctx, wu, tu, nb, c->N))) return rc;
                HIP_TRY(hipMemcpyAsync(u_hat + b0 * c->N, tu, nb * c->N, hipMemcpyDeviceToHost, ctx->stream));
            }
            if (dec_llr) HIP_TRY(hipMemcpyAsync((char *)dec_llr + b0 * c->N * es, ddec, nb * c->N * es, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            if (info_hat && (rc = launch_unpack_bits(ctx, wi, info_hat + b0 * c->K, nb, c->K))) return rc;
            if (u_hat && (rc = launch_unpack_bits(ctx, wu, u_hat + b0 * c->N, nb, c->N))) return rc;
        }
    }
    return POLAR_OK;
}

// ---------------------------------------------------------------------------------------------- dispatch table
// No GPU is needed for these two: they read the static table of host_tables.cpp.
extern "C" int polar_dispatch_select(int op, int N, int K, int dtype, int L, int outputs, int flags, char *kernel, int kernel_len)
{
    ARG_TRY(N >= 1 && (N & (N - 1)) == 0, "N must be a power of two");
    int n = 0;
    while ((1 << n) < N) n++;
    const DispatchQuery q{op, n, K, dtype == POLAR_F32 ? 0 : dtype == POLAR_F64 ? 1 : -1, L, outputs, flags};
    const int row = dispatch_select(q);
    if (row < 0) {
        polar_set_error("no kernel serves this call (op %d, N=%d, K=%d, dtype %d, L=%d, outputs 0x%x, flags 0x%x)", op, N, K, dtype, L, outputs, flags);
        return POLAR_EUNSUPPORTED;
    }
    if (kernel && kernel_len > 0) snprintf(kernel, (size_t)kernel_len, "%s", dispatch_table(nullptr)[row].kernel);
    return row;
}

extern "C" int polar_dispatch_table(char *buf, int len)
{
    int count = 0;
    const DispatchRow *t = dispatch_table(&count);
    std::string out = "row\top\tkernel\tdtypes\tlog2N\tL\tK>=\toutputs\tone_of\tneed\tforbid\twhat\n";
    static const char *ops[] = {"SC", "SCL", "PBF", "MC_SC"};
    char line[512];
    for (int i = 0; i < count; i++) {
        std::string ls;
        for (int l = 1; l <= 32; l++)
            if ((t[i].l_mask >> (l - 1)) & 1u) ls += (ls.empty() ? "" : ",") + std::to_string(l);
        snprintf(line, sizeof(line), "%d\t%s\t%s\t%s%s\t%d..%d\t%s\t%d\t0x%x\t0x%x\t0x%x\t0x%x\t%s\n", i, ops[t[i].op], t[i].kernel,
                 (t[i].dtypes & 1u) ? "f32" : "", (t[i].dtypes & 2u) ? ((t[i].dtypes & 1u) ? ",f64" : "f64") : "", t[i].n_min, t[i].n_max,
                 ls.empty() ? "-" : ls.c_str(), t[i].k_min, t[i].outputs_ok, t[i].outputs_any, t[i].need, t[i].forbid, t[i].what);
        out += line;
    }
    if (buf && len > 0) snprintf(buf, (size_t)len, "%s", out.c_str());
    return (int)out.size() + 1;  // bytes a complete copy needs
}

// ---------------------------------------------------------------------------------------------- counting / MC
extern "C" int polar_count_errors(polar_code *c, const uint32_t *info, const uint32_t *hat, int64_t B, int flags,
                                  uint64_t *counters)
{
    CHECK_COMMON(c, B, POLAR_DEVICE);
    CRC_TRY(c, flags);
    ARG_TRY(info && hat && counters, "NULL buffer");
    return launch_count_errors(c, info, hat, B, counters, (flags & POLAR_CRC) ? c->K - c->crc_w : c->K);
}

// SC decode + error count; nbits = leading bits of each information word that are compared
static int sc_decode_count(polar_code *c, const void *llr, int dtype, const uint32_t *info, uint32_t *info_hat, int64_t B, int flags,
                           uint64_t *counters, int nbits)
{
    if (B == 0) return POLAR_OK;
    int rc;
    if (sc_reg_supported(c, dtype, nullptr, nullptr)) {
        c->count_truth = info;
        c->count_nbits = nbits;
        c->count_dev = counters;
        rc = launch_sc_decode(c, llr, dtype, info_hat, nullptr, nullptr, B, flags);
        c->count_truth = nullptr;
        c->count_dev = nullptr;
        return rc;
    }
    if (!info_hat) {
        if ((rc = c->ctx->mc_hat.reserve(B * c->Kw * 4))) return rc;
        info_hat = c->ctx->mc_hat.as<uint32_t>();
    }
    if ((rc = launch_sc_decode(c, llr, dtype, info_hat, nullptr, nullptr, B, flags))) return rc;
    return launch_count_errors(c, info, info_hat, B, counters, nbits);
}

extern "C" int polar_sc_decode_count(polar_code *c, const void *llr, int dtype, const uint32_t *info, uint32_t *info_hat, int64_t B,
                                     int flags, uint64_t *counters)
{
    CHECK_COMMON(c, B, POLAR_DEVICE);
    ARG_TRY(llr && info && counters, "NULL buffer");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    CRC_TRY(c, flags);
    return sc_decode_count(c, llr, dtype, info, info_hat, B, flags, counters, (flags & POLAR_CRC) ? c->K - c->crc_w : c->K);
}

// Frames per chunk of the Monte-Carlo loop: `deflt`, or POLAR_MC_CHUNK when set (tests use it to reach the
// double-buffered pipeline with small runs)
static int64_t mc_chunk_limit(int64_t deflt)
{
    if (const char *e = polar_switch("POLAR_MC_CHUNK")) {
        const long long v = atoll(e);
        if (v > 0) return std::min<int64_t>(deflt, (int64_t)v);
    }
    return deflt;
}

extern "C" int polar_mc_run(polar_code *c, int decoder, int L, int dtype, int flags, double N0, uint64_t seed,
                            uint64_t frame0, int64_t nframes, uint64_t counters[3])
{
    ARG_TRY(c != nullptr && counters != nullptr, "NULL argument");
    ARG_TRY(nframes >= 0, "nframes must be >= 0");
    ARG_TRY(decoder == POLAR_DEC_SC || decoder == POLAR_DEC_SCL || decoder == POLAR_DEC_PBF,
            "decoder must be POLAR_DEC_SC, POLAR_DEC_SCL or POLAR_DEC_PBF");
    ARG_TRY(dtype == POLAR_F32 || dtype == POLAR_F64, "dtype must be POLAR_F32 or POLAR_F64");
    ARG_TRY(N0 > 0, "N0 must be positive");
    if (decoder == POLAR_DEC_SCL)
        ARG_TRY(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 32, "L must be a power of two in [1, 32]");
    CRC_TRY(c, flags);
    ARG_TRY(!((flags & POLAR_CRC) && decoder == POLAR_DEC_SC), "POLAR_CRC goes with the list or the flip decoder");
    if (decoder == POLAR_DEC_PBF) {
        ARG_TRY(flags & POLAR_CRC, "POLAR_DEC_PBF needs POLAR_CRC");
        ARG_TRY(L >= 0 && L <= 3, "POLAR_DEC_PBF: L is max_level in [0, 3]");
        ARG_TRY(L == 0 || c->have_mu, "bit flipping needs polar_code_set_reliability first");
    }
    ARG_TRY(!((flags & POLAR_GENIE) && decoder != POLAR_DEC_SCL), "POLAR_GENIE goes with the list decoder");
    ARG_TRY(!((flags & POLAR_GENIE) && (flags & POLAR_CRC)), "POLAR_GENIE and POLAR_CRC exclude each other");
    if (decoder != POLAR_DEC_SC) N_LIMIT_TRY(c, decoder == POLAR_DEC_PBF ? "flip decoder" : "list decoder");
    const bool crc = (flags & POLAR_CRC) != 0;
    polar_ctx *ctx = c->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (nframes == 0) return POLAR_OK;
    const size_t es = llr_size(dtype);
    // a code with a specialised decoder runs fp32 SC frames in one kernel, from the random bits to the counters, with no
    // LLR buffer at all (POLAR_MC_KERNEL=staged keeps the generator -> buffer -> decoder pipeline below)
    if (decoder == POLAR_DEC_SC && c->static_mc_fn) {
        const DispatchQuery q{DQ_OP_MC_SC, c->n, c->K, dtype == POLAR_F32 ? 0 : 1, 0, 0,
                              ((flags & POLAR_SC_FAST) ? DQ_FAST : 0) | DQ_SPECIALISED | (crc ? DQ_CRC : 0) | dq_switch_flags("POLAR_MC_KERNEL")};
        const int row = dispatch_select(q);
        if (row >= 0 && dispatch_table(nullptr)[row].family == KF_MC_FUSED) {
            int rc;
            if ((rc = ctx->counters.reserve(3 * sizeof(uint64_t)))) return rc;
            uint64_t *dcnt = ctx->counters.as<uint64_t>();
            HIP_TRY(hipMemsetAsync(dcnt, 0, 3 * sizeof(uint64_t), ctx->stream));
            // the kernel keeps its error sums in 32-bit lane registers until it ends: at most 2^30 frames (or the
            // POLAR_MC_CHUNK of the tests) per launch, so that no run length can wrap them
            const int64_t per = mc_chunk_limit((int64_t)1 << 30);
            for (int64_t f0 = 0; f0 < nframes; f0 += per)
                if ((rc = launch_sc_static_mc(c, std::min(per, nframes - f0), N0, seed, frame0 + f0, dcnt))) return rc;
            uint64_t h[3];
            HIP_TRY(hipMemcpyAsync(h, dcnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < 3; i++) counters[i] += h[i];
            return POLAR_OK;
        }
    }
    // chunks of about 4 GiB of LLRs (10^6 frames at N = 1024: long enough that the kernels' tails do not count), two buffers: chunk i+1 is generated on its own stream while chunk i is
    // decoded and counted (the decoders leave registers and issue slots free that the generator can use)
    int64_t chunk = mc_chunk_limit(std::max<int64_t>(1024, ((int64_t)1 << 32) / ((int64_t)c->N * (int64_t)es)));
    chunk = std::min(chunk, nframes);
    const bool two = nframes > chunk;
    int rc;
    if ((rc = ctx->llr_buf.reserve(chunk * c->N * es)) || (rc = ctx->mc_info.reserve(chunk * c->Kw * 4)) ||
        (rc = ctx->mc_hat.reserve(chunk * c->Kw * 4)) || (rc = ctx->counters.reserve(3 * sizeof(uint64_t))))
        return rc;
    if (two && ((rc = ctx->mc_llr2.reserve(chunk * c->N * es)) || (rc = ctx->mc_info2.reserve(chunk * c->Kw * 4)))) return rc;
    void *llr_b[2] = {ctx->llr_buf.p, two ? ctx->mc_llr2.p : ctx->llr_buf.p};
    uint32_t *info_b[2] = {ctx->mc_info.as<uint32_t>(), two ? ctx->mc_info2.as<uint32_t>() : ctx->mc_info.as<uint32_t>()};
    uint64_t *dcnt = ctx->counters.as<uint64_t>();
    hipStream_t main_stream = ctx->stream;
    HIP_TRY(hipMemsetAsync(dcnt, 0, 3 * sizeof(uint64_t), main_stream));
    // the generator must not overtake work the caller queued earlier on the context's stream
    HIP_TRY(hipEventRecord(ctx->dec_done[1], main_stream));
    HIP_TRY(hipStreamWaitEvent(ctx->gen_stream, ctx->dec_done[1], 0));
    struct Overlap {  // decoders size their grids for a GPU they share with the generator while this loop runs
        polar_ctx *c;
        explicit Overlap(polar_ctx *x) : c(x) { c->overlapped = true; }
        ~Overlap() { c->overlapped = false; }
    } overlap(ctx);
    // the two-stream loop; on an error it returns at once and the caller below drains both streams
    auto pipeline = [&]() -> int {
        int64_t idx = 0;
        for (int64_t f0 = 0; f0 < nframes; f0 += chunk, idx++) {
            const int64_t nb = std::min(chunk, nframes - f0);
            const int b = (int)(idx & 1);
            // generate into buffer b once the decode that last read it (chunk idx-2) is done
            if (idx >= 2) HIP_TRY(hipStreamWaitEvent(ctx->gen_stream, ctx->dec_done[b], 0));
            ctx->stream = ctx->gen_stream;
            int r = launch_gen_frames(c, info_b[b], llr_b[b], dtype, N0, seed, frame0 + f0, nb, crc);
            ctx->stream = main_stream;
            if (r) return r;
            HIP_TRY(hipEventRecord(ctx->gen_done[b], ctx->gen_stream));
            HIP_TRY(hipStreamWaitEvent(main_stream, ctx->gen_done[b], 0));
            if (idx == 1 && polar_fault_at("mc_run")) {
                polar_set_error("polar_mc_run: injected failure in chunk %lld", (long long)idx);
                return POLAR_EHIP;
            }
            if (decoder == POLAR_DEC_SC) {
                if ((r = sc_decode_count(c, llr_b[b], dtype, info_b[b], nullptr, nb, flags, dcnt, c->K))) return r;
                HIP_TRY(hipEventRecord(ctx->dec_done[b], main_stream));
                continue;
            } else if (decoder == POLAR_DEC_PBF)
                r = pbf_device(c, llr_b[b], dtype, L, ctx->mc_hat.as<uint32_t>(), nullptr, nullptr, nullptr, nb);
            else
                r = launch_scl_decode(c, llr_b[b], dtype, L, ctx->mc_hat.as<uint32_t>(), nullptr, nullptr, nb, crc,
                                      (flags & POLAR_GENIE) ? info_b[b] : nullptr);
            if (r) return r;
            if ((r = launch_count_errors(c, info_b[b], ctx->mc_hat.as<uint32_t>(), nb, dcnt, crc ? c->K - c->crc_w : c->K))) return r;
            HIP_TRY(hipEventRecord(ctx->dec_done[b], main_stream));
        }
        return POLAR_OK;
    };
    if ((rc = pipeline())) {
        // nothing stays in flight: the generator may still be writing a buffer that the next call hands out again
        ctx->stream = main_stream;
        (void)hipStreamSynchronize(ctx->gen_stream);
        (void)hipStreamSynchronize(main_stream);
        return rc;
    }
    uint64_t h[3];
    HIP_TRY(hipMemcpyAsync(h, dcnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((rc = polar_ctx_check_fault(ctx))) return rc;
    for (int i = 0; i < 3; i++) counters[i] += h[i];
    return POLAR_OK;
}
