// Run-time specialisation: builds kernels_sc_static.h for one rate profile with hiprtc and launches the result.
#include <hip/hiprtc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "polar_internal.h"
#include "build/embedded_sources.h"  // POLAR_SRC_decode_common, POLAR_SRC_reg_common, POLAR_SRC_philox, POLAR_SRC_kernels_sc_static

static int compile_static(const uint8_t *mask, int N, const char *arch, std::vector<char> &code)
{
    int n = 0;
    while ((1 << n) < N) n++;
    std::string src;
    // experiment knobs: channel pairs kept in registers / in LDS per lane, and the occupancy the compiler aims for
    if (const char *e = polar_switch("POLAR_STATIC_KEEP")) src += "#define POLAR_STATIC_KEEP " + std::to_string(atoi(e)) + "\n";
    if (const char *e = polar_switch("POLAR_STATIC_KEEP_LDS")) src += "#define POLAR_STATIC_KEEP_LDS " + std::to_string(atoi(e)) + "\n";
    if (const char *e = polar_switch("POLAR_STATIC_WAVES")) src += "#define POLAR_STATIC_WAVES " + std::to_string(atoi(e)) + "\n";
    src += "#define POLAR_STATIC_NLOG " + std::to_string(n) + "\n#define POLAR_STATIC_WORDS ";
    for (int w = 0; w < (N + 31) / 32; w++) {
        uint32_t v = 0;
        for (int i = 0; i < 32 && 32 * w + i < N; i++) v |= (uint32_t)(mask[32 * w + i] ? 1 : 0) << i;
        char buf[24];
        snprintf(buf, sizeof(buf), "%s0x%08Xu", w ? "," : "", v);
        src += buf;
    }
    src += "\n#include \"kernels_sc_static.h\"\n";
    const char *names[] = {"decode_common.h", "reg_common.h", "philox.h", "kernels_sc_static.h"};
    const char *texts[] = {POLAR_SRC_decode_common, POLAR_SRC_reg_common, POLAR_SRC_philox, POLAR_SRC_kernels_sc_static};
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "polar_static.hip", 4, texts, names);
    if (r != HIPRTC_SUCCESS) {
        polar_set_error("hiprtcCreateProgram failed: %s", hiprtcGetErrorString(r));
        return POLAR_EUNSUPPORTED;
    }
    const std::string archopt = std::string("--offload-arch=") + arch;
    // -ffinite-math-only: the decoder never produces or tests NaN / infinity; it lets min(|a|,|b|) be one instruction
    const char *opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-ffinite-math-only"};
    r = hiprtcCompileProgram(prog, 5, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        polar_set_error("run-time compilation failed (%s): %.1500s", hiprtcGetErrorString(r), log.c_str());
        hiprtcDestroyProgram(&prog);
        return POLAR_EUNSUPPORTED;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    // POLAR_STATIC_DUMP=<file>: keep the code object (llvm-objdump -d shows what the specialisation produced)
    if (const char *e = polar_switch("POLAR_STATIC_DUMP")) {
        if (FILE *fp = fopen(e, "wb")) {
            fwrite(code.data(), 1, code.size(), fp);
            fclose(fp);
        }
    }
    return POLAR_OK;
}

static int check_profile(const uint8_t *mask, int N)
{
    ARG_TRY(mask != nullptr, "mask is NULL");
    ARG_TRY(N >= 32 && N <= 1024 && (N & (N - 1)) == 0, "specialised decoder: N must be a power of two in [32, 1024]");
    int K = 0;
    for (int i = 0; i < N; i++) K += mask[i] ? 1 : 0;
    ARG_TRY(K > 0 && K < N, "specialised decoder: needs 0 < K < N");
    return POLAR_OK;
}

extern "C" int polar_static_compile_check(const uint8_t *mask, int N, const char *arch)
{
    int rc = check_profile(mask, N);
    if (rc) return rc;
    ARG_TRY(arch != nullptr, "arch is NULL");
    std::vector<char> code;
    if ((rc = compile_static(mask, N, arch, code))) return rc;
    return (int)code.size();
}

extern "C" int polar_code_specialize(polar_code *c)
{
    ARG_TRY(c != nullptr, "code is NULL");
    if (c->static_fn) return POLAR_OK;
    int rc = check_profile(c->mask.data(), c->N);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->ctx->device));
    std::string arch(prop.gcnArchName);
    arch = arch.substr(0, arch.find(':'));  // "gfx950:sramecc+:xnack-" -> "gfx950"
    std::vector<char> code;
    if ((rc = compile_static(c->mask.data(), c->N, arch.c_str(), code))) return rc;
    HIP_TRY(hipModuleLoadData(&c->static_mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&c->static_fn, c->static_mod, "sc_static_kernel"));
    if (hipModuleGetFunction(&c->static_mc_fn, c->static_mod, "sc_static_mc_kernel") != hipSuccess) {
        c->static_mc_fn = nullptr;  // short codes have no fused Monte-Carlo kernel
        (void)hipGetLastError();
    }
    return POLAR_OK;
}

int launch_sc_static(polar_code *c, const void *llr, uint32_t *info_hat, int64_t B)
{
    if (B == 0) return POLAR_OK;
    polar_ctx *ctx = c->ctx;
    constexpr int WAVES = 4;  // wavefronts per workgroup (SC_STATIC_WAVES in the kernel source)
    int occ = 0;
    HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, c->static_fn, 64 * WAVES, 0));
    if (occ < 1) occ = 1;
    const int64_t per_block = 4 * WAVES;
    const int64_t need = (B + per_block - 1) / per_block;
    // workgroups per resident slot: measured 8.94 / 8.27 / 8.14 / 7.98 ms per 10^7 frames with 1 / 4 / 8 / 32 (more only adds
    // dispatch time); next to the frame generator of the staged Monte-Carlo loop the long queue starves the other stream
    // (241 against 407 M frames/s for the loop), so 8 there
    int rounds = ctx->overlapped ? 8 : 32;
    if (const char *e = polar_switch("POLAR_STATIC_ROUNDS")) rounds = std::max(1, atoi(e));
    const int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ * rounds);
    const void *a_llr = llr;
    const void *a_asl = c->d_ctab;  // the output stage's compress / place table
    void *a_out = info_hat;
    long long a_B = (long long)B;
    int a_K = c->K, a_Kw = c->Kw, a_nbits = c->count_nbits;
    const void *a_truth = c->count_truth;
    void *a_cnt = c->count_dev;
    void *args[] = {&a_llr, &a_asl, &a_out, &a_B, &a_K, &a_Kw, &a_truth, &a_nbits, &a_cnt};
    LaunchTimer tm(ctx);
    HIP_TRY(hipModuleLaunchKernel(c->static_fn, (unsigned)grid, 1, 1, 64 * WAVES, 1, 1, 0, ctx->stream, args, nullptr));
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_static_kernel<%d>", c->n);
    c->last.lanes_per_frame = 16;
    int lds_bytes = 0;
    (void)hipFuncGetAttribute(&lds_bytes, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, c->static_fn);
    c->last.lds_bytes_per_block = lds_bytes;
    c->last.block_threads = 64 * WAVES;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}

int launch_sc_static_mc(polar_code *c, int64_t B, double N0, uint64_t seed, uint64_t frame0, uint64_t *counters_dev)
{
    if (B == 0) return POLAR_OK;
    polar_ctx *ctx = c->ctx;
    constexpr int WAVES = 4;
    int occ = 0;
    HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, c->static_mc_fn, 64 * WAVES, 0));
    if (occ < 1) occ = 1;
    const int64_t per_block = 4 * WAVES;
    const int64_t need = (B + per_block - 1) / per_block;
    const int64_t grid = std::min<int64_t>(need, (int64_t)ctx->num_cus * occ * 32);
    const void *a_asl = c->d_ctab;  // the output stage's table, read backwards by the encoder side
    long long a_B = (long long)B;
    int a_K = c->K, a_Kw = c->Kw;
    float a_sigma = (float)sqrt(N0 / 2.0), a_gain = (float)(4.0 / N0);  // as launch_gen_frames passes them
    unsigned long long a_seed = seed, a_frame0 = frame0;
    void *a_cnt = counters_dev;
    void *args[] = {&a_asl, &a_B, &a_K, &a_Kw, &a_sigma, &a_gain, &a_seed, &a_frame0, &a_cnt};
    LaunchTimer tm(ctx);
    HIP_TRY(hipModuleLaunchKernel(c->static_mc_fn, (unsigned)grid, 1, 1, 64 * WAVES, 1, 1, 0, ctx->stream, args, nullptr));
    float ms = tm.stop();
    snprintf(c->last.kernel, sizeof(c->last.kernel), "sc_static_mc_kernel<%d>", c->n);
    c->last.lanes_per_frame = 16;
    int lds_bytes = 0;
    (void)hipFuncGetAttribute(&lds_bytes, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, c->static_mc_fn);
    c->last.lds_bytes_per_block = lds_bytes;
    c->last.block_threads = 64 * WAVES;
    c->last.grid_blocks = (int)grid;
    c->last.last_ms = ms;
    return POLAR_OK;
}
