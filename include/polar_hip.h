/*
 * polar_hip.h -- C ABI of libpolar_hip.so, the MI355X (gfx950) implementation of the
 * polar encode -> BPSK/AWGN -> SC / SCL decode path.
 *
 * The reference (Tigrexes/polar-codes) has no FFI: its boundary is the Python object surface of
 * cn/polarCode.py, cn/channel.py and the loop in cn/simulator.py.  Each entry point below names the
 * reference method it stands in for; the Python classes of the same names call these through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative POLAR_E* code on failure;
 *     polar_last_error() returns a thread-local message for the last failure on this thread.
 *   - `space` says where the caller's buffers live: POLAR_HOST (the library stages them through
 *     device memory) or POLAR_DEVICE (HIP device pointers, e.g. torch tensor data_ptr()).
 *   - bits: "bytes" buffers hold one 0/1 value per uint8; "packed" buffers hold 32 bits per
 *     uint32 word, bit i of a frame in word i>>5 at position i&31, frames padded to whole words.
 *   - dtype: POLAR_F32 or POLAR_F64 selects the LLR arithmetic type (float / double).
 *   - calls are stream-ordered on the context's stream and synchronous at the ABI for POLAR_HOST;
 *     for POLAR_DEVICE they return after enqueueing (use polar_ctx_sync or your own stream sync).
 *   - a context and the codes made from it belong to one host thread and one GPU.
 *   - code lengths: N is a power of two, 2 <= N <= 65536, for the encoder, the channel and fp32 SC decoding; fp64 SC decoding
 *     stops at N = 32768 (a frame's tree must fit a wavefront's LDS; POLAR_EUNSUPPORTED beyond); the list, genie and flip decoders take N <= 4096
 *     (POLAR_EUNSUPPORTED beyond, checked before anything is launched).
 */
#ifndef POLAR_HIP_H
#define POLAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct polar_ctx polar_ctx;
typedef struct polar_code polar_code;

enum { POLAR_HOST = 0, POLAR_DEVICE = 1 };
enum { POLAR_F32 = 0, POLAR_F64 = 1 };
enum { POLAR_DEC_SC = 0, POLAR_DEC_SCL = 1, POLAR_DEC_PBF = 2 };
enum {
    POLAR_OK = 0,
    POLAR_EINVAL = -1,   /* bad argument */
    POLAR_EHIP = -2,     /* HIP runtime error (message has the HIP error string) */
    POLAR_ENOMEM = -3,   /* device or host allocation failed */
    POLAR_EUNSUPPORTED = -4
};
/* flags */
enum {
    POLAR_SC_FAST = 1,   /* polar_sc_decode*, polar_mc_run: skip all-frozen subtrees and hard-decide
                            all-information subtrees (same decisions as plain SC; decision LLRs are then
                            not available) */
    POLAR_CRC = 2,       /* polar_gen_frames, polar_scl_decode*, polar_count_errors, polar_mc_run: the last
                            `width` information bits are the CRC set with polar_code_set_crc */
    POLAR_GENIE = 4      /* polar_mc_run with POLAR_DEC_SCL: the genie-aided list decoder (polar_scl_decode_genie);
                            not together with POLAR_CRC */
};

const char *polar_last_error(void);
/* Library version and the GPU architecture the kernels were built for ("gfx950"). */
const char *polar_version(void);

/* ---- context: one GPU, one stream, grow-only device workspace ------------------------------- */
int polar_ctx_create(int device, polar_ctx **out);
int polar_ctx_destroy(polar_ctx *ctx);
/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the context's own. */
int polar_ctx_set_stream(polar_ctx *ctx, void *hip_stream);
int polar_ctx_sync(polar_ctx *ctx);
/* Number of compute units of the context's GPU (for sizing reports). */
int polar_ctx_num_cus(polar_ctx *ctx);

/* ---- code object: replaces the per-code state of PolarCode.__init__ (cn/polarCode.py:116-157):
 *      polarMask / A, plus the decoder schedules compiled from the mask.  mask: N bytes, 1 = info. */
int polar_code_create(polar_ctx *ctx, int N, const uint8_t *mask, polar_code **out);
int polar_code_destroy(polar_code *code);
int polar_code_n(const polar_code *code);
int polar_code_k(const polar_code *code);
/* Number of decoder schedule operations: which = 0 plain SC (visits every leaf), 1 = fast SC. */
int polar_code_schedule_len(const polar_code *code, int which);
/* Compiles, with the HIP run-time compiler, an SC decoder specialised for this code's rate profile (an
 * "unrolled" decoder: the whole tree walk becomes straight-line code) and uses it from then on for
 * polar_sc_decode* / polar_mc_run calls with POLAR_SC_FAST, fp32 LLRs and information-bit output only.
 * Same decisions as the generic kernels.  32 <= N <= 1024, 0 < K < N.  For N >= 128 it also builds the kernel
 * that polar_mc_run then uses for such runs: the whole frame loop (random bits, encoder, channel, decoder,
 * error count) in one kernel with no LLR buffer, counters identical to the staged path.  Takes about a second; meant for long runs
 * over one code, like the per-code set-up of PolarCode.__init__ (cn/polarCode.py:116-157).  No counterpart
 * in the reference.  polar_static_compile_check is the host-only part (no GPU needed): it compiles for `arch`
 * (e.g. "gfx950") and returns the size of the code object in bytes, or a negative error code. */
int polar_code_specialize(polar_code *code);
int polar_static_compile_check(const uint8_t *mask, int N, const char *arch);
/* CRC of CRC-aided list decoding: pcode.crcPoly / pcode.crcWidth (cn/simulator.py:84-85) as consumed by
 * pcf.crcEncode (cn/functions.py:567-583).  crc_poly holds the generator polynomial with its leading term
 * (0x107 = x^8+x^2+x+1); the width is its degree, 1..32 and at most K.  0 removes the CRC.  The K information
 * bits are then K - width message bits followed by the CRC, MSB first. */
int polar_code_set_crc(polar_code *code, uint64_t crc_poly);
int polar_code_crc_width(const polar_code *code);

/* Host-only: writes the decoder schedule compiled from `mask` (one 32-bit word per operation, see
 * polar-codes_amd/csrc/polar_internal.h for the encoding) into out[0..cap) and returns its length, or a
 * negative error code.  fast: bit 0 = short-cut all-frozen / all-information subtrees, bit 1 = 16-leaf nodes as
 * single in-register operations.  Needs no GPU. */
int polar_schedule_build(const uint8_t *mask, int N, int fast, uint32_t *out, int cap);

/* Host-only: the 256-word table of the register SC kernels' output stage for `mask` (32 <= N <= 1024): per lane of a
 * frame's row the masks that compress the information bits of its 16-position pieces and the bit offsets the pieces go to
 * (polar-codes_amd/csrc/host_tables.h has the layout).  No counterpart in the reference (its extract() is
 * uHat[polarMask == 1], cn/polarCode.py:305-312); exposed for the CPU tests.  Needs no GPU. */
int polar_output_tables(const uint8_t *mask, int N, uint32_t *out256);

/* ---- encoder: PolarCode.polarEncode = mul_matrix(rateProfile(info)) (cn/polarCode.py:225-255).
 *      info [B x K] -> x [B x N], x = u F^{(x)n}, natural order, no bit reversal. */
int polar_encode(polar_code *code, const uint8_t *info, uint8_t *x, int64_t B, int space);
int polar_encode_packed(polar_code *code, const uint32_t *info, uint32_t *x, int64_t B, int space);
/* PolarCode.mul_matrix alone (cn/polarCode.py:232-248): u [B x N] -> x [B x N]. */
int polar_mul_matrix(polar_code *code, const uint8_t *u, uint8_t *x, int64_t B, int space);

/* ---- channel: Channel.modulate + addNoise + calcLLR (cn/channel.py:16-22,33-35,53-60) for a batch.
 *      s = 1-2x, y = s + sqrt(N0/2) z, llr = (4/N0) y; z ~ N(0,1) from Philox4x32-10 keyed by `seed`,
 *      counter (frame0 + row, block, stream 1), Box-Muller.  llr is float or double per dtype. */
int polar_channel_llr(polar_code *code, const uint8_t *x, void *llr, int dtype, double N0, uint64_t seed,
                      uint64_t frame0, int64_t B, int space);

/* ---- synthetic Monte-Carlo inputs: the body of the frame loop up to the decoder
 *      (cn/simulator.py:138,148-154): random info bits (Philox stream 0), encode, channel.
 *      Device buffers only.  info_packed [B x ceil(K/32)] (may be NULL), llr [B x N].
 *      With POLAR_CRC only the first K - width bits are random and the CRC is appended before encoding
 *      (PolarCode.CRCPolarEncode2, cn/polarCode.py:265-271). */
int polar_gen_frames(polar_code *code, uint32_t *info_packed, void *llr, int dtype, double N0, uint64_t seed,
                     uint64_t frame0, int64_t B, int flags);

/* ---- SC decoder: PolarCode.scDecoder (cn/polarCode.py:1525-1571) with Path.updateLLRs /
 *      updateBits (cn/polarCode.py:30-85) for a batch.  llr [B x N];
 *      info_hat [B x K] = extract(uHat) (cn/polarCode.py:305-312); u_hat [B x N] or NULL;
 *      dec_llr [B x N] decision LLRs (self.llrs) or NULL -- not available with POLAR_SC_FAST. */
int polar_sc_decode(polar_code *code, const void *llr, int dtype, uint8_t *info_hat, uint8_t *u_hat,
                    void *dec_llr, int64_t B, int flags, int space);
int polar_sc_decode_packed(polar_code *code, const void *llr, int dtype, uint32_t *info_hat, uint32_t *u_hat,
                           void *dec_llr, int64_t B, int flags, int space);

/* ---- SCL decoder: PolarCode.sclDecoder + polarSCLFork (cn/polarCode.py:2032-2102, 969-1005), L <= 32.
 *      info_hat [B x K] from the best path; pm_sorted [B x L] doubles, final path metrics in the
 *      reference's final order (stable, descending) or NULL; u_hat_all [B x L x N] or NULL
 *      (the whole sorted list, self.pathList[j].uHat).  Lists shorter than L (fewer than log2 L
 *      information bits) are padded with -inf metrics and zero bits.
 *      With POLAR_CRC (sclDecoder(isCRC=True), cn/polarCode.py:2079-2097) info_hat comes from the first
 *      path of the sorted list whose K information bits leave a zero CRC remainder, the best path if none
 *      does; it still holds all K bits (the caller drops the last `width`); the list outputs do not change. */
int polar_scl_decode(polar_code *code, const void *llr, int dtype, int L, uint8_t *info_hat, double *pm_sorted,
                     uint8_t *u_hat_all, int64_t B, int flags, int space);
int polar_scl_decode_packed(polar_code *code, const void *llr, int dtype, int L, uint32_t *info_hat,
                            double *pm_sorted, uint32_t *u_hat_all, int64_t B, int flags, int space);

/* ---- genie-aided list decoder: PolarCode.sclDecoderOracle (cn/polarCode.py:1267-1317), the decoder the
 *      reference's driver calls (cn/simulator.py:156): list decoding, then the first entry of the sorted list
 *      that equals the transmitted word (self.message, cn/simulator.py:140-141) is returned, the best entry if
 *      the list does not hold it.  info_true [B x K] = message[polarMask == 1].  Other arguments as in
 *      polar_scl_decode.  The reference forks in place (polarSCLFork2, cn/polarCode.py:1041-1097); its
 *      survivors and sorted list are those of polarSCLFork unless two metrics are exactly equal, where this
 *      library keeps polarSCLFork's order. */
int polar_scl_decode_genie(polar_code *code, const void *llr, int dtype, int L, const uint8_t *info_true,
                           uint8_t *info_hat, double *pm_sorted, uint8_t *u_hat_all, int64_t B, int space);
int polar_scl_decode_genie_packed(polar_code *code, const void *llr, int dtype, int L, const uint32_t *info_true,
                                  uint32_t *info_hat, double *pm_sorted, uint32_t *u_hat_all, int64_t B, int space);

/* ---- which kernel serves a call ---------------------------------------------------------------------
 *      The reference dispatches on a string (PolarCode.decode, cn/polarCode.py:316-350); this library on one static table
 *      (kernel family x arithmetic x log2 N range x list sizes x smallest K x outputs x properties of the call), read in
 *      order by every decoder entry point.  Both functions are host-only and need no GPU.
 *      polar_dispatch_select: row of the table that would serve the call, its kernel name into kernel[0..kernel_len), or
 *      POLAR_EUNSUPPORTED with a message for a call no kernel covers.  op: 0 = SC, 1 = SCL, 2 = flip search, 3 = the SC
 *      Monte-Carlo loop of polar_mc_run.  L: list size / largest number of simultaneous flips.  outputs: 1 information bits,
 *      2 u_hat (SCL: every path of the list), 4 decision LLRs, 8 sorted metrics.  flags: 1 short-cut schedule
 *      (POLAR_SC_FAST), 2 the code was specialised (polar_code_specialize), 4 CRC, 8 genie, 16 caller gives the flip list,
 *      32 workspace of the two-flip search in pieces at hand, 64 / 128 the experiment switches "lds" / "generic".
 *      polar_dispatch_table: the table as tab-separated text into buf[0..len); returns the bytes a complete copy needs. */
int polar_dispatch_select(int op, int N, int K, int dtype, int L, int outputs, int flags, char *kernel, int kernel_len);
int polar_dispatch_table(char *buf, int len);

/* ---- bit flipping ---------------------------------------------------------------------------------
 *      Host-only: pcf.generateCriticalSet((polarMask + 1) % 2) for after < 0, pcf.modifyCriticalSet(polarMask,
 *      after) otherwise (cn/functions.py:279-319): first positions of the maximal all-information subtrees,
 *      ascending, into out[0..cap); returns their number or a negative error code.  Needs no GPU. */
int polar_critical_set(const uint8_t *mask, int N, int after, int32_t *out, int cap);
/* Mean LLRs of the bit channels that rank the flip candidates, mu = pcfun.GA(N, K, designSNR)
 * (cn/polarCode.py:1579): N doubles, host memory. */
int polar_code_set_reliability(polar_code *code, const double *mu);
/* PolarCode.scDecoder(softMess, sequenceU) (cn/polarCode.py:1525-1571 with the hook at :1555-1558) for a batch:
 * flips [B x 3] int32 holds each frame's sequenceU, unused entries -1; the decision at a listed position is
 * inverted (frozen positions included, as in the reference).  Outputs as in polar_sc_decode. */
int polar_sc_decode_flip(polar_code *code, const void *llr, int dtype, const int32_t *flips, uint8_t *info_hat,
                         uint8_t *u_hat, void *dec_llr, int64_t B, int space);
/* PolarCode.progressiveBitFlipping (cn/polarCode.py:1573-1628) for a batch; needs polar_code_set_crc and, for
 * max_level >= 1, polar_code_set_reliability.  max_level = pcode.maxLevel in [0, 3] (0: SC and the CRC test only).
 * info_hat [B x K]: information word of the last SC run made for the frame, CRC bits included (the reference
 * returns it whether or not its CRC holds); runs [B]: SC runs the reference would have made (1 = no flipping);
 * crc_ok [B]: 1 if the returned word passes the CRC; last_seq [B x 3]: flips of the last run, -1 padded.
 * runs, crc_ok, last_seq may be NULL.  Frames whose first SC run passes the CRC never enter the search.  With two or three
 * flips the frames that fail every single flip are searched in (frame, first flip) pieces by all wavefronts of the
 * GPU, so one frame that needs 10^5 runs does not hold the call: same runs, flips, flags and bits as the sequential search. */
int polar_pbf_decode(polar_code *code, const void *llr, int dtype, int max_level, uint8_t *info_hat, int32_t *runs,
                     uint8_t *crc_ok, int32_t *last_seq, int64_t B, int space);
int polar_pbf_decode_packed(polar_code *code, const void *llr, int dtype, int max_level, uint32_t *info_hat,
                            int32_t *runs, uint8_t *crc_ok, int32_t *last_seq, int64_t B, int space);

/* polar_pbf_decode_packed on device buffers (POLAR_DEVICE) for callers that cannot allocate device memory: code running
 * under HIP graph capture, serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The call makes
 * no hipMalloc / hipFree: the list of frames whose first SC run fails the CRC and the two counters of the search live in a
 * file-scope device array of 4 MiB + 256 B that each GPU the library runs on holds until the process ends.
 * Limits: B <= 2^20 frames per call (more is rejected with POLAR_EINVAL), N <= 1024; polar_code_set_crc and (for
 * max_level >= 1) polar_code_set_reliability must have been called: they build every table the search needs.
 * With max_level = 2 or 3 the search runs with one wavefront per failed frame and LDS-staged SC passes (the search in pieces that
 * polar_pbf_decode_packed uses needs a workspace the array has no room for): same results, several times slower with two
 * flips; with three, one wavefront makes the up to 10^5 runs of a frame that needs the third level (tens of seconds at
 * N = 1024 whatever the batch): callers that can allocate should use polar_pbf_decode_packed for three flips.
 * The context's timing bracket (polar_ctx_enable_timing) synchronises after each launch, which capture forbids: with it on
 * the call is refused (POLAR_EINVAL).
 * All calls on one GPU share that array: calls that overlap on one GPU (two streams, two contexts) race. */
int polar_pbf_decode_noalloc(polar_code *code, const void *llr, int dtype, int max_level, uint32_t *info_hat, int32_t *runs,
                             uint8_t *crc_ok, int32_t *last_seq, int64_t B);

/* ---- error counting: pcf.fails per frame (cn/functions.py:107-109) plus the fer/ber counters of the
 *      driver loop (cn/simulator.py:216-226).  Adds to counters[0..2] = frames, frame errors, bit errors
 *      (device uint64[3]).  Packed device buffers.  With POLAR_CRC only the K - width message bits count. */
int polar_count_errors(polar_code *code, const uint32_t *info_packed, const uint32_t *info_hat_packed, int64_t B,
                       int flags, uint64_t *counters_dev);

/* ---- decode and count in one pass: polar_sc_decode_packed followed by polar_count_errors, device buffers
 *      only; where the register-resident kernels apply (fp32, N <= 1024) the comparison happens in the
 *      decoder's output stage and info_hat may be NULL (the decoded words are then not stored at all). */
int polar_sc_decode_count(polar_code *code, const void *llr, int dtype, const uint32_t *info_packed, uint32_t *info_hat,
                          int64_t B, int flags, uint64_t *counters_dev);

/* ---- fused Monte-Carlo run: the frame loop of cn/simulator.py:136-156,216-226 for global frame indices
 *      [frame0, frame0 + nframes).  Nothing but the three counters leaves the GPU.
 *      counters (host uint64[3]) are ADDED to. Frame f always sees the same inputs whatever the chunking
 *      or the number of GPUs, because the generator is keyed by the global frame index.
 *      flags: POLAR_SC_FAST (SC), POLAR_CRC (SCL: CRC-aided frames, selection and counting),
 *      POLAR_GENIE (SCL: genie-aided choice against the generated message).
 *      decoder POLAR_DEC_PBF: progressive bit flipping with L = max_level; needs POLAR_CRC. */
int polar_mc_run(polar_code *code, int decoder, int L, int dtype, int flags, double N0, uint64_t seed,
                 uint64_t frame0, int64_t nframes, uint64_t counters[3]);

/* ---- counter reduction over the GPUs of one node: the reduction of frameError / bit errors over the workers of the
 *      reference's parallel loop (mysimulator.m:85-118, parfor), for callers that drive every GPU from one process
 *      (one polar_ctx per GPU).  polar_comm_init_all makes one RCCL communicator per listed device (ncclCommInitAll;
 *      devices = NULL means 0 .. ndev-1); polar_allreduce_counters sums, in place, n uint64 counters over the devices:
 *      dev_counters[i] is a device pointer on devices[i] (e.g. the uint64[3] that polar_count_errors adds to).  The call
 *      returns when every device holds the totals; decoder work that writes the counters must have finished
 *      (polar_ctx_sync).  RCCL is loaded on first use; without it POLAR_EUNSUPPORTED.  The Python driver
 *      (simulator.simulate) runs one process per GPU and reduces through torch.distributed instead. */
typedef struct polar_comm polar_comm;
int polar_comm_init_all(int ndev, const int *devices, polar_comm **out);
int polar_comm_destroy(polar_comm *comm);
int polar_comm_size(const polar_comm *comm);
int polar_allreduce_counters(polar_comm *comm, uint64_t *const *dev_counters, int n);

/* ---- introspection for benchmarks: name, LDS bytes per frame and lanes per frame of the kernel the
 *      last SC/SCL decode call on this code launched, its grid size, and (when timing is enabled with
 *      polar_ctx_enable_timing) the duration of that launch in milliseconds from HIP events recorded
 *      on the launch stream. */
typedef struct {
    char kernel[64];
    int lanes_per_frame;
    int lds_bytes_per_block;
    int block_threads;
    int grid_blocks;
    float last_ms;
} polar_launch_info;
int polar_ctx_enable_timing(polar_ctx *ctx, int on);
int polar_last_launch(polar_code *code, polar_launch_info *out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* POLAR_HIP_H */
