// Bit packing, the u.F^{(x)n} butterfly encoder, the BPSK/AWGN channel with Philox noise, the fused
// frame generator and the error counter.  All integer/bit work is one 32-bit word per lane; the
// cross-word butterfly stages go through LDS.
#include <algorithm>

#include "philox.h"
#include "polar_internal.h"

#define BLOCK 256

// ------------------------------------------------------------------------------------------ pack / unpack
__global__ void __launch_bounds__(BLOCK) pack_bits_kernel(const uint8_t *__restrict__ bytes, uint32_t *__restrict__ words,
                                                         int64_t total_words, int nbits, int nw)
{
    int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= total_words) return;
    int64_t frame = t / nw;
    int w = (int)(t - frame * nw);
    const uint8_t *src = bytes + frame * nbits + w * 32;
    int lim = min(32, nbits - w * 32);
    uint32_t v = 0;
    for (int b = 0; b < lim; b++) v |= (uint32_t)(src[b] & 1u) << b;
    words[t] = v;
}

__global__ void __launch_bounds__(BLOCK) unpack_bits_kernel(const uint32_t *__restrict__ words, uint8_t *__restrict__ bytes,
                                                           int64_t total_bits, int nbits, int nw)
{
    int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= total_bits) return;
    int64_t frame = t / nbits;
    int b = (int)(t - frame * nbits);
    bytes[t] = (words[frame * nw + (b >> 5)] >> (b & 31)) & 1u;
}

int launch_pack_bits(polar_ctx *ctx, const uint8_t *bytes, uint32_t *words, int64_t B, int nbits)
{
    int nw = nbits > 0 ? (nbits + 31) / 32 : 1;
    int64_t total = B * nw;
    if (total == 0) return POLAR_OK;
    if (nbits == 0) {
        HIP_TRY(hipMemsetAsync(words, 0, total * 4, ctx->stream));
        return POLAR_OK;
    }
    hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, bytes, words,
                       total, nbits, nw);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}

int launch_unpack_bits(polar_ctx *ctx, const uint32_t *words, uint8_t *bytes, int64_t B, int nbits)
{
    int nw = nbits > 0 ? (nbits + 31) / 32 : 1;
    int64_t total = B * nbits;
    if (total == 0) return POLAR_OK;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, words,
                       bytes, total, nbits, nw);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}

// ------------------------------------------------------------------------------------------ encoder
// Stages 0..4 of x[j] ^= x[j + 2^s] (bit s of j clear) stay inside one 32-bit word.
__device__ __forceinline__ uint32_t butterfly_in_word(uint32_t x, int n)
{
    if (n > 0) x ^= (x >> 1) & 0x55555555u;
    if (n > 1) x ^= (x >> 2) & 0x33333333u;
    if (n > 2) x ^= (x >> 4) & 0x0F0F0F0Fu;
    if (n > 3) x ^= (x >> 8) & 0x00FF00FFu;
    if (n > 4) x ^= (x >> 16) & 0x0000FFFFu;
    return x;
}

// Scatter the frame's K info bits into the information positions of word m (rateProfile).
__device__ __forceinline__ uint32_t rate_profile_word(const uint32_t *info, uint32_t maskw, uint32_t rank)
{
    uint32_t u = 0;
    uint32_t r = rank;
    while (maskw) {
        int b = __ffs(maskw) - 1;
        maskw &= maskw - 1;
        u |= ((info[r >> 5] >> (r & 31)) & 1u) << b;
        r++;
    }
    return u;
}

// Cross-word stages on a block-resident array of `nwords` words holding whole frames of Nw words.
__device__ __forceinline__ void butterfly_cross_words(uint32_t *lds, int nwords, int Nw, int n)
{
    for (int s = 5; s < n; s++) {
        __syncthreads();
        int d = 1 << (s - 5);
        for (int idx = threadIdx.x; idx < nwords; idx += BLOCK) {
            int m = idx & (Nw - 1);
            if (!(m & d)) lds[idx] ^= lds[idx + d];
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BLOCK) encode_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ x,
                                                      const uint32_t *__restrict__ maskw, const uint32_t *__restrict__ rank,
                                                      int64_t B, int Nw, int Kw, int n, int fpb, int rate_profile)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    int64_t f0 = (int64_t)blockIdx.x * fpb;
    int nfr = (int)min((int64_t)fpb, B - f0);
    int nwords = nfr * Nw;
    for (int idx = threadIdx.x; idx < nwords; idx += BLOCK) {
        int fl = idx / Nw, m = idx - fl * Nw;
        int64_t f = f0 + fl;
        uint32_t u = rate_profile ? rate_profile_word(in + f * Kw, maskw[m], rank[m]) : in[f * Nw + m];
        lds[idx] = butterfly_in_word(u, n);
    }
    butterfly_cross_words(lds, nwords, Nw, n);
    for (int idx = threadIdx.x; idx < nwords; idx += BLOCK) x[f0 * Nw + idx] = lds[idx];
}

static int frames_per_block(int Nw) { return Nw >= BLOCK ? 1 : BLOCK / Nw; }

int launch_encode(polar_code *c, const uint32_t *in, uint32_t *x, int64_t B, bool rate_profile)
{
    if (B == 0) return POLAR_OK;
    int fpb = frames_per_block(c->Nw);
    int64_t blocks = (B + fpb - 1) / fpb;
    size_t lds = (size_t)fpb * c->Nw * 4;
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)blocks), dim3(BLOCK), lds, c->ctx->stream, in, x, c->d_maskw, c->d_rank, B,
                       c->Nw, c->Kw, c->n, fpb, rate_profile ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}

// ------------------------------------------------------------------------------------------ channel
template <typename T> struct ChanConst {
    T sigma, gain;
};

// Four LLRs (code positions 4j..4j+3 of one frame) from one Philox block.
template <typename T>
__device__ __forceinline__ void llr4(uint64_t seed, uint64_t frame, uint32_t j, uint32_t xbits4, T sigma, T gain, T out[4])
{
    uint32_t r[4];
    philox4x32_10(frame, j, 1u, seed, r);
    T z[4];
    box_muller<T>(r[0], r[1], z[0], z[1]);
    box_muller<T>(r[2], r[3], z[2], z[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        T s = ((xbits4 >> i) & 1u) ? (T)-1 : (T)1;
        out[i] = gain * (s + sigma * z[i]);
    }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) channel_kernel(const uint32_t *__restrict__ xw, T *__restrict__ llr, int64_t B, int N,
                                                       int Nw, T sigma, T gain, uint64_t seed, uint64_t frame0)
{
    int q = (N + 3) / 4;  // Philox blocks per frame
    int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= B * q) return;
    int64_t f = t / q;
    int j = (int)(t - f * q);
    uint32_t bits = (xw[f * Nw + (j >> 3)] >> ((j & 7) * 4)) & 0xFu;
    T o[4];
    llr4<T>(seed, frame0 + (uint64_t)f, (uint32_t)j, bits, sigma, gain, o);
    T *dst = llr + f * N + 4 * j;
    if ((N & 3) == 0) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            *reinterpret_cast<double2 *>(dst) = make_double2(o[0], o[1]);
            *reinterpret_cast<double2 *>(dst + 2) = make_double2(o[2], o[3]);
        }
    } else {
        for (int i = 0; i < 4 && 4 * j + i < N; i++) dst[i] = o[i];
    }
}

int launch_channel(polar_code *c, const uint32_t *xw, void *llr, int dtype, double N0, uint64_t seed, uint64_t frame0, int64_t B)
{
    if (B == 0) return POLAR_OK;
    int q = (c->N + 3) / 4;
    int64_t total = B * q;
    dim3 grid((unsigned)((total + BLOCK - 1) / BLOCK));
    double sigma = sqrt(N0 / 2.0), gain = 4.0 / N0;
    if (dtype == POLAR_F32)
        hipLaunchKernelGGL(channel_kernel<float>, grid, dim3(BLOCK), 0, c->ctx->stream, xw, (float *)llr, B, c->N, c->Nw,
                           (float)sigma, (float)gain, seed, frame0);
    else
        hipLaunchKernelGGL(channel_kernel<double>, grid, dim3(BLOCK), 0, c->ctx->stream, xw, (double *)llr, B, c->N, c->Nw, sigma,
                           gain, seed, frame0);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}

// ------------------------------------------------------------------------------------------ frame generator
// info bits (Philox stream 0) -> rate profile -> butterfly (LDS) -> BPSK + AWGN -> LLR, one block per
// `fpb` frames.  The K info bits are also written out (packed, zero padded) for the error counter.
// With a CRC table (crcE != NULL) only the first Km = K - crc_w bits are random; the CRC of those, a GF(2)
// linear map given per message bit by crcE, fills the last crc_w information bits.
template <typename T>
__global__ void __launch_bounds__(BLOCK) gen_frames_kernel(uint32_t *__restrict__ info_out, T *__restrict__ llr,
                                                          const uint32_t *__restrict__ maskw, const uint32_t *__restrict__ rank,
                                                          const uint32_t *__restrict__ crcE, int Km, int crc_w,
                                                          int64_t B, int N, int Nw, int K, int Kw, int n, int fpb, T sigma, T gain,
                                                          uint64_t seed, uint64_t frame0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *xw = lds;               // fpb * Nw
    uint32_t *iw = lds + fpb * Nw;    // fpb * Kw
    uint32_t *crcv = iw + fpb * Kw;   // fpb
    int64_t f0 = (int64_t)blockIdx.x * fpb;
    int nfr = (int)min((int64_t)fpb, B - f0);
    for (int idx = threadIdx.x; idx < nfr * Kw; idx += BLOCK) {
        int fl = idx / Kw, w = idx - fl * Kw;
        uint32_t r[4];
        philox4x32_10(frame0 + (uint64_t)(f0 + fl), (uint32_t)(w >> 2), 0u, seed, r);
        uint32_t v = r[w & 3];
        int valid = Km - 32 * w;
        if (valid < 32) v = valid <= 0 ? 0u : (v & ((1u << valid) - 1u));
        iw[idx] = v;
        if (idx < nfr) crcv[idx] = 0u;
    }
    __syncthreads();
    if (crcE) {
        for (int idx = threadIdx.x; idx < nfr * Kw; idx += BLOCK) {
            int fl = idx / Kw, w = idx - fl * Kw;
            uint32_t v = iw[idx], acc = 0u;
            while (v) {
                acc ^= crcE[32 * w + __ffs(v) - 1];
                v &= v - 1;
            }
            if (acc) atomicXor(&crcv[fl], acc);
        }
        __syncthreads();
        for (int fl = threadIdx.x; fl < nfr; fl += BLOCK) {
            const uint32_t c = crcv[fl];
            const int sh = Km & 31;
            iw[fl * Kw + (Km >> 5)] |= c << sh;
            if (sh + crc_w > 32) iw[fl * Kw + (Km >> 5) + 1] |= c >> (32 - sh);
        }
        __syncthreads();
    }
    if (info_out)
        for (int idx = threadIdx.x; idx < nfr * Kw; idx += BLOCK) info_out[f0 * Kw + idx] = iw[idx];
    // N is a power of two, so words per frame and Philox blocks per frame are too: shifts, not divisions
    const int nwlog = n > 5 ? n - 5 : 0, qlog = n > 2 ? n - 2 : 0;
    int nwords = nfr * Nw;
    for (int idx = threadIdx.x; idx < nwords; idx += BLOCK) {
        int fl = idx >> nwlog, m = idx & (Nw - 1);
        xw[idx] = butterfly_in_word(rate_profile_word(iw + fl * Kw, maskw[m], rank[m]), n);
    }
    butterfly_cross_words(xw, nwords, Nw, n);
    int q = (N + 3) / 4;
    for (int idx = threadIdx.x; idx < nfr * q; idx += BLOCK) {
        int fl = idx >> qlog, j = idx & (q - 1);
        uint32_t bits = (xw[fl * Nw + (j >> 3)] >> ((j & 7) * 4)) & 0xFu;
        T o[4];
        llr4<T>(seed, frame0 + (uint64_t)(f0 + fl), (uint32_t)j, bits, sigma, gain, o);
        T *dst = llr + (f0 + fl) * N + 4 * j;
        if ((N & 3) == 0) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                *reinterpret_cast<double2 *>(dst) = make_double2(o[0], o[1]);
                *reinterpret_cast<double2 *>(dst + 2) = make_double2(o[2], o[3]);
            }
        } else {
            for (int i = 0; i < 4 && 4 * j + i < N; i++) dst[i] = o[i];
        }
    }
}

int launch_gen_frames(polar_code *c, uint32_t *info, void *llr, int dtype, double N0, uint64_t seed, uint64_t frame0, int64_t B,
                      bool crc)
{
    if (B == 0) return POLAR_OK;
    int fpb = frames_per_block(c->Nw);
    int64_t blocks = (B + fpb - 1) / fpb;
    size_t lds = (size_t)fpb * (c->Nw + c->Kw + 1) * 4;
    double sigma = sqrt(N0 / 2.0), gain = 4.0 / N0;
    const uint32_t *crcE = crc ? c->d_crcE : nullptr;
    const int crc_w = crc ? c->crc_w : 0, Km = c->K - crc_w;
    if (dtype == POLAR_F32)
        hipLaunchKernelGGL(gen_frames_kernel<float>, dim3((unsigned)blocks), dim3(BLOCK), lds, c->ctx->stream, info, (float *)llr,
                           c->d_maskw, c->d_rank, crcE, Km, crc_w, B, c->N, c->Nw, c->K, c->Kw, c->n, fpb, (float)sigma, (float)gain,
                           seed, frame0);
    else
        hipLaunchKernelGGL(gen_frames_kernel<double>, dim3((unsigned)blocks), dim3(BLOCK), lds, c->ctx->stream, info, (double *)llr,
                           c->d_maskw, c->d_rank, crcE, Km, crc_w, B, c->N, c->Nw, c->K, c->Kw, c->n, fpb, sigma, gain, seed, frame0);
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}

// ------------------------------------------------------------------------------------------ error counter
// Bits of word w (of a packed frame) that lie below bit index nbits of the frame.
__device__ __forceinline__ uint32_t leading_bits_mask(int nbits, int w)
{
    int valid = nbits - 32 * w;
    return valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
}

// One lane per 32-bit word (coalesced); the words of a frame sit in Kw consecutive lanes (Kw a power of two
// <= 64), so a frame's bit errors are a segmented butterfly sum over those lanes.
__global__ void __launch_bounds__(BLOCK) count_errors_words_kernel(const uint32_t *__restrict__ info, const uint32_t *__restrict__ hat,
                                                                  int64_t total_words, int Kw, int nbits,
                                                                  unsigned long long *__restrict__ counters)
{
    unsigned int be = 0, fe = 0, fr = 0;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t - (threadIdx.x & 63) < total_words; t += (int64_t)gridDim.x * BLOCK) {
        unsigned int e = t < total_words ? __popc((info[t] ^ hat[t]) & leading_bits_mask(nbits, (int)(t & (Kw - 1)))) : 0u;
        for (int off = 1; off < Kw; off <<= 1) e += __shfl_xor(e, off);
        if ((t & (Kw - 1)) == 0 && t < total_words) {
            be += e;
            fe += e > 0 ? 1u : 0u;
            fr += 1u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        be += __shfl_down(be, off);
        fe += __shfl_down(fe, off);
        fr += __shfl_down(fr, off);
    }
    if ((threadIdx.x & 63) == 0 && fr) {
        atomicAdd(&counters[0], (unsigned long long)fr);
        if (fe) atomicAdd(&counters[1], (unsigned long long)fe);
        if (be) atomicAdd(&counters[2], (unsigned long long)be);
    }
}

// General case: one frame per lane.
__global__ void __launch_bounds__(BLOCK) count_errors_kernel(const uint32_t *__restrict__ info, const uint32_t *__restrict__ hat,
                                                            int64_t B, int Kw, int nbits, unsigned long long *__restrict__ counters)
{
    int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    unsigned int be = 0;
    if (f < B)
        for (int w = 0; w < Kw; w++) be += __popc((info[f * Kw + w] ^ hat[f * Kw + w]) & leading_bits_mask(nbits, w));
    unsigned int fe = be > 0 ? 1u : 0u, fr = f < B ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        be += __shfl_down(be, off);
        fe += __shfl_down(fe, off);
        fr += __shfl_down(fr, off);
    }
    if ((threadIdx.x & 63) == 0 && fr) {
        atomicAdd(&counters[0], (unsigned long long)fr);
        if (fe) atomicAdd(&counters[1], (unsigned long long)fe);
        if (be) atomicAdd(&counters[2], (unsigned long long)be);
    }
}

int launch_count_errors(polar_code *c, const uint32_t *info, const uint32_t *hat, int64_t B, uint64_t *counters, int nbits)
{
    if (B == 0) return POLAR_OK;
    const int Kw = c->Kw;
    if (Kw <= 64 && (Kw & (Kw - 1)) == 0) {
        int64_t total = B * Kw;
        int64_t blocks = std::min<int64_t>((total + BLOCK - 1) / BLOCK, (int64_t)c->ctx->num_cus * 16);
        hipLaunchKernelGGL(count_errors_words_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, c->ctx->stream, info, hat, total, Kw,
                           nbits, (unsigned long long *)counters);
    } else {
        hipLaunchKernelGGL(count_errors_kernel, dim3((unsigned)((B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->ctx->stream, info, hat, B,
                           Kw, nbits, (unsigned long long *)counters);
    }
    HIP_TRY(hipGetLastError());
    return POLAR_OK;
}
