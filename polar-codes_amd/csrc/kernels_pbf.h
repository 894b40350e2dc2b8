// Arguments shared by the kernels of the bit-flipping search (kernels_pbf.hip: one wavefront per SC run, LLR tree in
// LDS, any N <= 4096; kernels_pbf_reg.hip: four SC runs per wavefront in registers, N = 32 .. 1024).
#pragma once
#include <stdint.h>

template <typename T> struct PBFParams {
    const T *llr;
    const uint32_t *rows;                 // frame index of work item w, or NULL for w itself
    const unsigned long long *nrows_dev;  // number of work items when rows != NULL
    unsigned long long *work;             // atomic work counter (zero at launch)
    const uint32_t *ops;                  // every-leaf schedule (16-leaf nodes as single operations when N >= 32)
    const uint32_t *A;
    const uint32_t *crcS;                 // NULL: no CRC test (flip-list mode)
    const int32_t *S0;                    // level-0 candidates (critical set)
    const int32_t *crit_off;              // CSR over code positions: candidates after a flip at position i
    const int32_t *crit_idx;
    const double *sqrt_mu;
    const int32_t *flips_in;              // [B x 3] fixed flip lists (flip-list mode), or NULL
    uint32_t *info_hat;                   // [B x Kw]
    uint32_t *u_hat;                      // [B x Nw] or NULL
    T *dec_llr;                           // [B x N] or NULL
    int32_t *runs;                        // [B] or NULL
    uint8_t *crc_ok;                      // [B] or NULL
    int32_t *last_seq;                    // [B x 3] or NULL
    // split two-flip search (all NULL otherwise): frames that fail every single flip, their count, their ranked
    // first-level candidates (nS0 per frame), and per frame the smallest successful pair found so far as
    // rank of first flip : rank of second flip : first flip : second flip, 16 bits each (all ones = none)
    uint32_t *rows2;
    unsigned long long *nrows2;
    uint16_t *order_rec;
    unsigned long long *best;
    // split three-flip search: per frame the smallest successful triple found so far (triple_key in kernels_pbf_reg.hip), and
    // per code position the number of candidates two levels below a flip there (sum of the candidate counts of its candidates)
    unsigned long long *best3;
    const int32_t *crit_sum;
    int64_t B;
    int N, n, Nw, K, Kw, nS0, maxS, max_level;
};

__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
    for (int off = 1; off < 64; off <<= 1) v ^= (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}


// the smallest successful pair a frame has so far (other wavefronts update it while the search runs)
__device__ __forceinline__ unsigned long long load_best(const unsigned long long *a)
{
    return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

