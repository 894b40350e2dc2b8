// Register-resident SC decoder state, and the decoder specialised at compile time for one rate profile.
//
// This header is device code only (it needs <hip/hip_runtime.h> and <stdint.h>, nothing from the host side), so
// that the run-time compiler can build it: with POLAR_STATIC_NLOG and POLAR_STATIC_WORDS defined (log2 N and
// the N/32 rate-profile words, bit i of word w = position 32 w + i carries information) it also defines the
// kernel sc_static_kernel, in which every decision about the code tree -- which nodes are all-frozen or
// all-information, the profile of every 16-leaf node, the plane a node's sums go to -- is a compile-time
// constant.  The result is straight-line code: no loop, no branch, no scalar bookkeeping in the tree walk
// (unrolled decoders in the sense of Giard et al.).  Arithmetic and results are those of sc_reg_kernel.
#pragma once
#include "reg_common.h"
#include "philox.h"

typedef uint64_t ps_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V> struct IntTag {
    static constexpr int value = V;
};
// f(IntTag<I>{}) for I = BEGIN .. END-1, unrolled at compile time
template <int BEGIN, int END, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (BEGIN < END) {
        f(IntTag<BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// QUADS: the "channel" is an array the caller laid out for 16-byte accesses, element 16 j + lane at index
// 64 (j / 4) + 4 lane + j % 4 (chan_lane then points at index 4 lane); otherwise element 16 j + lane at 16 j + lane.
// REGCH: the "channel" is not in memory at all but in the lane's registers Ch[0 .. 2^(NLOG-4)), element 16 j + lane at
// Ch[j] (the long-code kernel hands its 1024-leaf subtrees over this way).
template <typename T, int NLOG, int KEEP = 0, bool QUADS = false, bool REGCH = false> struct RegDec {
    typedef T value_type;
    T Ch[REGCH ? (1 << (NLOG - 4)) : 1];
    static __device__ __forceinline__ constexpr int chan_index(int j) { return QUADS ? 64 * (j >> 2) + (j & 3) : 16 * j; }
    static constexpr int S = 1 << (NLOG - 4);  // slice length per lane
    T R[S];                                    // level l (4 <= l < NLOG): 2^(l-4) values at offset 2^(l-4)
    // the first KEEP channel pairs of the lane's slice stay in registers between the top-level f and g, the rest
    // is read from memory a second time
    T CA[KEEP > 0 ? KEEP : 1], CB[KEEP > 0 ? KEEP : 1];
    ps_t ps0, ps1;                             // partial sums (plane 0 / 1), same offsets, one bit per value
    ps_t root;

    // f or g from level LEV to LEV-1 (5 <= LEV <= NLOG); the top level reads the channel from memory
    template <int LEV, bool IS_G> __device__ __forceinline__ void step(const T *chan_lane)
    {
        constexpr int c = 1 << (LEV - 5);
        if constexpr (REGCH && LEV == NLOG) {
#pragma unroll
            for (int j = 0; j < c; j++) {
                if constexpr (IS_G) R[c + j] = g_fast(Ch[j], Ch[j + c], (uint32_t)(ps0 >> (c + j)) & 1u);
                else R[c + j] = f_med(Ch[j], Ch[j + c]);
            }
            return;
        }
        if constexpr (QUADS && LEV == NLOG && c >= 4 && sizeof(T) == 4) {
            // four elements of the lane per 16-byte load
#pragma unroll
            for (int J = 0; J < c / 4; J++) {
                const float4 a4 = *reinterpret_cast<const float4 *>(chan_lane + 64 * J);
                const float4 b4 = *reinterpret_cast<const float4 *>(chan_lane + 64 * (J + c / 4));
                const T a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = 4 * J + i;
                    if constexpr (IS_G) R[c + j] = g_fast(a[i], b[i], (uint32_t)(ps0 >> (c + j)) & 1u);
                    else R[c + j] = f_med(a[i], b[i]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < c; j++) {
            T a, b;
            if constexpr (LEV == NLOG) {
                if (IS_G && j < KEEP) {
                    a = CA[j];
                    b = CB[j];
                } else {
                    a = chan_lane[chan_index(j)];
                    b = chan_lane[chan_index(j + c)];
                    if (!IS_G && j < KEEP) {
                        CA[j] = a;
                        CB[j] = b;
                    }
                }
            } else {
                a = R[2 * c + j];
                b = R[3 * c + j];
            }
            if constexpr (IS_G) R[c + j] = g_fast(a, b, (uint32_t)(ps0 >> (c + j)) & 1u);
            else R[c + j] = f_med(a, b);
        }
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane)
    {
        constexpr int c = 1 << (LEV - 5);
        const ps_t m = (c >= 64) ? ~(ps_t)0 : (((ps_t)1 << c) - 1);
        const ps_t xl = (ps0 >> c) & m, xr = (ps1 >> c) & m;
        if constexpr (LEV == NLOG) {
            root = (xl ^ xr) | (c >= 64 ? 0 : (xr << c));
        } else {
            const ps_t x = (xl ^ xr) | (xr << c);
            const ps_t m2 = (((ps_t)1 << (2 * c)) - 1) << (2 * c);
            if (plane) ps1 = (ps1 & ~m2) | (x << (2 * c));
            else ps0 = (ps0 & ~m2) | (x << (2 * c));
        }
    }
    template <int LEV> __device__ __forceinline__ void rate0(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        const ps_t m2 = (((ps_t)1 << c) - 1) << c;
        if (plane) ps1 &= ~m2;
        else ps0 &= ~m2;
    }
    template <int LEV> __device__ __forceinline__ void rate1(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        ps_t x = 0;
#pragma unroll
        for (int j = 0; j < c; j++) x |= (ps_t)(R[c + j] < (T)0 ? 1u : 0u) << j;
        const ps_t m2 = (((ps_t)1 << c) - 1) << c;
        if (plane) ps1 = (ps1 & ~m2) | (x << c);
        else ps0 = (ps0 & ~m2) | (x << c);
    }
};


// ---- compile-time rate profile --------------------------------------------------------------------------------
template <int NW> struct StaticProfile {
    uint32_t w[NW];
    // information leaves of node idx at level lev (2^lev leaves)
    constexpr int ninfo(int lev, int idx) const
    {
        int c = 0;
        for (int i = 0; i < (1 << lev); i++) {
            const int pos = (idx << lev) + i;
            c += (int)((w[pos >> 5] >> (pos & 31)) & 1u);
        }
        return c;
    }
    constexpr uint32_t mask16(int idx) const { return (w[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu; }
};

// Decoder state of the specialised kernel.  The LLR slice is that of RegDec; the partial sums have one register
// per level and plane, X[l - 4][plane] for a node of level l (4 <= l < NLOG), element j of the lane's slice at bit
// 31 - j: a 16-leaf node delivers its bit in bit 31 (other bits arbitrary), which g applies with one masked
// exclusive-or, and combining two
// children is (xl ^ xr) | (xr >> c) with nothing to clear or to insert, since which register a node writes is
// known when the kernel is compiled.
// Of the lane's 2^(NLOG-5) channel pairs the first KEEP stay in registers between the top-level f and g, the next KL in
// LDS (kept[k * 64], k = 2 pair, 2 pair + 1: the wavefront's region, one column per lane); the rest is read again.
// With GEN the channel LLRs are not read but made: the lane's slice of the transmitted codeword is in xlo / xhi, the
// noise comes from the same Philox blocks as in gen_frames_kernel (block j of stream 1 gives the normals of code
// positions 4j .. 4j+3), so the LLRs are bit for bit those the frame generator would have written.
struct ChannelGen {
    uint64_t frame, seed;
    uint32_t xlo, xhi;
    float sigma, gain;
    int sub;
    float *xq;  // 4 floats of LDS per lane, the lane's own at xq[0..3], its quad's first lane's at xq[-4 (sub & 3)]: the 4 x 4 exchange
};
template <int NLOG_, int KEEP = 0, int KL = 0, bool GEN = false> struct StaticDec {
    static constexpr int NLOG = NLOG_;
    static constexpr int S = 1 << (NLOG - 4);
    float R[S];
    float CA[KEEP > 0 ? KEEP : 1], CB[KEEP > 0 ? KEEP : 1];
    float *kept;
    uint32_t X[NLOG - 4][2];
    ps_t root;  // element j at bit j, as the output stage takes it
    ChannelGen cg;

    // LLRs of slice elements t = 4g .. 4g+3 (code positions 16 t + sub).  Lane 4a + b of a row makes Philox block
    // 4 (4g + b) + a, the normals of positions 16 (4g + b) + 4a + i, i = 0..3: element b of what each lane b' of its
    // quad made is what it needs, a 4 x 4 transpose inside the quad (two exchanges).
    template <int G> __device__ __forceinline__ void llr_group(float out[4])
    {
        const int a = cg.sub >> 2, b = cg.sub & 3;
        uint32_t r[4];
        philox4x32_10(cg.frame, (uint32_t)(4 * (4 * G + b) + a), 1u, cg.seed, r);
        float z[4];
        box_muller<float>(r[0], r[1], z[0], z[1]);
        box_muller<float>(r[2], r[3], z[2], z[3]);
#ifdef POLAR_MC_QUAD_DPP
        const bool b0 = (b & 1) != 0, b1 = (b & 2) != 0;
        {
            const float r01 = dpp_mov<0xB1>(b0 ? z[0] : z[1]), r23 = dpp_mov<0xB1>(b0 ? z[2] : z[3]);  // partner b ^ 1
            if (b0) {
                z[0] = r01;
                z[2] = r23;
            } else {
                z[1] = r01;
                z[3] = r23;
            }
        }
        {
            const float r02 = dpp_mov<0x4E>(b1 ? z[0] : z[2]), r13 = dpp_mov<0x4E>(b1 ? z[1] : z[3]);  // partner b ^ 2
            if (b1) {
                z[0] = r02;
                z[1] = r13;
            } else {
                z[2] = r02;
                z[3] = r13;
            }
        }
#else
        // the transpose through LDS: one 16-byte write and two paired reads per lane instead of two exchange steps of
        // selects and DPP moves (16 vector instructions per group; the vector ALU is what this kernel is short of)
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            wave_sync();  // the previous group's reads are done
            *reinterpret_cast<f4 *>(cg.xq) = f4{z[0], z[1], z[2], z[3]};
            wave_sync();
            const float *col = cg.xq - 4 * b + b;  // element b of the quad's lane 0; lane i's is 4 floats further
            z[0] = col[0];
            z[1] = col[4];
            z[2] = col[8];
            z[3] = col[12];
        }
#endif
        const uint32_t xw = (4 * G < 32) ? cg.xlo : cg.xhi;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // BPSK: +1 for a zero, -1 for a one -- the code bit moved to the sign position of the constant 1.0 (two instructions;
            // as a comparison and a select it was three)
            const float s = __uint_as_float(((xw << (31 - ((4 * G + i) & 31))) & 0x80000000u) | 0x3F800000u);
            out[i] = cg.gain * (s + cg.sigma * z[i]);
        }
    }
    // top-level f with generated LLRs: pairs (t, t + 2^(NLOG-5)); every pair is kept (registers, then LDS) for the g
    __device__ __forceinline__ void top_f_generate()
    {
        constexpr int c = 1 << (NLOG - 5);
        static_assert(!GEN || (c >= 4 && KEEP + KL >= c), "generated channel: every pair must stay on chip");
        auto group = [&](auto gtag) {
            constexpr int G = decltype(gtag)::value;
            float a4[4], b4[4];
            llr_group<G>(a4);
            llr_group<G + c / 4>(b4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 4 * G + i;
                if (j < KEEP) {
                    CA[j] = a4[i];
                    CB[j] = b4[i];
                } else {
                    kept[(2 * (j - KEEP)) * 64] = a4[i];
                    kept[(2 * (j - KEEP) + 1) * 64] = b4[i];
                }
                R[c + j] = f_med(a4[i], b4[i]);
            }
            // one group at a time: letting the compiler overlap several of them costs more registers than there are
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, c / 4>(group);
    }

    template <int LEV, bool IS_G> __device__ __forceinline__ void step(const float *chan_lane)
    {
        constexpr int c = 1 << (LEV - 5);
        if constexpr (GEN && LEV == NLOG && !IS_G) {
            top_f_generate();
            return;
        }
        if constexpr (LEV == NLOG && IS_G && !GEN) {
            // the second read of the channel goes through an offset the compiler cannot see is zero: reusing
            // the values of the first read would keep the whole slice in registers, KEEP decides how much of it is
            uint32_t zero = 0;
            asm volatile("" : "+v"(zero));
            chan_lane += zero;
        }
        auto fetch = [&](int j, float &a, float &b) {
            if constexpr (LEV == NLOG) {
                if (IS_G && j < KEEP) {
                    a = CA[j];
                    b = CB[j];
                } else if (IS_G && j < KEEP + KL) {
                    a = kept[(2 * (j - KEEP)) * 64];
                    b = kept[(2 * (j - KEEP) + 1) * 64];
                } else {
                    a = chan_lane[16 * j];
                    b = chan_lane[16 * (j + c)];
                    if (!IS_G && j < KEEP) {
                        CA[j] = a;
                        CB[j] = b;
                    } else if (!IS_G && j < KEEP + KL) {
                        kept[(2 * (j - KEEP)) * 64] = a;
                        kept[(2 * (j - KEEP) + 1) * 64] = b;
                    }
                }
            } else {
                a = R[2 * c + j];
                b = R[3 * c + j];
            }
        };
        const uint32_t sg = vgpr_const(0x80000000u);
        if constexpr (IS_G && c >= 2) {
            // two elements per addition (packed fp32 add); the sum of element j reaches bit 31 by doubling
            uint32_t t = X[LEV - 5][0];
#pragma unroll
            for (int j = 0; j < c; j += 2) {
                float a0, b0, a1, b1;
                fetch(j, a0, b0);
                fetch(j + 1, a1, b1);
                const uint32_t s0 = t & sg;
                t = twice(t);
                const uint32_t s1 = t & sg;
                t = twice(t);
                const f32x2 r = f32x2{b0, b1} + f32x2{__uint_as_float(__float_as_uint(a0) ^ s0), __uint_as_float(__float_as_uint(a1) ^ s1)};
                R[c + j] = r.x;
                R[c + j + 1] = r.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < c; j++) {
                float a, b;
                fetch(j, a, b);
                if constexpr (IS_G) {
                    const uint32_t sgn = X[LEV - 5][0] & sg;  // c == 1
                    R[c + j] = b + __uint_as_float(__float_as_uint(a) ^ sgn);
                } else {
                    R[c + j] = f_med(a, b);
                }
            }
        }
    }
    template <int LEV> __device__ __forceinline__ void combine(int plane)
    {
        constexpr int c = 1 << (LEV - 5);  // elements per child
        uint32_t xl = X[LEV - 5][0], xr = X[LEV - 5][1];
        if constexpr (LEV == 5) {  // 16-leaf nodes deliver bit 31 only
            const uint32_t sg = vgpr_const(0x80000000u);
            xl &= sg;
            xr &= sg;
        }
        if constexpr (LEV == NLOG) {
            if constexpr (c == 32) root = (ps_t)__builtin_bitreverse32(xl ^ xr) | ((ps_t)__builtin_bitreverse32(xr) << 32);
            else root = (ps_t)__builtin_bitreverse32((xl ^ xr) | (xr >> c));
        } else {
            X[LEV - 4][plane] = (xl ^ xr) | (xr >> c);
        }
    }
    template <int LEV> __device__ __forceinline__ void rate0(int plane) { X[LEV - 4][plane] = 0u; }
    template <int LEV> __device__ __forceinline__ void rate1(int plane)
    {
        constexpr int c = 1 << (LEV - 4);
        if constexpr (c == 1) {
            X[0][plane] = decision_word(R[1]);
        } else {
            // (x << 1) | decision per element, then the c bits moved to the top: element j at bit 31 - j
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < c; j++) x = __builtin_amdgcn_alignbit(x, decision_word(R[c + j]), 31);
            X[LEV - 4][plane] = x << (32 - c);
        }
    }
};

// Node idx of level LEV (>= 5), not all-frozen and not all-information; PLANE = 0 left child, 1 right child.
template <typename P, int LEV, int IDX, int PLANE, typename Dec>
__device__ __forceinline__ void decode_static(Dec &d, const float *chan_lane, const RowConst &rc)
{
    constexpr auto prof = P::value();  // StaticProfile<N/32>
    constexpr int full = 1 << (LEV - 1);
    constexpr int nl = prof.ninfo(LEV - 1, 2 * IDX), nr = prof.ninfo(LEV - 1, 2 * IDX + 1);
    // left child (an all-frozen left child needs no LLRs: its sums are zero)
    if constexpr (nl == 0) {
        d.template rate0<LEV - 1>(0);
    } else {
        d.template step<LEV, false>(chan_lane);
        if constexpr (nl == full) d.template rate1<LEV - 1>(0);
        else if constexpr (LEV - 1 == 4) d.X[0][0] = sub_node_s<16, prof.mask16(2 * IDX)>(d.R[1], rc);
        else decode_static<P, LEV - 1, 2 * IDX, 0>(d, chan_lane, rc);
    }
    // right child
    if constexpr (nr == 0) {
        d.template rate0<LEV - 1>(1);
    } else {
        d.template step<LEV, true>(chan_lane);
        if constexpr (nr == full) d.template rate1<LEV - 1>(1);
        else if constexpr (LEV - 1 == 4) d.X[0][1] = sub_node_s<16, prof.mask16(2 * IDX + 1)>(d.R[1], rc);
        else decode_static<P, LEV - 1, 2 * IDX + 1, 1>(d, chan_lane, rc);
    }
    d.template combine<LEV>(PLANE);
}

// Output stage shared with sc_reg_kernel: u_hat = x_hat F^{(x)n} on the sliced layout, staged in LDS, then lane `sub` of the frame's row assembles output words sub and sub + 16 on its own, one staged bit
// per step, and the words leave as contiguous stores.  Asl is the gather table in LDS, [bit i][word w] with
// GATHER_TW words per row: lanes of a row read neighbouring entries, the four rows the same ones.
// x F^{(x)n} of a frame on the sliced layout (element k of the frame = bit k / 16 of lane k mod 16's 64-bit slice lo, hi;
// hi is zero below N = 1024).  The transform is its own inverse: it takes x_hat to u_hat and u to x.
template <int NLOG> __device__ __forceinline__ void slice_transform(uint32_t &lo, uint32_t &hi, int sub)
{
    // stages 4 .. NLOG-1 pair (j, j + 2^(s-4)) inside a lane
    if constexpr (NLOG > 9) lo ^= hi;
    if constexpr (NLOG > 4) lo ^= (lo >> 1) & 0x55555555u;
    if constexpr (NLOG > 5) lo ^= (lo >> 2) & 0x33333333u;
    if constexpr (NLOG > 6) lo ^= (lo >> 4) & 0x0F0F0F0Fu;
    if constexpr (NLOG > 7) lo ^= (lo >> 8) & 0x00FF00FFu;
    if constexpr (NLOG > 8) lo ^= (lo >> 16) & 0x0000FFFFu;
    if constexpr (NLOG > 9) {
        hi ^= (hi >> 1) & 0x55555555u;
        hi ^= (hi >> 2) & 0x33333333u;
        hi ^= (hi >> 4) & 0x0F0F0F0Fu;
        hi ^= (hi >> 8) & 0x00FF00FFu;
        hi ^= (hi >> 16) & 0x0000FFFFu;
    }
    // stages 0 .. 3 pair lanes (l, l + 2^s) of the row; the lower lane of a pair takes the upper one's word.  Inside
    // a quad the word is masked to the upper lanes before the exchange; across quads the exchange itself is limited
    // to the banks (quads) of the lower lanes.
    const uint32_t odd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, odd2 = (sub & 2) ? 0xFFFFFFFFu : 0u;
    lo ^= dpp_mov_u32<0xB1>(lo & odd1);  // quad_perm:[1,0,3,2]
    lo ^= dpp_mov_u32<0x4E>(lo & odd2);  // quad_perm:[2,3,0,1]
    lo ^= dpp_banks_u32<0x104, 0x5>(lo);  // row_shl:4 into quads 0 and 2
    lo ^= dpp_banks_u32<0x108, 0x3>(lo);  // row_shl:8 into quads 0 and 1
    if constexpr (NLOG > 9) {
        hi ^= dpp_mov_u32<0xB1>(hi & odd1);
        hi ^= dpp_mov_u32<0x4E>(hi & odd2);
        hi ^= dpp_banks_u32<0x104, 0x5>(hi);
        hi ^= dpp_banks_u32<0x108, 0x3>(hi);
    }
}

constexpr int SC_STAGE_WORDS = 256;  // per wavefront: 32 words per frame, one frame's word every 32 bytes
constexpr int SC_STATIC_WAVES = 4;   // wavefronts per workgroup of the specialised kernel
template <int NLOG> struct GatherShape {
    static constexpr int TW = (NLOG >= 10) ? 32 : 16;  // words per table row (>= 16: every lane has an entry)
    static constexpr int WORDS = 32 * TW;
};
// Error counting fused into the output stage: per-lane running sums of bit errors, frame errors and frames
// (only the first lane of each frame's row counts), added to the global counters once per wavefront.
struct ErrAcc {
    uint32_t be = 0, fe = 0, fr = 0;
};
// The sums of a wavefront (its four counting lanes are the first of each row), in every lane.
__device__ __forceinline__ ErrAcc err_acc_wave(const ErrAcc &a)
{
    ErrAcc w;
    w.be = w.fe = w.fr = 0u;
#pragma unroll
    for (int r = 0; r < 64; r += 16) {
        w.be += (uint32_t)__builtin_amdgcn_readlane((int)a.be, r);
        w.fe += (uint32_t)__builtin_amdgcn_readlane((int)a.fe, r);
        w.fr += (uint32_t)__builtin_amdgcn_readlane((int)a.fr, r);
    }
    return w;
}
// Adds to the global counters once per workgroup: every atomic of a launch lands on the same cache line, where they are
// served one after the other (a few nanoseconds each), so a launch should bring thousands of them, not millions.
// block3: three words of LDS that were zero at the workgroup's first barrier (unused by workgroups of one wavefront).
// Every thread of the workgroup must call this.
template <int WAVES> __device__ __forceinline__ void err_acc_flush(const ErrAcc &a, unsigned long long *counters, uint32_t *block3)
{
    const ErrAcc w = err_acc_wave(a);
    const int l = threadIdx.x & 63;
    const uint32_t mine = l == 0 ? w.fr : (l == 1 ? w.fe : w.be);  // lanes 0, 1, 2 of a wavefront carry frames, frame errors, bit errors
    if constexpr (WAVES == 1) {
        if (l < 3 && mine) atomicAdd(&counters[l], (unsigned long long)mine);
    } else {
        if (l < 3 && mine) atomicAdd(&block3[l], mine);
        __syncthreads();
        if (threadIdx.x < 3 && block3[threadIdx.x]) atomicAdd(&counters[threadIdx.x], (unsigned long long)block3[threadIdx.x]);
    }
}

// First half of the output stage: u_hat of the four frames staged in LDS (sliced), then the information words the
// lane is responsible for (words sub and sub + 16 of its frame).  The caller synchronises the wavefront before the
// stage is written again.
template <int NLOG>
__device__ __forceinline__ void sc_reg_words(ps_t root, uint32_t *stage, const uint16_t *Asl, int lane, int K, int Kw, bool gather_words,
                                             uint32_t &w_lo, uint32_t &w_hi)
{
    const int grp = lane >> 4, sub = lane & 15;
    uint32_t lo = (uint32_t)root, hi = (uint32_t)(root >> 32);
    slice_transform<NLOG>(lo, hi, sub);
    {
        // staged word w (= 2 lane + half) of the frame in row g at byte 32 w + 4 g: a gather address is then the
        // table entry with its low five bits (the bit index) replaced
        stage[16 * sub + grp] = lo;
        if constexpr (NLOG > 9) stage[16 * sub + 8 + grp] = hi;
    }
    wave_sync();
    w_lo = w_hi = 0u;
    if (gather_words) {
        constexpr int TW = GatherShape<NLOG>::TW;
        const unsigned char *base = reinterpret_cast<const unsigned char *>(stage);
        const uint32_t goff = (uint32_t)grp << 2, wordmask = vgpr_const(~31u);
        auto gather = [&](int w) {
            const uint16_t *t = Asl + w;
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const uint32_t e = t[i * TW];  // (staged word << 5) | bit
                const uint32_t word = *reinterpret_cast<const uint32_t *>(base + and_or(e, wordmask, goff));
                acc = __builtin_amdgcn_alignbit(word >> (e & 31u), acc, 1);  // bit i of acc after 32 steps
            }
            const int valid = K - 32 * w;  // unused table entries point at bit 0 of the slice
            return acc & (valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u)));
        };
        w_lo = gather(sub);
        w_hi = (TW > 16 && Kw > 16) ? gather(sub + 16) : 0u;
    }
}

// Second half of the output stage: store the lane's information words and / or compare them with the transmitted ones.
__device__ __forceinline__ void sc_reg_finish(uint32_t w_lo, uint32_t w_hi, uint32_t *__restrict__ info_hat, int64_t fc, bool active, int lane,
                                              int K, int Kw, const uint32_t *__restrict__ truth, int nbits, ErrAcc *acc, bool truth_regs,
                                              uint32_t tr_lo, uint32_t tr_hi)
{
    const int sub = lane & 15;
    if (info_hat || truth || truth_regs) {
        if (info_hat && active) {
            uint32_t *dst = info_hat + fc * Kw;
            if (sub < Kw) dst[sub] = w_lo;
            if (sub + 16 < Kw) dst[sub + 16] = w_hi;
            if (K == 0 && sub == 0) dst[0] = 0u;
        }
        if (truth || truth_regs) {
            // the transmitted words: lane `sub` holds words sub and sub + 16 itself, or they come from memory
            if (!truth_regs) {
                const uint32_t *tw = truth + fc * Kw;
                tr_lo = sub < Kw ? tw[sub] : 0u;
                tr_hi = sub + 16 < Kw ? tw[sub + 16] : 0u;
            }
            uint32_t e = 0;
            if (sub < Kw) {
                const int valid = nbits - 32 * sub;
                const uint32_t m = valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
                e += __popc((w_lo ^ tr_lo) & m);
            }
            if (sub + 16 < Kw) {
                const int valid = nbits - 32 * (sub + 16);
                const uint32_t m = valid >= 32 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
                e += __popc((w_hi ^ tr_hi) & m);
            }
            // sum over the 16 lanes of the frame's row
            e += dpp_mov_u32<0x128>(e);
            e += dpp_mov_u32<0x124>(e);
            e += dpp_mov_u32<0x122>(e);
            e += dpp_mov_u32<0x121>(e);
            if (active && sub == 0) {
                acc->be += e;
                acc->fe += e ? 1u : 0u;
                acc->fr += 1u;
            }
        }
    }
}

// truth (packed [B x Kw], or NULL): the transmitted information words; their first nbits bits are compared with the
// decoded ones and the result goes into acc.  info_hat may then be NULL (nothing is stored).  With truth_regs the
// transmitted words are in tr_lo / tr_hi (lane sub of the frame's row: words sub and sub + 16).
template <int NLOG>
__device__ __forceinline__ void sc_reg_epilogue(ps_t root, uint32_t *stage, const uint16_t *Asl, uint32_t *__restrict__ info_hat,
                                                int64_t fc, bool active, int lane, int K, int Kw,
                                                const uint32_t *__restrict__ truth = nullptr, int nbits = 0, ErrAcc *acc = nullptr,
                                                bool truth_regs = false, uint32_t tr_lo = 0u, uint32_t tr_hi = 0u)
{
    uint32_t w_lo, w_hi;
    sc_reg_words<NLOG>(root, stage, Asl, lane, K, Kw, info_hat || truth || truth_regs, w_lo, w_hi);
    sc_reg_finish(w_lo, w_hi, info_hat, fc, active, lane, K, Kw, truth, nbits, acc, truth_regs, tr_lo, tr_hi);
    wave_sync();
}

// ---- the output stage without a gather ------------------------------------------------------------------------
// 16 x 16 bit transpose across the lanes of a row, in every halfword of x at once: bit b of lane s becomes bit s of lane b.
// Four exchange steps (lane distances 8, 4, 2, 1); a step swaps the bit blocks of that size between partner lanes.
__device__ __forceinline__ uint32_t row_transpose16(uint32_t x, const RowConst &rc)
{
    {   // distance 8: whole bytes, one byte permute
        const uint32_t t = dpp_mov_u32<0x128>(x);  // row_ror:8
        const uint32_t sel = (rc.lowmask[0] & 0x05050505u) ^ 0x03070105u;  // lower lanes x.b0 t.b0 x.b2 t.b2, upper t.b1 x.b1 t.b3 x.b3
        x = __builtin_amdgcn_perm(t, x, sel);
    }
    {   // distance 4: the partner's word rotated by a nibble, merged under a lane-dependent mask
        uint32_t t = dpp_banks_u32<0x104, 0x5>(x);                                          // row_shl:4 into quads 0 and 2
        t = (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)x, 0x114, 0xF, 0xA, false);  // row_shr:4 into quads 1 and 3
        const uint32_t amt = 4u ^ (rc.lowmask[1] & 24u), m = 0xF0F0F0F0u ^ rc.lowmask[1];
        x = __builtin_amdgcn_bitop3_b32(x, __builtin_amdgcn_alignbit(t, t, amt), m, 0xE4);   // (x & m) | (rot & ~m)
    }
    {
        const uint32_t t = dpp_mov_u32<0x4E>(x);  // quad_perm:[2,3,0,1]
        const uint32_t amt = 2u ^ (rc.lowmask[2] & 28u), m = 0xCCCCCCCCu ^ rc.lowmask[2];
        x = __builtin_amdgcn_bitop3_b32(x, __builtin_amdgcn_alignbit(t, t, amt), m, 0xE4);
    }
    {
        const uint32_t t = dpp_mov_u32<0xB1>(x);  // quad_perm:[1,0,3,2]
        const uint32_t amt = 1u ^ (rc.lowmask[3] & 30u), m = 0xAAAAAAAAu ^ rc.lowmask[3];
        x = __builtin_amdgcn_bitop3_b32(x, __builtin_amdgcn_alignbit(t, t, amt), m, 0xE4);
    }
    return x;
}
// the information bits of both halfwords of x moved to the low end of their halfword (c: mask and four move masks)
__device__ __forceinline__ uint32_t compress_halves(uint32_t x, uint32_t m, uint32_t mv0, uint32_t mv1, uint32_t mv2, uint32_t mv3)
{
    x &= m;
    uint32_t t;
    t = x & mv0;
    x = (x ^ t) | (t >> 1);
    t = x & mv1;
    x = (x ^ t) | (t >> 2);
    t = x & mv2;
    x = (x ^ t) | (t >> 4);
    t = x & mv3;
    x = (x ^ t) | (t >> 8);
    return x;
}
// the inverse: the low bits of both halfwords of x (already cut to as many bits as the halfword has information
// positions) moved up to those positions; the moves of compress_halves undone in reverse order
__device__ __forceinline__ uint32_t expand_halves(uint32_t x, uint32_t m, uint32_t mv0, uint32_t mv1, uint32_t mv2, uint32_t mv3)
{
    uint32_t t;
    t = x & (mv3 >> 8);
    x = (x ^ t) | (t << 8);
    t = x & (mv2 >> 4);
    x = (x ^ t) | (t << 4);
    t = x & (mv1 >> 2);
    x = (x ^ t) | (t << 2);
    t = x & (mv0 >> 1);
    x = (x ^ t) | (t << 1);
    return x & m;
}
// The rate profile u[polarMask == 1] = info (cn/polarCode.py:225-230) on the sliced layout, as the inverse of the output
// stage below: the frame's information words lie in row[0 .. Kw) (LDS, 64 words per row, the rest zero or anything);
// every lane cuts the pieces of its (at most four) 16-position halfwords out of them at the bit offsets of the table,
// expands them to the information positions and the 16 x 16 transpose across the row puts them on the sliced layout.
// Returns the lane's two slice words of u (zero at frozen positions).
template <int NLOG>
__device__ __forceinline__ void sc_reg_profile_c(const uint32_t *row, const uint32_t *ctab, int sub, const RowConst &rc, uint32_t &u_lo, uint32_t &u_hi)
{
    const uint4 c0 = *reinterpret_cast<const uint4 *>(ctab + 4 * sub);          // mask and moves 0 .. 2 of slice word 0
    const uint4 c1 = *reinterpret_cast<const uint4 *>(ctab + 64 + 4 * sub);     // move 3 of word 0; mask, moves 0, 1 of word 1
    const uint4 c2 = *reinterpret_cast<const uint4 *>(ctab + 128 + 4 * sub);    // moves 2, 3 of word 1; places of halfwords 0, 1
    const uint4 c3 = *reinterpret_cast<const uint4 *>(ctab + 192 + 4 * sub);    // places of halfwords 2, 3; low masks of words 0, 1
    auto piece = [&](uint32_t e) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(row) + (e & 0xFFFFu));
        return __builtin_amdgcn_alignbit(src[1], src[0], e >> 16);  // 32 bits of the stream from the piece's first bit on
    };
    uint32_t lo = piece(c2.z) & 0xFFFFu;
    if constexpr (NLOG > 8) lo |= piece(c2.w) << 16;
    u_lo = row_transpose16(expand_halves(lo & c3.z, c0.x, c0.y, c0.z, c0.w, c1.x), rc);
    u_hi = 0u;
    if constexpr (NLOG > 9) {
        const uint32_t hi = (piece(c3.x) & 0xFFFFu) | (piece(c3.y) << 16);
        u_hi = row_transpose16(expand_halves(hi & c3.w, c1.y, c1.z, c1.w, c2.x, c2.y), rc);
    }
}
// Information words sub and sub + 16 of the lane's frame from the root partial sums, without touching one bit at a time:
// u_hat on the sliced layout, a 16 x 16 bit transpose across the row (lane l then holds code positions 16 (l + 16 q) ..
// + 15 in halfword q of its slice), the information bits of every halfword compressed to its low end, and the (at most
// four) pieces of a lane placed at their bit offset in the frame's words with LDS atomics.  ctab: build_compress_tables'
// 256 words in LDS; out: 256 words of LDS per wavefront (64 per row).  The caller synchronises the wavefront before out is
// written again.
template <int NLOG>
__device__ __forceinline__ void sc_reg_words_c(ps_t root, uint32_t *out, const uint32_t *ctab, int lane, const RowConst &rc, uint32_t &w_lo,
                                               uint32_t &w_hi)
{
    const int grp = lane >> 4, sub = lane & 15;
    uint32_t lo = (uint32_t)root, hi = (uint32_t)(root >> 32);
    slice_transform<NLOG>(lo, hi, sub);
    uint32_t *row = out + 64 * grp;
    row[sub] = 0u;
    row[sub + 16] = 0u;
    row[sub + 32] = 0u;
    const uint4 c0 = *reinterpret_cast<const uint4 *>(ctab + 4 * sub);          // mask and moves 0 .. 2 of slice word 0
    const uint4 c1 = *reinterpret_cast<const uint4 *>(ctab + 64 + 4 * sub);     // move 3 of word 0; mask, moves 0, 1 of word 1
    const uint4 c2 = *reinterpret_cast<const uint4 *>(ctab + 128 + 4 * sub);    // moves 2, 3 of word 1; places of halfwords 0, 1
    lo = compress_halves(row_transpose16(lo, rc), c0.x, c0.y, c0.z, c0.w, c1.x);
    wave_sync();
    auto place = [&](uint32_t piece, uint32_t e) {
        const uint64_t v = (uint64_t)piece << (e >> 16);
        uint32_t *dst = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(row) + (e & 0xFFFFu));
        // an LDS atomic costs by the lanes it serves: empty pieces and pieces that end inside their first word stay out
        if ((uint32_t)v) atomicOr(dst, (uint32_t)v);
        if ((uint32_t)(v >> 32)) atomicOr(dst + 1, (uint32_t)(v >> 32));
    };
    place(lo & 0xFFFFu, c2.z);
    if constexpr (NLOG > 8) place(lo >> 16, c2.w);
    if constexpr (NLOG > 9) {
        const uint2 c3 = *reinterpret_cast<const uint2 *>(ctab + 192 + 4 * sub);  // places of halfwords 2, 3
        hi = compress_halves(row_transpose16(hi, rc), c1.y, c1.z, c1.w, c2.x, c2.y);
        place(hi & 0xFFFFu, c3.x);
        place(hi >> 16, c3.y);
    }
    wave_sync();
    w_lo = row[sub];
    w_hi = row[sub + 16];
}
template <int NLOG>
__device__ __forceinline__ void sc_reg_epilogue_c(ps_t root, uint32_t *out, const uint32_t *ctab, const RowConst &rc, uint32_t *__restrict__ info_hat,
                                                  int64_t fc, bool active, int lane, int K, int Kw, const uint32_t *__restrict__ truth = nullptr,
                                                  int nbits = 0, ErrAcc *acc = nullptr)
{
    uint32_t w_lo, w_hi;
    sc_reg_words_c<NLOG>(root, out, ctab, lane, rc, w_lo, w_hi);
    sc_reg_finish(w_lo, w_hi, info_hat, fc, active, lane, K, Kw, truth, nbits, acc, false, 0u, 0u);
    wave_sync();
}

#if defined(POLAR_STATIC_NLOG) && defined(POLAR_STATIC_WORDS)
struct PolarStaticProfile {
    static constexpr StaticProfile<((1 << POLAR_STATIC_NLOG) + 31) / 32> value()
    {
        return StaticProfile<((1 << POLAR_STATIC_NLOG) + 31) / 32>{{POLAR_STATIC_WORDS}};
    }
};

// Channel pairs per lane that stay on chip between the top-level f and g: POLAR_STATIC_KEEP in registers (20 is what
// fits beside the decoder state at four wavefronts per SIMD: 24 spill) and up to POLAR_STATIC_KEEP_LDS more in LDS; what
// is left is read a second time, mostly from the L2 / Infinity Cache.  20 + 12 measured 0.5 % faster than 16 + 16.
#ifndef POLAR_STATIC_KEEP
#define POLAR_STATIC_KEEP 20
#endif
#ifndef POLAR_STATIC_KEEP_LDS
#define POLAR_STATIC_KEEP_LDS 12
#endif
#ifndef POLAR_STATIC_WAVES
#define POLAR_STATIC_WAVES 4
#endif
struct PolarStaticShape {
    static constexpr int NLOG = POLAR_STATIC_NLOG;
    static constexpr int PAIRS = 1 << (NLOG - 5);
    static constexpr int KEEP = POLAR_STATIC_KEEP < PAIRS ? POLAR_STATIC_KEEP : PAIRS;
    static constexpr int KL = (PAIRS - KEEP) < POLAR_STATIC_KEEP_LDS ? (PAIRS - KEEP) : POLAR_STATIC_KEEP_LDS;
    static constexpr int KEPT_WORDS = 2 * KL * 64;  // per wavefront
};
// Workgroups of SC_STATIC_WAVES wavefronts share the gather table; everything else is per wavefront (the wavefronts
// of a workgroup never wait for each other after the table is loaded).
extern "C" __global__ void __launch_bounds__(64 * SC_STATIC_WAVES, POLAR_STATIC_WAVES) sc_static_kernel(
    const float *__restrict__ llr, const uint32_t *__restrict__ ctab, uint32_t *__restrict__ info_hat, long long B, int K, int Kw,
    const uint32_t *__restrict__ truth, int nbits, unsigned long long *__restrict__ counters)
{
    constexpr int NLOG = POLAR_STATIC_NLOG;
    constexpr int N = 1 << NLOG;
    constexpr int KEPT = PolarStaticShape::KEPT_WORDS;
    __shared__ __attribute__((aligned(32))) uint32_t stage_s[SC_STATIC_WAVES * SC_STAGE_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t ctab_s[256];
    __shared__ float kept_s[SC_STATIC_WAVES * (KEPT > 0 ? KEPT : 1)];
    __shared__ uint32_t count_s[3];
    for (int k = threadIdx.x; k < 256; k += 64 * SC_STATIC_WAVES) ctab_s[k] = ctab[k];
    if (threadIdx.x < 3) count_s[threadIdx.x] = 0u;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: frame offsets stay in scalar registers
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    uint32_t *stage = stage_s + wave * SC_STAGE_WORDS;
    ErrAcc acc;
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (sub & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }
    StaticDec<NLOG, PolarStaticShape::KEEP, PolarStaticShape::KL> d;
    d.kept = kept_s + wave * KEPT + lane;
    const long long stride = (long long)gridDim.x * (4 * SC_STATIC_WAVES);
    for (long long f0 = ((long long)blockIdx.x * SC_STATIC_WAVES + wave) * 4; f0 < B; f0 += stride) {
        const long long f = f0 + grp;
        const bool active = f < B;
        const long long fc = active ? f : B - 1;
        decode_static<PolarStaticProfile, NLOG, 0, 0>(d, llr + fc * N + sub, rc);
        sc_reg_epilogue_c<NLOG>(d.root, stage, ctab_s, rc, info_hat, fc, active, lane, K, Kw, truth, nbits, &acc);
    }
    if (truth) err_acc_flush<SC_STATIC_WAVES>(acc, counters, count_s);
}

#if POLAR_STATIC_NLOG >= 7  // a Philox block spans four slice elements: below N = 128 a lane has fewer pairs than that
// Monte-Carlo frames without a channel buffer: information bits (Philox stream 0), rate profile and encoder, BPSK,
// AWGN and LLRs (stream 1), the decoder above and the error count, all in registers and LDS of the wavefront that owns
// the four frames.  Counters and every intermediate value are those of gen_frames_kernel followed by
// sc_static_kernel: same Philox blocks, same arithmetic.  counters: frames, frame errors, bit errors.
extern "C" __global__ void __launch_bounds__(64 * SC_STATIC_WAVES, POLAR_STATIC_WAVES) sc_static_mc_kernel(
    const uint32_t *__restrict__ ctab, long long B, int K, int Kw, float sigma, float gain, unsigned long long seed,
    unsigned long long frame0, unsigned long long *__restrict__ counters)
{
    constexpr int NLOG = POLAR_STATIC_NLOG;
    constexpr int PAIRS = PolarStaticShape::PAIRS;
    // nothing can be read again: what is not in registers waits in LDS, at most 16 pairs (8 KB per wavefront) of it
    constexpr int KL = PAIRS > 16 ? 16 : 0;  // this kernel has the generator's registers to feed as well: 16 pairs in registers
    constexpr int KEEP = PAIRS - KL;
    constexpr int KEPT = 2 * KL * 64;
    __shared__ __attribute__((aligned(32))) uint32_t stage_s[SC_STATIC_WAVES * SC_STAGE_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t ctab_s[256];
    __shared__ float kept_s[SC_STATIC_WAVES * (KEPT > 0 ? KEPT : 1)];
    __shared__ uint32_t count_s[3];
    for (int k = threadIdx.x; k < 256; k += 64 * SC_STATIC_WAVES) ctab_s[k] = ctab[k];
    if (threadIdx.x < 3) count_s[threadIdx.x] = 0u;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    uint32_t *stage = stage_s + wave * SC_STAGE_WORDS;
    ErrAcc acc;
    RowConst rc;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rc.up[i] = (sub & (8 >> i)) != 0;
        rc.lowmask[i] = rc.up[i] ? 0u : 0xFFFFFFFFu;
        rc.low31[i] = rc.up[i] ? 0u : 0x80000000u;
    }
    StaticDec<NLOG, KEEP, KL, true> d;
    d.kept = kept_s + wave * KEPT + lane;
    d.cg.seed = seed;
    d.cg.sigma = sigma;
    d.cg.gain = gain;
    d.cg.sub = sub;
    d.cg.xq = reinterpret_cast<float *>(stage) + 4 * lane;  // the staging words are idle while the channel is made
    const long long stride = (long long)gridDim.x * (4 * SC_STATIC_WAVES);
    for (long long f0 = ((long long)blockIdx.x * SC_STATIC_WAVES + wave) * 4; f0 < B; f0 += stride) {
        const long long f = f0 + grp;
        const bool active = f < B;
        const long long fc = active ? f : B - 1;
        d.cg.frame = frame0 + (unsigned long long)fc;
        // information words sub and sub + 16 of the frame, as gen_frames_kernel draws them
        uint32_t iw[2] = {0u, 0u};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int w = sub + 16 * s;
            if (s == 0 || Kw > 16) {
                uint32_t r[4];
                philox4x32_10(d.cg.frame, (uint32_t)(w >> 2), 0u, seed, r);
                uint32_t v = (w & 2) ? ((w & 1) ? r[3] : r[2]) : ((w & 1) ? r[1] : r[0]);
                const int valid = K - 32 * w;
                v = valid >= 32 ? v : (valid <= 0 ? 0u : (v & ((1u << valid) - 1u)));
                iw[s] = v;
            }
        }
        // rate profile: the information words go to the frame's row of the staging area and every lane takes the bits of
        // its own code positions out of them (the output stage run backwards; no bit is moved on its own)
        uint32_t *row = stage + 64 * grp;
        wave_sync();  // the last frame's exchanges through these words are done
        row[sub] = iw[0];
        row[sub + 16] = iw[1];
        row[sub + 32] = 0u;
        wave_sync();
        uint32_t xlo, xhi;
        {
            // the table words are the same for every frame of the lane: read through an index the compiler cannot see through,
            // or it keeps all sixteen in registers across the decoder (22 spill slots)
            int sq = sub;
            asm volatile("" : "+v"(sq));
            sc_reg_profile_c<NLOG>(row, ctab_s, sq, rc, xlo, xhi);
        }
        wave_sync();
        const uint32_t ulo = xlo, uhi = xhi;   // the transmitted u, sliced: zero at frozen positions
        slice_transform<NLOG>(xlo, xhi, sub);  // u -> x
        d.cg.xlo = xlo;
        d.cg.xhi = xhi;
        decode_static<PolarStaticProfile, NLOG, 0, 0>(d, nullptr, rc);
        // errors are counted where the decoder's result lies, in the sliced u domain: u_hat is zero at frozen positions as
        // well, so the differing bits are exactly the wrong information bits and no bit has to be gathered
        uint32_t hlo = (uint32_t)d.root, hhi = (uint32_t)(d.root >> 32);
        slice_transform<NLOG>(hlo, hhi, sub);  // x_hat -> u_hat
        uint32_t e = __popc(hlo ^ ulo) + ((NLOG > 9) ? __popc(hhi ^ uhi) : 0u);
        e += dpp_mov_u32<0x128>(e);
        e += dpp_mov_u32<0x124>(e);
        e += dpp_mov_u32<0x122>(e);
        e += dpp_mov_u32<0x121>(e);
        if (active && sub == 0) {
            acc.be += e;
            acc.fe += e ? 1u : 0u;
            acc.fr += 1u;
        }
    }
    err_acc_flush<SC_STATIC_WAVES>(acc, counters, count_s);
}
#endif
#endif
