// The generic (run-time rate profile) walk of a register-resident SC subtree of up to 1024 leaves: shared by
// sc_reg_kernel and by the decoder for longer codes, which hands it one 1024-leaf subtree at a time.
#pragma once
#include "kernels_sc_static.h"

// Node classes per tree level (bit i of level l = node i of that level, l = 4 .. 10 -> index l-4):
// r0 = every leaf frozen, r1 = every leaf information.  In plain (non-fast) mode both are all zero.
struct NodeClasses {
    uint64_t r0[7];
    uint64_t r1[7];
};

// The 16-leaf decoder that visits every leaf and keeps the decision LLR of leaf e in lane e of the row (`keep`):
// the every-leaf schedule with the reference's self.llrs as a by-product.  Same decisions as sub_node<16, false>.
template <int P, int OFF, typename T>
__device__ __forceinline__ uint32_t sub_node_dl(T v, uint32_t mask, const RowConst &rc, int sub, T &keep)
{
    if constexpr (P == 1) {
        keep = (sub == OFF) ? v : keep;
        return pin(v < (T)0 ? 1u : 0u) & mask;
    } else {
        constexpr uint32_t HALF = (1u << (P / 2)) - 1u;
        constexpr int I = P == 16 ? 0 : P == 8 ? 1 : P == 4 ? 2 : 3;
        const T o = partner<P>(v);
        const T vl = f_fast(v, o);
        const uint32_t xl = sub_node_dl<P / 2, OFF>(vl, mask & HALF, rc, sub, keep);
        const uint32_t y = xl & rc.lowmask[I];
        const T t = flip_sign(v, y);
        const T vr = t + partner<P>(t);
        const uint32_t xr = sub_node_dl<P / 2, OFF + P / 2>(vr, (mask >> (P / 2)) & HALF, rc, sub, keep);
        return xr ^ y;
    }
}

// A mixed node at level LEV >= 5, index idx within its level.  The two children are visited by a
// two-trip loop that is deliberately NOT unrolled: every level's code then exists exactly once in the
// program (F, G, rate-0, rate-1, combine), the recursion is a fixed nest of NLOG-4 loops, and the
// register array keeps fixed homes.
// DL: every leaf is visited and the decision LLRs of 16-leaf node i go to dl_lane[16 i] (dl_lane = the frame's
// decision-LLR row + the lane's index in its row), one coalesced store per node.
template <int LEV, int NLOG, bool FAST, typename Dec, bool DL = false>
__device__ __forceinline__ void decode_node(Dec &d, const NodeClasses &nc, uint32_t maskv, uint32_t pidv,
                                            const typename Dec::value_type *chan_lane, uint32_t idx, int plane, const RowConst &rc,
                                            typename Dec::value_type *dl_lane = nullptr, int sub = 0)
{
#pragma nounroll
    for (uint32_t h = 0; h < 2; h++) {
        if (h == 0) d.template step<LEV, false>(chan_lane);
        else d.template step<LEV, true>(chan_lane);
        const uint32_t cidx = 2 * idx + h;
        const bool is0 = FAST && ((nc.r0[LEV - 5] >> cidx) & 1ull);
        const bool is1 = FAST && ((nc.r1[LEV - 5] >> cidx) & 1ull);
        if (is0) {
            d.template rate0<LEV - 1>((int)h);
        } else if (is1) {
            d.template rate1<LEV - 1>((int)h);
        } else {
            if constexpr (LEV - 1 == 4) {
                // lane i of maskv holds word i of the rate profile; the index is wave-uniform
                const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)maskv, (int)(cidx >> 1));
                const uint32_t m16 = (mw >> (16 * (cidx & 1))) & 0xFFFFu;
                uint32_t x;
                if constexpr (DL) {
                    typename Dec::value_type keep = 0;
                    x = sub_node_dl<16, 0>(d.R[1], m16, rc, sub, keep);
                    if (dl_lane) dl_lane[16 * cidx] = keep;
                } else if constexpr (sizeof(typename Dec::value_type) == 4) {
                    // lane i of pidv names the profile of 16-leaf node i (0: none of the compiled ones)
                    const uint32_t pid = FAST ? (uint32_t)__builtin_amdgcn_readlane((int)pidv, (int)cidx) : 0u;
                    switch (pid) {
#define POLAR_CASE16(ID, MASK) \
    case ID: x = sub_node_s<16, MASK>(d.R[1], rc) >> 31; break;
                        POLAR_PROFILES16(POLAR_CASE16)
#undef POLAR_CASE16
                    default: x = sub_node<16, FAST>(d.R[1], m16, rc); break;
                    }
                } else {
                    x = sub_node<16, FAST>(d.R[1], m16, rc);
                }
                if (h) d.ps1 = (d.ps1 & ~(ps_t)2) | ((ps_t)x << 1);
                else d.ps0 = (d.ps0 & ~(ps_t)2) | ((ps_t)x << 1);
            } else {
                decode_node<LEV - 1, NLOG, FAST, Dec, DL>(d, nc, maskv, pidv, chan_lane, cidx, (int)h, rc, dl_lane, sub);
            }
        }
    }
    d.template combine<LEV>(plane);
}

