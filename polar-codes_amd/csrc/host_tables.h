// Host-only declarations: the decoder schedule encoding and the table builders of host_tables.cpp (no HIP needed).
#pragma once
// (see build_compress_tables below for the table of the register SC kernels' output stage)
#include <stdint.h>

#include <vector>

// ---- decoder schedule -------------------------------------------------------------------------
// One 32-bit word per operation; the list is the depth-first walk of the code tree.
//   bits 0..3  type, bits 4..8 level (log2 of the width of the node the op belongs to),
//   bit 9 destination plane of the node's partial sums (0 = node is a left child, 1 = right child),
//   bits 12..31 leaf index (LEAF ops).
enum : uint32_t {
    OP_F = 0,        // level-(l-1) LLRs of the left child from the level-l LLRs (min-sum)
    OP_G = 1,        // level-(l-1) LLRs of the right child, using the left child's partial sums
    OP_LEAF_INFO = 2,
    OP_LEAF_FROZEN = 3,
    OP_R0 = 4,       // whole node frozen: partial sums are zero
    OP_R1 = 5,       // whole node information: partial sums = hard decisions of its LLRs
    OP_COMBINE = 6,  // node partial sums (xl ^ xr, xr) from its two children
    OP_SUB16 = 7,    // a whole 16-leaf node decoded in registers; bits 12..27 = its rate profile (leaf i = bit i)
    OP_END = 15
};
static inline uint32_t make_op(uint32_t type, uint32_t level, uint32_t plane, uint32_t leaf = 0)
{
    return type | (level << 4) | (plane << 9) | (leaf << 12);
}
#define OP_TYPE(w) ((w)&15u)
#define OP_LEVEL(w) (((w) >> 4) & 31u)
#define OP_PLANE(w) (((w) >> 9) & 1u)
#define OP_LEAFIDX(w) ((w) >> 12)

// sub16: 16-leaf nodes become one OP_SUB16 each instead of their 61 leaf-level operations
std::vector<uint32_t> build_schedule(const uint8_t *mask, int N, bool fast, bool sub16 = false);


// first positions of the maximal all-information subtrees of `mask` restricted to indices > after (after = -1: all)
std::vector<int32_t> critical_set(const uint8_t *mask, int N, int after);
// CRC remainder maps of a K-bit information word whose last w bits are the CRC of generator crc_poly (leading term included)
void build_crc_tables(uint64_t crc_poly, int w, int K, std::vector<uint32_t> &E, std::vector<uint32_t> &S);

// Table of the register SC kernels' output stage (N = 32 .. 1024), 256 words.  After the 16 x 16 bit transpose across
// the lanes of a frame's row, lane l holds the u_hat bits of code positions 16 (l + 16 q) .. + 15 in halfword q of its
// slice (q = 0 .. 3, two halfwords per 32-bit word).  Per lane and word: the information mask of the two halfwords and
// the four move masks that compress each halfword's information bits towards its bit 0 (parallel suffix method, 16-bit
// halves handled independently); per lane and halfword: where the compressed piece goes in the frame's information
// words, as (byte offset of the word) | (bit offset << 16).  Layout: word (4 g + i) of lane l at [(16 g + l) * 4 + i];
// per lane words 0 .. 4 = mask, moves of slice word 0, 5 .. 9 = of slice word 1, 10 .. 13 = the four places.
std::vector<uint32_t> build_compress_tables(const uint8_t *mask, int N);
