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
// per lane words 0 .. 4 = mask, moves of slice word 0, 5 .. 9 = of slice word 1, 10 .. 13 = the four places, 14 .. 15 = per
// slice word the low-bit masks of its two compressed pieces (as many ones as the halfword has information positions).
std::vector<uint32_t> build_compress_tables(const uint8_t *mask, int N);


// ---- kernel dispatch table ----------------------------------------------------------------------------------------------
// Which kernel family serves a call is decided by ONE table (dispatch_table), read in order: the first row whose
// conditions hold wins, and a call no row covers is refused (POLAR_EUNSUPPORTED).  The launchers of every decoder entry
// point, polar_dispatch_select / polar_dispatch_table of the C ABI and tests/test_dispatch.py all read this table.
enum { DQ_OP_SC = 0, DQ_OP_SCL = 1, DQ_OP_PBF = 2, DQ_OP_MC_SC = 3 };
enum {  // outputs a call asks for
    DQ_OUT_INFO = 1,     // information bits of the decoded / chosen word
    DQ_OUT_UHAT = 2,     // u_hat (SC: the word; SCL: every path of the list)
    DQ_OUT_DECLLR = 4,   // decision LLRs (SC)
    DQ_OUT_PM = 8        // sorted path metrics (SCL)
};
enum {  // properties of the call and of the code
    DQ_FAST = 1,            // schedule with the rate-0 / rate-1 short-cuts (POLAR_SC_FAST)
    DQ_SPECIALISED = 2,     // polar_code_specialize has built this code's own decoder
    DQ_CRC = 4,             // CRC-aided choice / CRC-driven flip search
    DQ_GENIE = 8,           // genie-aided choice
    DQ_FLIP_LIST = 16,      // PBF: the caller gives the flip positions (polar_sc_decode_flip)
    DQ_SPLIT = 32,          // PBF: the workspace of the two-flip search in pieces is at hand
    DQ_FORCE_LDS = 64,      // experiment switch POLAR_*_KERNEL=lds
    DQ_FORCE_GENERIC = 128, // experiment switch POLAR_SC_KERNEL=generic / POLAR_MC_KERNEL=staged
    DQ_INTERIOR_K = 256     // set by dispatch_select itself: 0 < K < N
};
enum KernelFamily {
    KF_SC_STATIC, KF_SC_REG, KF_SC_REG_FULL, KF_SC_BIG, KF_SC_LDS, KF_SCL_REG, KF_SCL_WIDE, KF_SCL_LDS, KF_PBF_REG, KF_PBF_LDS,
    KF_MC_FUSED, KF_MC_STAGED
};
struct DispatchQuery {
    int op;       // DQ_OP_*
    int n;        // log2 N
    int K;
    int dtype;    // 0 = fp32, 1 = fp64 (POLAR_F32 / POLAR_F64)
    int L;        // list size (SCL), largest number of simultaneous flips (PBF), else 0
    int outputs;  // DQ_OUT_*
    int flags;    // DQ_*
};
struct DispatchRow {
    int op;
    int family;
    const char *kernel;     // kernel name as polar_last_launch reports it (prefix)
    unsigned dtypes;        // bit 0: fp32, bit 1: fp64
    int n_min, n_max;       // log2 N, inclusive
    unsigned l_mask;        // bit L - 1 set = that L (1 .. 32) is served (0 = L is not looked at)
    int k_min;              // smallest K served
    unsigned outputs_ok;    // outputs the family can give
    unsigned outputs_any;   // if non-zero: at least one of these must be asked for
    unsigned need, forbid;  // flags that must / must not be set
    const char *what;
};
const DispatchRow *dispatch_table(int *count);
// index of the row that serves q, or -1
int dispatch_select(const DispatchQuery &q);
