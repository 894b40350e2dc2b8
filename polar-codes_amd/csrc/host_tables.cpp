// Host-only parts of libpolar_hip.so: the decoder-schedule compiler, the critical sets of the flip decoders and the
// CRC remainder tables.  No HIP here, so that this file also builds with a plain C++ compiler under the sanitizers
// (tests/test_sanitizers.py).
#include <algorithm>

#include "host_tables.h"

// Depth-first walk of the code tree.  A node at `level` covers leaves [pos, pos + 2^level).
static void walk(const uint8_t *mask, int level, int pos, int plane, bool fast, bool sub16, std::vector<uint32_t> &ops)
{
    int w = 1 << level;
    int ninfo = 0;
    for (int i = 0; i < w; i++) ninfo += mask[pos + i] ? 1 : 0;
    if (fast && ninfo == 0) {
        ops.push_back(make_op(OP_R0, level, plane));
        return;
    }
    if (fast && ninfo == w && level > 0) {
        ops.push_back(make_op(OP_R1, level, plane));
        return;
    }
    if (level == 0) {
        ops.push_back(make_op(mask[pos] ? OP_LEAF_INFO : OP_LEAF_FROZEN, 0, plane, pos));
        return;
    }
    if (sub16 && level == 4) {
        uint32_t m16 = 0;
        for (int i = 0; i < 16; i++) m16 |= (uint32_t)(mask[pos + i] ? 1 : 0) << i;
        ops.push_back(make_op(OP_SUB16, level, plane, m16));
        return;
    }
    ops.push_back(make_op(OP_F, level, 0));
    walk(mask, level - 1, pos, 0, fast, sub16, ops);
    ops.push_back(make_op(OP_G, level, 0));
    walk(mask, level - 1, pos + w / 2, 1, fast, sub16, ops);
    ops.push_back(make_op(OP_COMBINE, level, plane));
}

std::vector<uint32_t> build_schedule(const uint8_t *mask, int N, bool fast, bool sub16)
{
    std::vector<uint32_t> ops;
    int n = 0;
    while ((1 << n) < N) n++;
    walk(mask, n, 0, 0, fast, sub16 && n >= 5, ops);
    ops.push_back(make_op(OP_END, 0, 0));
    return ops;
}

// First positions of the maximal subtrees whose leaves are all information bits with index > after.
static void critical_heads(const std::vector<int> &pre, int lo, int size, int after, std::vector<int32_t> &out)
{
    const int ninfo = pre[lo + size] - pre[lo];
    if (ninfo == 0) return;
    if (ninfo == size) {
        out.push_back(lo);
        return;
    }
    critical_heads(pre, lo, size / 2, after, out);
    critical_heads(pre, lo + size / 2, size / 2, after, out);
}
std::vector<int32_t> critical_set(const uint8_t *mask, int N, int after)
{
    std::vector<int> pre(N + 1, 0);
    for (int i = 0; i < N; i++) pre[i + 1] = pre[i] + ((mask[i] && i > after) ? 1 : 0);
    std::vector<int32_t> out;
    critical_heads(pre, 0, N, after, out);
    return out;
}


// R[p] = x^(p + w) mod g: what a one at distance p from the end of a word leaves in the remainder register of the
// MSB-first shift register after the word and w zeros went through (cn/functions.py:567-583).
// E[k] (k < K - w): CRC bits (bit j = j-th appended bit) contributed by message bit k; S[k] (k < K): remainder
// contributed by bit k of the K-bit word (message + CRC).
void build_crc_tables(uint64_t crc_poly, int w, int K, std::vector<uint32_t> &E, std::vector<uint32_t> &S)
{
    const uint64_t keep = (w >= 64) ? ~0ull : ((1ull << w) - 1ull);
    const uint64_t low = crc_poly & keep;
    std::vector<uint32_t> R(K);
    uint64_t r = low;
    for (int p = 0; p < K; p++) {
        R[p] = (uint32_t)r;
        const uint64_t top = (r >> (w - 1)) & 1ull;
        r = ((r << 1) & keep) ^ (top ? low : 0ull);
    }
    const int Km = K - w;
    E.assign(std::max(1, Km), 0);
    S.assign(K, 0);
    for (int k = 0; k < Km; k++) {
        uint32_t v = R[Km - 1 - k], t = 0;  // appended bit j is register bit w-1-j
        for (int j = 0; j < w; j++) t |= ((v >> (w - 1 - j)) & 1u) << j;
        E[k] = t;
    }
    for (int k = 0; k < K; k++) S[k] = R[K - 1 - k];
}

std::vector<uint32_t> build_compress_tables(const uint8_t *mask, int N)
{
    std::vector<uint32_t> tab(256, 0u);
    if (N < 32 || N > 1024) return tab;
    std::vector<int> rank(N + 1, 0);  // information positions before position p
    for (int p = 0; p < N; p++) rank[p + 1] = rank[p] + (mask[p] ? 1 : 0);
    for (int l = 0; l < 16; l++) {
        uint32_t per_lane[16] = {0};
        for (int q = 0; q < 4; q++) {
            const int p0 = 16 * (l + 16 * q);
            uint32_t m = 0;
            if (p0 < N)
                for (int s = 0; s < 16; s++) m |= (uint32_t)(mask[p0 + s] ? 1 : 0) << s;
            // compress masks of the 16-bit mask m (the move of step i shifts by 2^i)
            uint32_t mv[4], mk = (~m << 1) & 0xFFFFu, mm = m;
            for (int i = 0; i < 4; i++) {
                uint32_t mp = mk ^ (mk << 1);
                mp ^= mp << 2;
                mp ^= mp << 4;
                mp ^= mp << 8;
                mp &= 0xFFFFu;
                mv[i] = mp & mm;
                mm = ((mm ^ mv[i]) | (mv[i] >> (1 << i))) & 0xFFFFu;
                mk &= ~mp;
            }
            const int w = q >> 1, sh = 16 * (q & 1);
            per_lane[5 * w] |= m << sh;
            for (int i = 0; i < 4; i++) per_lane[5 * w + 1 + i] |= mv[i] << sh;
            const int off = p0 < N ? rank[p0] : 0;
            per_lane[10 + q] = (uint32_t)(4 * (off >> 5)) | ((uint32_t)(off & 31) << 16);
            // the compressed piece: as many low bits as the halfword has information positions (the encoder side of the
            // Monte-Carlo kernel cuts a piece out of the information words with it before expanding)
            const int cnt = __builtin_popcount(m);
            per_lane[14 + w] |= (cnt >= 16 ? 0xFFFFu : ((1u << cnt) - 1u)) << sh;
        }
        for (int j = 0; j < 16; j++) tab[(size_t)(16 * (j >> 2) + l) * 4 + (j & 3)] = per_lane[j];
    }
    return tab;
}


// ---- kernel dispatch table ----------------------------------------------------------------------------------------------
namespace {
constexpr unsigned F32 = 1u, F64 = 2u;
// list sizes / flip counts served, as a mask over 1 .. 32 (bit L - 1)
constexpr unsigned LM(int a, int b = 0, int c = 0, int d = 0, int e = 0, int f = 0)
{
    return (a ? 1u << (a - 1) : 0u) | (b ? 1u << (b - 1) : 0u) | (c ? 1u << (c - 1) : 0u) | (d ? 1u << (d - 1) : 0u) |
           (e ? 1u << (e - 1) : 0u) | (f ? 1u << (f - 1) : 0u);
}
constexpr unsigned ALL_SC = DQ_OUT_INFO | DQ_OUT_UHAT | DQ_OUT_DECLLR, ALL_SCL = DQ_OUT_INFO | DQ_OUT_UHAT | DQ_OUT_PM;
const DispatchRow TABLE[] = {
    // ---- SC (polar_sc_decode*, polar_sc_decode_count, the SC passes of polar_mc_run's staged pipeline)
    {DQ_OP_SC, KF_SC_STATIC, "sc_static_kernel", F32, 5, 10, 0, 1, DQ_OUT_INFO, 0, DQ_FAST | DQ_SPECIALISED | DQ_INTERIOR_K, DQ_FORCE_LDS | DQ_FORCE_GENERIC,
     "register-resident, compiled per code at run time"},
    {DQ_OP_SC, KF_SC_REG, "sc_reg_kernel", F32, 5, 10, 0, 1, DQ_OUT_INFO, 0, DQ_INTERIOR_K, DQ_FORCE_LDS, "register-resident, any code (both schedules)"},
    {DQ_OP_SC, KF_SC_REG_FULL, "sc_reg_full_kernel<double", F64, 5, 10, 0, 0, ALL_SC, 0, 0, DQ_FORCE_LDS,
     "register-resident, every leaf, fp64 (the reference's arithmetic), u_hat and decision LLRs"},
    {DQ_OP_SC, KF_SC_REG_FULL, "sc_reg_full_kernel<float", F32, 5, 10, 0, 0, ALL_SC, DQ_OUT_UHAT | DQ_OUT_DECLLR, 0, DQ_FORCE_LDS | DQ_FAST,
     "register-resident, every leaf, fp32 with u_hat / decision LLRs"},
    {DQ_OP_SC, KF_SC_BIG, "sc_big_kernel", F32, 11, 16, 0, 1, DQ_OUT_INFO, 0, DQ_FAST | DQ_INTERIOR_K, DQ_FORCE_LDS,
     "long codes: wide levels in a global slab, 1024-leaf subtrees in registers"},
    {DQ_OP_SC, KF_SC_LDS, "sc_decode_kernel<float", F32, 0, 16, 0, 0, ALL_SC, 0, 0, 0, "LDS-staged, any code, every output"},
    {DQ_OP_SC, KF_SC_LDS, "sc_decode_kernel<double", F64, 0, 15, 0, 0, ALL_SC, 0, 0, 0, "LDS-staged, fp64"},
    // ---- SCL (polar_scl_decode*, genie and CRC variants, the list passes of polar_mc_run)
    {DQ_OP_SCL, KF_SCL_REG, "scl_reg_kernel<", F32, 6, 10, LM(2, 4, 8), 3, ALL_SCL, 0, 0, DQ_FORCE_LDS, "register-resident, eight lanes per path"},
    {DQ_OP_SCL, KF_SCL_REG, "scl_reg_kernel<", F64, 6, 10, LM(2, 4, 8), 3, ALL_SCL, 0, 0, DQ_FORCE_LDS,
     "the same in fp64 with the sorted list and its metrics"},
    {DQ_OP_SCL, KF_SCL_WIDE, "scl_wide_kernel<", F32, 6, 10, LM(16, 32), 5, ALL_SCL, 0, 0, DQ_FORCE_LDS, "register-resident, two lanes per path"},
    {DQ_OP_SCL, KF_SCL_WIDE, "scl_wide_kernel<", F64, 6, 10, LM(16, 32), 5, ALL_SCL, 0, 0, DQ_FORCE_LDS,
     "the same in fp64 with the sorted list and its metrics"},
    {DQ_OP_SCL, KF_SCL_WIDE, "scl_wide_kernel<", F32, 11, 12, LM(2, 4, 8, 16, 32), 5, ALL_SCL, 0, 0, DQ_FORCE_LDS,
     "N = 2048, 4096: the two-lanes-per-path kernel also serves L = 2, 4, 8 (part of its lanes at work, still several times the LDS-staged rate)"},
    {DQ_OP_SCL, KF_SCL_WIDE, "scl_wide_kernel<", F64, 11, 12, LM(2, 4, 8, 16, 32), 5, ALL_SCL, 0, 0, DQ_FORCE_LDS, "the same in fp64 with the sorted list and its metrics"},
    {DQ_OP_SCL, KF_SCL_LDS, "scl_decode_kernel<", F32 | F64, 0, 12, LM(1, 2, 4, 8, 16, 32), 0, ALL_SCL, 0, 0, 0, "LDS-staged, every output"},
    // ---- progressive bit flipping (polar_pbf_decode*, polar_sc_decode_flip); L = largest number of simultaneous flips
    {DQ_OP_PBF, KF_PBF_REG, "pbf_reg_kernel<", F32 | F64, 5, 10, LM(1), 1, DQ_OUT_INFO, 0, DQ_CRC, DQ_FORCE_LDS | DQ_FLIP_LIST,
     "SC runs in registers, four flip candidates per wavefront, single flips"},
    {DQ_OP_PBF, KF_PBF_REG, "pbf_reg_kernel<", F32 | F64, 5, 10, LM(2, 3), 1, DQ_OUT_INFO, 0, DQ_CRC | DQ_SPLIT, DQ_FORCE_LDS | DQ_FLIP_LIST,
     "the same for the two- and three-flip searches in pieces"},
    {DQ_OP_PBF, KF_PBF_LDS, "pbf_decode_kernel<", F32 | F64, 0, 12, 0, 0, ALL_SC, 0, 0, 0,
     "one wavefront per frame, LDS-staged SC runs: N > 1024, flip lists, side outputs, searches not worked on in pieces"},
    // ---- the SC Monte-Carlo loop of polar_mc_run
    {DQ_OP_MC_SC, KF_MC_FUSED, "sc_static_mc_kernel", F32, 7, 10, 0, 1, 0, 0, DQ_FAST | DQ_SPECIALISED | DQ_INTERIOR_K, DQ_CRC | DQ_FORCE_GENERIC,
     "draw, encode, channel, decode and count in one kernel, no LLR buffer"},
    {DQ_OP_MC_SC, KF_MC_STAGED, "gen_frames_kernel + SC decoder", F32 | F64, 0, 16, 0, 0, 0, 0, 0, 0, "generator -> LLR buffer -> decoder, two streams"},
};
}  // namespace

const DispatchRow *dispatch_table(int *count)
{
    if (count) *count = (int)(sizeof(TABLE) / sizeof(TABLE[0]));
    return TABLE;
}

int dispatch_select(const DispatchQuery &q)
{
    const int N = q.n >= 0 && q.n < 31 ? 1 << q.n : 0;
    const unsigned flags = ((unsigned)q.flags & ~(unsigned)DQ_INTERIOR_K) | ((q.K > 0 && q.K < N) ? (unsigned)DQ_INTERIOR_K : 0u);
    const unsigned dt = q.dtype == 0 ? F32 : q.dtype == 1 ? F64 : 0u;
    const int count = (int)(sizeof(TABLE) / sizeof(TABLE[0]));
    for (int i = 0; i < count; i++) {
        const DispatchRow &r = TABLE[i];
        if (r.op != q.op || !(r.dtypes & dt)) continue;
        if (q.n < r.n_min || q.n > r.n_max || q.K < r.k_min) continue;
        if (r.l_mask && (q.L < 1 || q.L > 32 || !((r.l_mask >> (q.L - 1)) & 1u))) continue;
        if ((unsigned)q.outputs & ~r.outputs_ok) continue;
        if (r.outputs_any && !((unsigned)q.outputs & r.outputs_any)) continue;
        if ((flags & r.need) != r.need || (flags & r.forbid)) continue;
        return i;
    }
    return -1;
}
