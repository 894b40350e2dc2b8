/*
 * polar_oracle.c -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker for the HIP library: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py are the only things allowed to load it.  The product path
 * never links or calls it.
 *
 * Every function restates one piece of the reference (Tigrexes/polar-codes, files under cn/)
 * in plain C, keeping the reference's data layout (the 2N-1 "heap" LLR array and the two
 * partial-sum planes) and its order of floating-point operations so that results are
 * bit-identical for fp64.  Parity is pinned: tests/test_oracle_golden.py checks these
 * functions against vectors produced by running the reference itself (tools/gen_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static int ilog2(int n) { int k = 0; while ((1 << k) < n) k++; return k; }

/* ------------------------------------------------------------------ encoder
 * cn/polarCode.py:225-230 (rateProfile): u = 0; u[mask==1] = info, increasing index order. */
ORC_API void orc_rate_profile(const uint8_t *info, const uint8_t *mask, int N, uint8_t *u)
{
    int k = 0;
    for (int i = 0; i < N; i++) u[i] = mask[i] ? info[k++] : 0;
}

/* cn/polarCode.py:232-248 (mul_matrix): in-place butterfly, stage i pairs (j, j+2^i) for every j
 * whose bit i is clear; natural order, no bit reversal. */
ORC_API void orc_mul_matrix(uint8_t *x, int N)
{
    int n = ilog2(N);
    for (int i = 0; i < n; i++) {
        int d = 1 << i;
        for (int j = 0; j < N; j++)
            if (!(j & d)) x[j] ^= x[j + d];
    }
}

/* cn/polarCode.py:250-255 (polarEncode). */
ORC_API void orc_polar_encode(const uint8_t *info, const uint8_t *mask, int N, uint8_t *x)
{
    orc_rate_profile(info, mask, N, x);
    orc_mul_matrix(x, N);
}

/* cn/polarCode.py:305-312 (extract). */
ORC_API int orc_extract(const uint8_t *uhat, const uint8_t *mask, int N, uint8_t *info)
{
    int k = 0;
    for (int i = 0; i < N; i++) if (mask[i]) info[k++] = uhat[i];
    return k;
}

/* cn/functions.py:107-109 (fails). */
ORC_API int64_t orc_fails(const uint8_t *a, const uint8_t *b, int K)
{
    int64_t e = 0;
    for (int i = 0; i < K; i++) e += (a[i] != b[i]);
    return e;
}

/* ------------------------------------------------------------------ per-path decoder state
 * cn/polarCode.py:14-28 (class Path).  LLRs[0] is the decision LLR; "level lev" (1..n) outputs sit at
 * [2^(lev-1)-1, 2^lev-2] and read their 2^lev inputs from [2^lev-1, 2^(lev+1)-2]; the channel LLRs
 * sit at [N-1, 2N-2].  BITs is two planes of N-1 partial sums with the same indexing. */
#define DEFINE_PATH(REAL, SUF)                                                                        \
    typedef struct {                                                                                  \
        REAL *llr;      /* 2N-1 */                                                                    \
        uint8_t *bit0;  /* N-1  */                                                                    \
        uint8_t *bit1;  /* N-1  */                                                                    \
        uint8_t *uhat;  /* N    */                                                                    \
        double pm;                                                                                    \
    } path_##SUF;                                                                                     \
                                                                                                      \
    static void path_alloc_##SUF(path_##SUF *p, int N)                                                \
    {                                                                                                 \
        p->llr = (REAL *)calloc(2 * (size_t)N, sizeof(REAL));                                         \
        p->bit0 = (uint8_t *)calloc(N, 1);                                                            \
        p->bit1 = (uint8_t *)calloc(N, 1);                                                            \
        p->uhat = (uint8_t *)calloc(N, 1);                                                            \
        p->pm = 0.0;                                                                                  \
    }                                                                                                 \
    static void path_free_##SUF(path_##SUF *p) { free(p->llr); free(p->bit0); free(p->bit1); free(p->uhat); } \
    static void path_copy_##SUF(path_##SUF *d, const path_##SUF *s, int N)                            \
    {                                                                                                 \
        memcpy(d->llr, s->llr, (2 * (size_t)N - 1) * sizeof(REAL));                                   \
        memcpy(d->bit0, s->bit0, N - 1);                                                              \
        memcpy(d->bit1, s->bit1, N - 1);                                                              \
        memcpy(d->uhat, s->uhat, N);                                                                  \
        d->pm = s->pm;                                                                                \
    }                                                                                                 \
                                                                                                      \
    /* cn/polarCode.py:30-58 (Path.updateLLRs).  The reference passes bitreversed(i); the level it    \
     * derives from the first '1' of that string is ctz(i)+1, used directly here.  One g level        \
     * (cn/functions.py:115-119 lowerconv) then min-sum f levels down to index 0. */                  \
    static int64_t update_llrs_##SUF(path_##SUF *p, int i, int n)                                     \
    {                                                                                                 \
        int64_t unit = 0;                                                                             \
        int next;                                                                                     \
        REAL *L = p->llr;                                                                             \
        if (i == 0) {                                                                                 \
            next = n;                                                                                 \
        } else {                                                                                      \
            int last = __builtin_ctz(i) + 1;                                                          \
            int start = (1 << (last - 1)) - 1, end = (1 << last) - 2, half = 1 << (last - 1);         \
            for (int k = start; k <= end; k++) {                                                      \
                REAL upper = L[end + (k - start) + 1];                                                \
                REAL lower = L[end + (k - start) + 1 + half];                                         \
                L[k] = p->bit0[k] == 0 ? lower + upper : lower - upper;                               \
                unit++;                                                                               \
            }                                                                                         \
            next = last - 1;                                                                          \
        }                                                                                             \
        for (int lev = next; lev >= 1; lev--) {                                                       \
            int start = (1 << (lev - 1)) - 1, end = (1 << lev) - 2, half = 1 << (lev - 1);            \
            for (int k = start; k <= end; k++) {                                                      \
                REAL a = L[end + (k - start) + 1];                                                    \
                REAL b = L[end + (k - start) + 1 + half];                                             \
                REAL sa = (REAL)((a > 0) - (a < 0)), sb = (REAL)((b > 0) - (b < 0));                  \
                REAL aa = a < 0 ? -a : a, ab = b < 0 ? -b : b;                                        \
                L[k] = sa * sb * (aa < ab ? aa : ab);                                                 \
                unit++;                                                                               \
            }                                                                                         \
        }                                                                                             \
        return unit;                                                                                  \
    }                                                                                                 \
                                                                                                      \
    /* cn/polarCode.py:61-85 (Path.updateBits): i even -> plane 0 of level 1; i odd -> climb          \
     * cto(i) levels combining (a^b, b); the last combine lands in plane 0. */                        \
    static void update_bits_##SUF(path_##SUF *p, int i, int N)                                        \
    {                                                                                                 \
        uint8_t b = p->uhat[i];                                                                       \
        if (i == N - 1) return;                                                                       \
        if (!(i & 1)) { p->bit0[0] = b; return; }                                                     \
        int t = __builtin_ctz(~(unsigned)i); /* trailing ones */                                      \
        p->bit1[0] = b;                                                                               \
        for (int lev = 1; lev <= t; lev++) {                                                          \
            int st = (1 << (lev - 1)) - 1, ed = (1 << lev) - 2, half = 1 << (lev - 1);                \
            uint8_t *dst = (lev == t) ? p->bit0 : p->bit1;                                            \
            for (int k = st; k <= ed; k++) {                                                          \
                dst[ed + (k - st) + 1] = p->bit0[k] ^ p->bit1[k];                                     \
                dst[ed + (k - st) + 1 + half] = p->bit1[k];                                           \
            }                                                                                         \
        }                                                                                             \
    }

DEFINE_PATH(double, f64)
DEFINE_PATH(float, f32)

/* ------------------------------------------------------------------ SC
 * cn/polarCode.py:1525-1571 (scDecoder).  Decision: info bit -> (LLR < 0); an exactly zero LLR leaves
 * the bit at 0 (the reference blocks on input() there and then continues with 0). Frozen -> 0.
 * flips: optional sorted list of bit indices whose decision is inverted (the sequenceU hook). */
#define DEFINE_SC(REAL, SUF)                                                                          \
    ORC_API int orc_sc_decode_##SUF(const REAL *llr, const uint8_t *mask, int N, uint8_t *uhat,       \
                                    REAL *decllr, const int32_t *flips, int nflips)                   \
    {                                                                                                 \
        int n = ilog2(N);                                                                             \
        path_##SUF p;                                                                                 \
        path_alloc_##SUF(&p, N);                                                                      \
        memcpy(p.llr + (N - 1), llr, (size_t)N * sizeof(REAL));                                       \
        for (int i = 0; i < N; i++) {                                                                 \
            update_llrs_##SUF(&p, i, n);                                                              \
            REAL d = p.llr[0];                                                                        \
            if (decllr) decllr[i] = d;                                                                \
            p.uhat[i] = (mask[i] && d < 0) ? 1 : 0;                                                   \
            for (int f = 0; f < nflips; f++) if (flips[f] == i) p.uhat[i] ^= 1;                       \
            update_bits_##SUF(&p, i, N);                                                              \
        }                                                                                             \
        memcpy(uhat, p.uhat, N);                                                                      \
        path_free_##SUF(&p);                                                                          \
        return 0;                                                                                     \
    }

DEFINE_SC(double, f64)
DEFINE_SC(float, f32)

/* ------------------------------------------------------------------ SCL
 * Branch metric, cn/polarCode.py:982-988 and :2066-2067: Li = LLR/ln2; bm(u) = 1 - log2(1 + 2^(-Li*(-1)^u)).
 * Evaluated in this order in fp64; 2^x overflows to +inf for x >= 1024, giving bm = -inf (never NaN). */
static double branch_metric(double llr, int u)
{
    double Li = llr / log(2.0);
    double e = u ? Li : -Li;
    return 1.0 - log2(1.0 + pow(2.0, e));
}

/* The fp32 throughput mode of the HIP library evaluates the same quantity in the overflow-free form
 * max(e,0) + log2(1 + 2^-|e|) and only reproduces the reference's -inf for e >= 1024. */
static double branch_metric_f32(float llr, int u)
{
    float Li = llr * 1.44269504088896340736f;
    float e = u ? Li : -Li;
    if (e >= 1024.0f) return -INFINITY;
    float m = e > 0 ? e : 0.0f;
    float r = 1.0f - (m + log2f(1.0f + exp2f(-fabsf(e))));
    return (double)r;
}

typedef struct { double pm; int pos; } cand_key;

/* Sensitivity of the last orc_scl_decode_* call of this thread to small errors of the metrics: the smallest gap
 * between metrics that are neighbours in a sorted candidate list (every prune, and the final sort), relative to
 * max(|metric|, 1); pairs of -inf are ordered by position alone and do not count.  A decoder whose metrics differ
 * from the oracle's by less than half of this makes the same choices. */
static __thread double orc_last_gap = 0.0;
ORC_API double orc_scl_last_min_gap(void) { return orc_last_gap; }
static void note_gaps(const double *sorted_desc, int n)
{
    for (int c = 0; c + 1 < n; c++) {
        double a = sorted_desc[c], b = sorted_desc[c + 1];
        if (isinf(a) || isinf(b)) continue;
        double sc = fabs(a) > 1.0 ? fabs(a) : 1.0, sb = fabs(b) > 1.0 ? fabs(b) : 1.0;
        double g = (a - b) / (sc > sb ? sc : sb);
        if (g < orc_last_gap) orc_last_gap = g;
    }
}

/* cn/polarCode.py:2032-2102 (sclDecoder) with :969-1005 (polarSCLFork).
 * List order is kept exactly: below saturation list = originals ++ copies; at saturation the
 * candidates are copies(u=1) ++ originals(u=0), stable-sorted by metric descending, first L kept in
 * sorted order.  Final list is stable-sorted descending.  Outputs the whole sorted list.
 * counts[0..2] = ANV, sortNum, unitCal as the reference accumulates them. */
#define DEFINE_SCL(REAL, SUF, BM, PMR)                                                                    \
    ORC_API int orc_scl_decode_##SUF(const REAL *llr, const uint8_t *mask, int N, int L,              \
                                     uint8_t *uhat_all, double *pm_all, int64_t *counts)              \
    {                                                                                                 \
        int n = ilog2(N);                                                                             \
        path_##SUF *list = (path_##SUF *)malloc(sizeof(path_##SUF) * 2 * L);                          \
        path_##SUF *tmp = (path_##SUF *)malloc(sizeof(path_##SUF) * 2 * L);                           \
        for (int j = 0; j < 2 * L; j++) path_alloc_##SUF(&list[j], N);                                \
        int cur = 1;                                                                                  \
        int64_t anv = 0, sortnum = 0, unit = 0;                                                       \
        orc_last_gap = INFINITY;                                                                      \
        memcpy(list[0].llr + (N - 1), llr, (size_t)N * sizeof(REAL));                                 \
        for (int i = 0; i < N; i++) {                                                                 \
            for (int j = 0; j < cur; j++) { unit += update_llrs_##SUF(&list[j], i, n); anv++; }       \
            if (mask[i]) {                                                                            \
                /* fork every path: original takes u=0, copy takes u=1 */                             \
                for (int j = 0; j < cur; j++) {                                                       \
                    path_copy_##SUF(&list[cur + j], &list[j], N);                                     \
                    list[j].uhat[i] = 0;                                                              \
                    list[j].pm = PMR(list[j].pm + BM(list[j].llr[0], 0));                                            \
                    list[cur + j].uhat[i] = 1;                                                        \
                    list[cur + j].pm = PMR(list[cur + j].pm + BM(list[cur + j].llr[0], 1));                                \
                }                                                                                     \
                if (2 * cur <= L) {                                                                   \
                    cur *= 2; /* originals ++ copies, already laid out that way */                    \
                } else {                                                                              \
                    /* candidates C = copies ++ originals; stable sort by pm descending */            \
                    int nc = 2 * cur;                                                                 \
                    int order[128];                                                                   \
                    for (int c = 0; c < nc; c++) order[c] = (c < cur) ? cur + c : c - cur;            \
                    for (int a = 1; a < nc; a++) { /* insertion sort is stable */                     \
                        int v = order[a];                                                             \
                        int b = a - 1;                                                                \
                        while (b >= 0 && list[order[b]].pm < list[v].pm) { order[b + 1] = order[b]; b--; } \
                        order[b + 1] = v;                                                             \
                    }                                                                                 \
                    sortnum++;                                                                        \
                    for (int c = 0; c < nc; c++) tmp[c] = list[order[c]];                             \
                    for (int c = 0; c < nc; c++) list[c] = tmp[c];                                    \
                    { double sp[128]; for (int c = 0; c < nc; c++) sp[c] = list[c].pm; note_gaps(sp, nc); } \
                    cur = L;                                                                          \
                }                                                                                     \
            } else {                                                                                  \
                for (int j = 0; j < cur; j++) {                                                       \
                    list[j].uhat[i] = 0;                                                              \
                    list[j].pm = PMR(list[j].pm + BM(list[j].llr[0], 0));                                            \
                }                                                                                     \
            }                                                                                         \
            for (int j = 0; j < cur; j++) update_bits_##SUF(&list[j], i, N);                          \
        }                                                                                             \
        {                                                                                             \
            int order[128];                                                                           \
            for (int c = 0; c < cur; c++) order[c] = c;                                               \
            for (int a = 1; a < cur; a++) {                                                           \
                int v = order[a];                                                                     \
                int b = a - 1;                                                                        \
                while (b >= 0 && list[order[b]].pm < list[v].pm) { order[b + 1] = order[b]; b--; }    \
                order[b + 1] = v;                                                                     \
            }                                                                                         \
            sortnum++;                                                                                \
            for (int c = 0; c < cur; c++) {                                                           \
                memcpy(uhat_all + (size_t)c * N, list[order[c]].uhat, N);                             \
                pm_all[c] = list[order[c]].pm;                                                        \
            }                                                                                         \
            note_gaps(pm_all, cur);                                                                                         \
        }                                                                                             \
        if (counts) { counts[0] = anv; counts[1] = sortnum; counts[2] = unit; }                       \
        for (int j = 0; j < 2 * L; j++) path_free_##SUF(&list[j]);                                    \
        free(list);                                                                                   \
        free(tmp);                                                                                    \
        return cur;                                                                                   \
    }

#define PM_KEEP(x) (x)
#define PM_F32(x) ((double)(float)(x))
DEFINE_SCL(double, f64, branch_metric, PM_KEEP)
DEFINE_SCL(float, f32, branch_metric_f32, PM_F32)

/* cn/polarCode.py:1267-1317 (sclDecoderOracle, the decoder cn/simulator.py:156 calls) with :1041-1097
 * (polarSCLFork2).  Same list decoder, but paths stay in their slots: at saturation the 2n candidate metrics
 * are laid out [u=0 of slot 0..n-1, u=1 of slot 0..n-1] and stable-sorted descending (equal metrics: lower
 * index first); a slot whose two continuations both survive keeps u=0 and copies itself, with u=1, into the
 * lowest-numbered slot that lost both (processed in slot order).  The final list is stable-sorted descending
 * from slot order.  Outputs the whole sorted list; the genie's choice is orc_genie_select on it. */
#define DEFINE_SCL2(REAL, SUF, BM, PMR)                                                                   \
    ORC_API int orc_scl2_decode_##SUF(const REAL *llr, const uint8_t *mask, int N, int L,             \
                                      uint8_t *uhat_all, double *pm_all, int64_t *counts)             \
    {                                                                                                 \
        int n = ilog2(N);                                                                             \
        path_##SUF *list = (path_##SUF *)malloc(sizeof(path_##SUF) * L);                              \
        for (int j = 0; j < L; j++) path_alloc_##SUF(&list[j], N);                                    \
        int cur = 1;                                                                                  \
        int64_t anv = 0, sortnum = 0, unit = 0;                                                       \
        memcpy(list[0].llr + (N - 1), llr, (size_t)N * sizeof(REAL));                                 \
        for (int i = 0; i < N; i++) {                                                                 \
            for (int j = 0; j < cur; j++) { unit += update_llrs_##SUF(&list[j], i, n); anv++; }       \
            if (mask[i] && 2 * cur <= L) {                                                            \
                for (int j = 0; j < cur; j++) {                                                       \
                    path_copy_##SUF(&list[cur + j], &list[j], N);                                     \
                    double d = list[j].llr[0];                                                        \
                    list[j].pm = PMR(list[j].pm + BM(d, 0));                                          \
                    list[cur + j].uhat[i] = 1;                                                        \
                    list[cur + j].pm = PMR(list[cur + j].pm + BM(d, 1));                              \
                }                                                                                     \
                cur *= 2;                                                                             \
            } else if (mask[i]) {                                                                     \
                double PM[128];                                                                       \
                int order[128], discard[128], state[64], lost[64], nlost = 0, next_lost = 0;          \
                for (int j = 0; j < cur; j++) {                                                       \
                    double d = list[j].llr[0];                                                        \
                    PM[j] = PMR(list[j].pm + BM(d, 0));                                               \
                    PM[j + cur] = PMR(list[j].pm + BM(d, 1));                                         \
                }                                                                                     \
                sortnum++;                                                                            \
                for (int c = 0; c < 2 * cur; c++) { order[c] = c; discard[c] = 1; }                   \
                for (int a = 1; a < 2 * cur; a++) {                                                   \
                    int v = order[a], b = a - 1;                                                      \
                    while (b >= 0 && PM[order[b]] < PM[v]) { order[b + 1] = order[b]; b--; }          \
                    order[b + 1] = v;                                                                 \
                }                                                                                     \
                for (int c = 0; c < L; c++) discard[order[c]] = 0;                                    \
                for (int j = 0; j < cur; j++) {                                                       \
                    if (discard[j] && discard[j + cur]) { state[j] = 1; lost[nlost++] = j; }          \
                    else if (discard[j]) state[j] = 2;                                                \
                    else if (discard[j + cur]) state[j] = 3;                                          \
                    else state[j] = 4;                                                                \
                }                                                                                     \
                for (int j = 0; j < cur; j++) {                                                       \
                    if (state[j] == 2) { list[j].pm = PM[j + cur]; list[j].uhat[i] = 1; }             \
                    else if (state[j] == 3) { list[j].pm = PM[j]; }                                   \
                    else if (state[j] == 4) {                                                         \
                        int t = lost[next_lost++];                                                    \
                        path_copy_##SUF(&list[t], &list[j], N);                                       \
                        list[j].pm = PM[j];                                                           \
                        list[t].pm = PM[j + cur];                                                     \
                        list[t].uhat[i] = 1;                                                          \
                    }                                                                                 \
                }                                                                                     \
            } else {                                                                                  \
                for (int j = 0; j < cur; j++) {                                                       \
                    list[j].uhat[i] = 0;                                                              \
                    list[j].pm = PMR(list[j].pm + BM(list[j].llr[0], 0));                             \
                }                                                                                     \
            }                                                                                         \
            for (int j = 0; j < cur; j++) update_bits_##SUF(&list[j], i, N);                          \
        }                                                                                             \
        {                                                                                             \
            int order[128];                                                                           \
            for (int c = 0; c < cur; c++) order[c] = c;                                               \
            for (int a = 1; a < cur; a++) {                                                           \
                int v = order[a], b = a - 1;                                                          \
                while (b >= 0 && list[order[b]].pm < list[v].pm) { order[b + 1] = order[b]; b--; }    \
                order[b + 1] = v;                                                                     \
            }                                                                                         \
            sortnum++;                                                                                \
            for (int c = 0; c < cur; c++) {                                                           \
                memcpy(uhat_all + (size_t)c * N, list[order[c]].uhat, N);                             \
                pm_all[c] = list[order[c]].pm;                                                        \
            }                                                                                         \
        }                                                                                             \
        if (counts) { counts[0] = anv; counts[1] = sortnum; counts[2] = unit; }                       \
        for (int j = 0; j < L; j++) path_free_##SUF(&list[j]);                                        \
        free(list);                                                                                   \
        return cur;                                                                                   \
    }

DEFINE_SCL2(double, f64, branch_metric, PM_KEEP)
DEFINE_SCL2(float, f32, branch_metric_f32, PM_F32)

/* The genie of sclDecoderOracle (cn/polarCode.py:1314-1317): the first entry of the sorted list equal to the
 * transmitted u (self.message), the best entry if there is none.  Returns the row. */
ORC_API int orc_genie_select(const uint8_t *uhat_sorted, int nsurv, int N, const uint8_t *u_true)
{
    for (int j = 0; j < nsurv; j++)
        if (memcmp(uhat_sorted + (size_t)j * N, u_true, (size_t)N) == 0) return j;
    return 0;
}

/* ------------------------------------------------------------------ batched wrappers (cpu_baseline, tests) */
ORC_API void orc_sc_decode_batch_f32(const float *llr, const uint8_t *mask, int N, int64_t B, uint8_t *uhat)
{
    for (int64_t b = 0; b < B; b++) orc_sc_decode_f32(llr + b * N, mask, N, uhat + b * N, NULL, NULL, 0);
}
ORC_API void orc_sc_decode_batch_f64(const double *llr, const uint8_t *mask, int N, int64_t B, uint8_t *uhat)
{
    for (int64_t b = 0; b < B; b++) orc_sc_decode_f64(llr + b * N, mask, N, uhat + b * N, NULL, NULL, 0);
}

/* ------------------------------------------------------------------ synthetic-input generator
 * Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the HIP library uses for Monte-Carlo
 * inputs.  Not part of the reference (which draws from numpy's global MT19937, cn/simulator.py:138 and
 * cn/channel.py:35); restated here so the device stream can be checked word for word.
 *   key     = (seed_lo, seed_hi)
 *   counter = (frame_lo, frame_hi, block, stream)   stream 0 = info bits, 1 = channel noise
 *   info bit k of a frame = bit (k & 31) of word (k >> 5); word w = output (w & 3) of block (w >> 2)
 *   noise block j -> outputs r0..r3 -> Box-Muller pairs (r0,r1) -> positions 4j,4j+1; (r2,r3) -> 4j+2,4j+3
 *       u1 = (r + 0.5) * 2^-32 in (0,1], u2 = r' * 2^-32 in [0,1); z0 = R cos(2 pi u2), z1 = R sin(2 pi u2)
 *   channel (cn/channel.py:19,35,56): s = 1-2x; y = s + sqrt(N0/2) z; llr = (4/N0) y                        */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo)
{
    uint64_t p = (uint64_t)a * b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

ORC_API void orc_gen_info(uint64_t seed, uint64_t frame, int K, uint8_t *info)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int w4 = 0; w4 * 128 < K; w4++) {
        uint32_t ctr[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)w4, 0u}, r[4];
        orc_philox4x32_10(ctr, key, r);
        for (int b = 0; b < 128 && w4 * 128 + b < K; b++) info[w4 * 128 + b] = (r[b >> 5] >> (b & 31)) & 1u;
    }
}

ORC_API void orc_gen_noise(uint64_t seed, uint64_t frame, int N, double *z)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j * 4 < N; j++) {
        uint32_t ctr[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)j, 1u}, r[4];
        orc_philox4x32_10(ctr, key, r);
        for (int h = 0; h < 2; h++) {
            double u1 = ((double)r[2 * h] + 0.5) * (1.0 / 4294967296.0);
            double u2 = (double)r[2 * h + 1] * (1.0 / 4294967296.0);
            double R = sqrt(-2.0 * log(u1));
            if (4 * j + 2 * h < N) z[4 * j + 2 * h] = R * cos(two_pi * u2);
            if (4 * j + 2 * h + 1 < N) z[4 * j + 2 * h + 1] = R * sin(two_pi * u2);
        }
    }
}

/* One synthetic frame: info -> u -> x -> BPSK -> AWGN -> LLR (fp64). */
ORC_API void orc_gen_frame(uint64_t seed, uint64_t frame, const uint8_t *mask, int N, int K, double N0,
                           uint8_t *info, uint8_t *x, double *llr)
{
    double *z = (double *)malloc(sizeof(double) * N);
    orc_gen_info(seed, frame, K, info);
    orc_polar_encode(info, mask, N, x);
    orc_gen_noise(seed, frame, N, z);
    double sigma = sqrt(N0 / 2.0), gain = 4.0 / N0;
    for (int i = 0; i < N; i++) llr[i] = gain * ((double)(1 - 2 * (int)x[i]) + sigma * z[i]);
    free(z);
}

/* Monte-Carlo reference loop over [frame0, frame0+nframes): counters = frames, frame errors, bit errors.
 * decoder: 0 = SC, 1 = SCL(L); dtype: 0 = f32 LLRs, 1 = f64 LLRs. */
ORC_API void orc_mc_run(const uint8_t *mask, int N, int K, int decoder, int L, int dtype, double N0,
                        uint64_t seed, uint64_t frame0, int64_t nframes, uint64_t counters[3])
{
    uint8_t *info = malloc(K), *x = malloc(N), *uh = malloc((size_t)N * (L > 0 ? L : 1)), *dec = malloc(K);
    double *llr = malloc(sizeof(double) * N), *pm = malloc(sizeof(double) * (L > 0 ? L : 1));
    float *llrf = malloc(sizeof(float) * N);
    for (int64_t f = 0; f < nframes; f++) {
        orc_gen_frame(seed, frame0 + f, mask, N, K, N0, info, x, llr);
        if (dtype == 0) for (int i = 0; i < N; i++) llrf[i] = (float)llr[i];
        if (decoder == 0) {
            if (dtype == 0) orc_sc_decode_f32(llrf, mask, N, uh, NULL, NULL, 0);
            else orc_sc_decode_f64(llr, mask, N, uh, NULL, NULL, 0);
        } else {
            if (dtype == 0) orc_scl_decode_f32(llrf, mask, N, L, uh, pm, NULL);
            else orc_scl_decode_f64(llr, mask, N, L, uh, pm, NULL);
        }
        orc_extract(uh, mask, N, dec);
        int64_t e = orc_fails(info, dec, K);
        counters[0] += 1;
        counters[1] += (e > 0);
        counters[2] += (uint64_t)e;
    }
    free(info); free(x); free(uh); free(dec); free(llr); free(pm); free(llrf);
}

/* ------------------------------------------------------------------------------------------------
 * CRC-aided list decoding
 * ------------------------------------------------------------------------------------------------ */

/* pcf.crcEncode (cn/functions.py:567-583) with the bit-array register of the reference: the polynomial's
 * coefficients below the leading term (pcf.standardForm, cn/functions.py:83-92), MSB first; message of n bits
 * followed by w zeros is shifted in; out = message ++ register (n + w bits).  crc_poly carries the leading
 * term (0x107: w = 8). */
ORC_API int orc_crc_encode(const uint8_t *message, int n, uint64_t crc_poly, uint8_t *out)
{
    int w = 0;
    while ((crc_poly >> (w + 1)) != 0) w++;
    uint8_t poly[64], reg[64];
    for (int j = 0; j < w; j++) {
        poly[j] = (uint8_t)((crc_poly >> (w - 1 - j)) & 1u);
        reg[j] = 0;
    }
    for (int i = 0; i < n + w; i++) {
        uint8_t bit = i < n ? message[i] : 0;
        uint8_t top = reg[0];
        for (int j = 0; j + 1 < w; j++) reg[j] = reg[j + 1];
        reg[w - 1] = bit;
        if (top)
            for (int j = 0; j < w; j++) reg[j] ^= poly[j];
    }
    for (int i = 0; i < n; i++) out[i] = message[i];
    for (int j = 0; j < w; j++) out[n + j] = reg[j];
    return w;
}

/* The choice at the end of sclDecoder(isCRC=True) (cn/polarCode.py:2079-2097): the sorted list is scanned
 * for the first path whose extracted information word, run through crcEncode once more, leaves an all-zero
 * tail; without one the best path stays.  uhat_sorted: nsurv rows of N bits.  Returns the chosen row. */
ORC_API int orc_ca_select(const uint8_t *uhat_sorted, const uint8_t *mask, int N, int nsurv, uint64_t crc_poly)
{
    uint8_t *word = malloc((size_t)N), *enc = malloc((size_t)N + 64);
    int chosen = 0, found = 0;
    for (int j = 0; j < nsurv && !found; j++) {
        int K = orc_extract(uhat_sorted + (size_t)j * N, mask, N, word);
        int w = orc_crc_encode(word, K, crc_poly, enc);
        int sum = 0;
        for (int t = 0; t < w; t++) sum += enc[K + t];
        if (sum == 0) {
            chosen = j;
            found = 1;
        }
    }
    free(word);
    free(enc);
    return chosen;
}

/* CRC-aided Monte-Carlo loop: as orc_mc_run with the list decoder, but the first K - w information bits are
 * the message (the same Philox bits a CRC-less frame would start with), the CRC fills the rest
 * (CRCPolarEncode2, cn/polarCode.py:265-271), the decoder's choice is CRC-aided and only message bits count. */
ORC_API void orc_mc_run_crc(const uint8_t *mask, int N, int K, int L, int dtype, double N0, uint64_t crc_poly,
                            uint64_t seed, uint64_t frame0, int64_t nframes, uint64_t counters[3])
{
    int w = 0;
    while ((crc_poly >> (w + 1)) != 0) w++;
    const int Km = K - w;
    uint8_t *info = malloc(K + 64), *msg = malloc(K), *x = malloc(N), *uh = malloc((size_t)N * L), *dec = malloc(K);
    double *llr = malloc(sizeof(double) * N), *pm = malloc(sizeof(double) * L), *z = malloc(sizeof(double) * N);
    float *llrf = malloc(sizeof(float) * N);
    int64_t counts[3];
    for (int64_t f = 0; f < nframes; f++) {
        orc_gen_info(seed, frame0 + f, K, msg);
        orc_crc_encode(msg, Km, crc_poly, info);
        orc_polar_encode(info, mask, N, x);
        orc_gen_noise(seed, frame0 + f, N, z);
        double sigma = sqrt(N0 / 2.0), gain = 4.0 / N0;
        for (int i = 0; i < N; i++) llr[i] = gain * ((double)(1 - 2 * (int)x[i]) + sigma * z[i]);
        if (dtype == 0) {
            for (int i = 0; i < N; i++) llrf[i] = (float)llr[i];
            orc_scl_decode_f32(llrf, mask, N, L, uh, pm, counts);
        } else {
            orc_scl_decode_f64(llr, mask, N, L, uh, pm, counts);
        }
        int nsurv = L;
        if (K < 6 && (1 << K) < L) nsurv = 1 << K;
        int j = orc_ca_select(uh, mask, N, nsurv, crc_poly);
        orc_extract(uh + (size_t)j * N, mask, N, dec);
        int64_t e = orc_fails(info, dec, Km);
        counters[0] += 1;
        counters[1] += (e > 0);
        counters[2] += (uint64_t)e;
    }
    free(info); free(msg); free(x); free(uh); free(dec); free(llr); free(pm); free(z); free(llrf);
}

/* Genie-aided Monte-Carlo loop: orc_mc_run with sclDecoderOracle in place of sclDecoder (cn/simulator.py:156). */
ORC_API void orc_mc_run_genie(const uint8_t *mask, int N, int K, int L, int dtype, double N0,
                              uint64_t seed, uint64_t frame0, int64_t nframes, uint64_t counters[3])
{
    uint8_t *info = malloc(K), *x = malloc(N), *uh = malloc((size_t)N * L), *dec = malloc(K), *ut = malloc(N);
    double *llr = malloc(sizeof(double) * N), *pm = malloc(sizeof(double) * L);
    float *llrf = malloc(sizeof(float) * N);
    for (int64_t f = 0; f < nframes; f++) {
        orc_gen_frame(seed, frame0 + f, mask, N, K, N0, info, x, llr);
        int nsurv;
        if (dtype == 0) {
            for (int i = 0; i < N; i++) llrf[i] = (float)llr[i];
            nsurv = orc_scl2_decode_f32(llrf, mask, N, L, uh, pm, NULL);
        } else {
            nsurv = orc_scl2_decode_f64(llr, mask, N, L, uh, pm, NULL);
        }
        for (int i = 0, k = 0; i < N; i++) ut[i] = mask[i] ? info[k++] : 0;
        int j = orc_genie_select(uh, nsurv, N, ut);
        orc_extract(uh + (size_t)j * N, mask, N, dec);
        int64_t e = orc_fails(info, dec, K);
        counters[0] += 1;
        counters[1] += (e > 0);
        counters[2] += (uint64_t)e;
    }
    free(info); free(x); free(uh); free(dec); free(ut); free(llr); free(pm); free(llrf);
}

/* ------------------------------------------------------------------------------------------------
 * Bit flipping
 * ------------------------------------------------------------------------------------------------ */

/* pcf.generateCriticalSet (cn/functions.py:279-310): table A of frozen-leaf counts per tree node; walking the
 * levels from the root, every node without frozen leaves whose ancestors all had some contributes its first
 * leaf and blanks its descendants.  frozen: N bytes, 1 = frozen.  Returns the count; out is ascending. */
ORC_API int orc_critical_set(const uint8_t *frozen, int N, int32_t *out)
{
    int n = ilog2(N), cnt = 0;
    int32_t **A = (int32_t **)malloc(sizeof(int32_t *) * (n + 1));
    for (int i = 0; i <= n; i++) A[i] = (int32_t *)malloc(sizeof(int32_t) * ((size_t)1 << i));
    for (int j = 0; j < N; j++) A[n][j] = frozen[j];
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < (1 << i); j++) A[i][j] = A[i + 1][2 * j] + A[i + 1][2 * j + 1];
    for (int i = 0; i <= n; i++)
        for (int j = 0; j < (1 << i); j++)
            if (A[i][j] == 0) {
                int i1 = j, i2 = j;
                for (int k = 1; k <= n - i; k++) {
                    i1 = 2 * i1;
                    i2 = 2 * i2 + 1;
                    for (int q = i1; q <= i2; q++) A[i + k][q] = -1;
                }
                out[cnt++] = i1;
            }
    for (int a = 1; a < cnt; a++) { /* criticalSet.sort() */
        int32_t v = out[a];
        int b = a - 1;
        while (b >= 0 && out[b] > v) { out[b + 1] = out[b]; b--; }
        out[b + 1] = v;
    }
    for (int i = 0; i <= n; i++) free(A[i]);
    free(A);
    return cnt;
}

/* pcf.modifyCriticalSet (cn/functions.py:312-319): the critical set of the code whose information bits are
 * those of `mask` with index > iMax. */
ORC_API int orc_modify_critical_set(const uint8_t *mask, int N, int iMax, int32_t *out)
{
    uint8_t *frozen = (uint8_t *)malloc((size_t)N);
    for (int i = 0; i < N; i++) frozen[i] = (i > iMax && mask[i]) ? 0 : 1;
    int c = orc_critical_set(frozen, N, out);
    free(frozen);
    return c;
}

static int crc_tail_zero(const uint8_t *word, int K, uint64_t crc_poly, uint8_t *scratch)
{
    int w = orc_crc_encode(word, K, crc_poly, scratch), sum = 0;
    for (int t = 0; t < w; t++) sum += scratch[K + t];
    return sum == 0;
}

/* Candidates ranked by |decision LLR| / sqrt(mu), ascending (cn/polarCode.py:1586-1590, 1606-1610; numpy's
 * argsort is not stable, equal values keep index order here). */
static void rank_candidates(int32_t *S, int ns, const double *decllr, const double *mu)
{
    double *M = (double *)malloc(sizeof(double) * (ns > 0 ? ns : 1));
    for (int a = 0; a < ns; a++) M[a] = fabs(decllr[S[a]] / sqrt(mu[S[a]]));
    for (int a = 1; a < ns; a++) {
        int32_t v = S[a];
        double m = M[a];
        int b = a - 1;
        while (b >= 0 && M[b] > m) { S[b + 1] = S[b]; M[b + 1] = M[b]; b--; }
        S[b + 1] = v;
        M[b + 1] = m;
    }
    free(M);
}

/* PolarCode.progressiveBitFlipping (cn/polarCode.py:1573-1628).  SC; while the CRC fails, the flip tree is
 * searched level by level: level 0 = the critical set ranked by reliability, one SC run per node with that
 * node's flips; a failing node at a level below the last spawns children (critical set of the information bits
 * after its last flip, ranked with that run's decision LLRs) into the next level's queue.  The result is the
 * information word of the LAST run made, successful or not.  word_out: K bits; last_seq: the flips of the last
 * run, -1 padded to max_level; returns the number of SC runs. */
#define DEFINE_PBF(REAL, SUF)                                                                         \
    ORC_API int orc_pbf_decode_##SUF(const REAL *llr, const uint8_t *mask, int N, uint64_t crc_poly,  \
                                     int max_level, const double *mu, uint8_t *word_out, int32_t *last_seq) \
    {                                                                                                 \
        uint8_t *uhat = malloc(N), *frozen = malloc(N), *scratch = malloc((size_t)N + 64);            \
        REAL *dl = malloc(sizeof(REAL) * N);                                                          \
        double *dld = malloc(sizeof(double) * N);                                                     \
        int32_t *cand = malloc(sizeof(int32_t) * N);                                                  \
        int runs = 0, K;                                                                              \
        int32_t **seqs = calloc(max_level, sizeof(int32_t *));                                        \
        int *count = calloc(max_level, sizeof(int)), *cap = calloc(max_level, sizeof(int));           \
        for (int t = 0; t < max_level; t++) last_seq[t] = -1;                                         \
        orc_sc_decode_##SUF(llr, mask, N, uhat, dl, NULL, 0);                                         \
        runs++;                                                                                       \
        K = orc_extract(uhat, mask, N, word_out);                                                     \
        if (!crc_tail_zero(word_out, K, crc_poly, scratch)) {                                         \
            for (int i = 0; i < N; i++) { frozen[i] = mask[i] ? 0 : 1; dld[i] = (double)dl[i]; }      \
            int ns = orc_critical_set(frozen, N, cand);                                               \
            rank_candidates(cand, ns, dld, mu);                                                       \
            cap[0] = ns > 0 ? ns : 1;                                                                 \
            seqs[0] = malloc(sizeof(int32_t) * max_level * cap[0]);                                   \
            for (int a = 0; a < ns; a++) seqs[0][(size_t)a * max_level] = cand[a];                    \
            count[0] = ns;                                                                            \
            int l = 0, ok = 0;                                                                        \
            while (l <= max_level - 1 && !ok) {                                                       \
                for (int cur = 0; cur < count[l]; cur++) {                                            \
                    const int32_t *seq = seqs[l] + (size_t)cur * max_level;                           \
                    orc_sc_decode_##SUF(llr, mask, N, uhat, dl, seq, l + 1);                          \
                    runs++;                                                                           \
                    for (int t = 0; t < max_level; t++) last_seq[t] = t <= l ? seq[t] : -1;           \
                    orc_extract(uhat, mask, N, word_out);                                             \
                    if (crc_tail_zero(word_out, K, crc_poly, scratch)) { ok = 1; break; }             \
                    if (l + 1 <= max_level - 1) {                                                     \
                        for (int i = 0; i < N; i++) dld[i] = (double)dl[i];                           \
                        int nc = orc_modify_critical_set(mask, N, seq[l], cand);                      \
                        rank_candidates(cand, nc, dld, mu);                                           \
                        if (count[l + 1] + nc > cap[l + 1]) {                                         \
                            cap[l + 1] = 2 * (count[l + 1] + nc) + 16;                                \
                            seqs[l + 1] = realloc(seqs[l + 1], sizeof(int32_t) * max_level * cap[l + 1]); \
                            seq = seqs[l] + (size_t)cur * max_level;                                  \
                        }                                                                             \
                        for (int a = 0; a < nc; a++) {                                                \
                            int32_t *dst = seqs[l + 1] + (size_t)(count[l + 1] + a) * max_level;      \
                            for (int t = 0; t <= l; t++) dst[t] = seq[t];                             \
                            dst[l + 1] = cand[a];                                                     \
                        }                                                                             \
                        count[l + 1] += nc;                                                           \
                    }                                                                                 \
                }                                                                                     \
                l++;                                                                                  \
            }                                                                                         \
        }                                                                                             \
        for (int t = 0; t < max_level; t++) free(seqs[t]);                                            \
        free(seqs); free(count); free(cap); free(uhat); free(frozen); free(scratch); free(dl); free(dld); free(cand); \
        return runs;                                                                                  \
    }

DEFINE_PBF(double, f64)
DEFINE_PBF(float, f32)
