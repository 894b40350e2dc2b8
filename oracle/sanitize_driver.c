/* Exercises every entry point of the CPU oracle on small random inputs; built with -fsanitize=address,undefined by
 * `make -C oracle sanitize` and run by tests/test_sanitizers.py (CPU only).  Test infrastructure. */
#include "polar_oracle.c"

#include <stdio.h>

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static double gauss(void)
{
    double s = 0;
    for (int i = 0; i < 12; i++) s += (rnd() & 0xFFFF) / 65536.0;
    return s - 6.0;
}

int main(void)
{
    const int sizes[] = {2, 8, 64, 256};
    for (int si = 0; si < 4; si++) {
        const int N = sizes[si];
        for (int trial = 0; trial < 3; trial++) {
            uint8_t mask[256], info[256], x[256], u[256], uh[32 * 256], word[256], frozen[256];
            double llr[256], pm[32], mu[256];
            float llrf[256];
            int32_t cs[256], last[3];
            int64_t counts[3];
            int K = 0;
            for (int i = 0; i < N; i++) { mask[i] = (trial == 0) ? (i >= N / 2) : (rnd() & 1); K += mask[i]; frozen[i] = !mask[i]; mu[i] = 1.0 + i; }
            for (int i = 0; i < K; i++) info[i] = rnd() & 1;
            orc_rate_profile(info, mask, N, u);
            orc_polar_encode(info, mask, N, x);
            for (int i = 0; i < N; i++) { llr[i] = 2.0 * (1 - 2 * x[i]) + 1.5 * gauss(); llrf[i] = (float)llr[i]; }
            if (trial == 2) llr[0] = 900.0, llrf[0] = 900.0f; /* metric overflow */
            orc_sc_decode_f64(llr, mask, N, uh, NULL, NULL, 0);
            orc_sc_decode_f32(llrf, mask, N, uh, NULL, NULL, 0);
            orc_extract(uh, mask, N, word);
            (void)orc_fails(word, info, K);
            for (int L = 1; L <= 32; L *= 2) {
                int n1 = orc_scl_decode_f64(llr, mask, N, L, uh, pm, counts);
                int n2 = orc_scl_decode_f32(llrf, mask, N, L, uh, pm, counts);
                int n3 = orc_scl2_decode_f64(llr, mask, N, L, uh, pm, counts);
                (void)orc_genie_select(uh, n3, N, u);
                if (n1 != n2 || n1 != n3) { printf("list sizes differ\n"); return 1; }
                if (K >= 8) (void)orc_ca_select(uh, mask, N, n3, 0x107);
            }
            (void)orc_scl_last_min_gap();
            (void)orc_critical_set(frozen, N, cs);
            (void)orc_modify_critical_set(mask, N, N / 2, cs);
            if (K > 8) {
                for (int T = 1; T <= 3; T++) {
                    (void)orc_pbf_decode_f64(llr, mask, N, 0x107, T, mu, word, last);
                    (void)orc_pbf_decode_f32(llrf, mask, N, 0x107, T, mu, word, last);
                }
                uint8_t enc[300];
                (void)orc_crc_encode(info, K > 40 ? 40 : K, 0x107, enc);
            }
            uint64_t cnt[3] = {0, 0, 0};
            orc_mc_run(mask, N, K, 0, 1, 0, 1.26, 7, 0, 20, cnt);
            orc_mc_run(mask, N, K, 1, 4, 1, 1.26, 7, 0, 5, cnt);
            if (K > 8) orc_mc_run_crc(mask, N, K, 4, 0, 1.26, 0x107, 7, 0, 5, cnt);
            orc_mc_run_genie(mask, N, K, 4, 0, 1.26, 7, 0, 5, cnt);
        }
    }
    printf("oracle sanitizer run ok\n");
    return 0;
}
