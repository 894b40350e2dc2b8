// Runs the host-only table builders of the library on many rate profiles; built with -fsanitize=address,undefined by
// `make -C polar-codes_amd/csrc sanitize` and run by tests/test_sanitizers.py (CPU only).
#include <stdio.h>

#include "host_tables.h"

int main()
{
    uint32_t st = 99u;
    auto rnd = [&]() {
        st = st * 1664525u + 1013904223u;
        return st >> 8;
    };
    size_t total = 0;
    for (int n = 1; n <= 12; n++) {
        const int N = 1 << n;
        for (int trial = 0; trial < 4; trial++) {
            std::vector<uint8_t> mask(N);
            int K = 0;
            for (int i = 0; i < N; i++) {
                mask[i] = trial == 0 ? 0 : trial == 1 ? 1 : trial == 2 ? (i >= N / 2) : (rnd() & 1);
                K += mask[i];
            }
            for (int fast = 0; fast < 2; fast++)
                for (int sub16 = 0; sub16 < 2; sub16++) {
                    const std::vector<uint32_t> ops = build_schedule(mask.data(), N, fast != 0, sub16 != 0);
                    if (ops.empty() || OP_TYPE(ops.back()) != OP_END) return 1;
                    total += ops.size();
                }
            for (int after : {-1, 0, N / 2, N - 1}) total += critical_set(mask.data(), N, after).size();
            {
                const std::vector<uint32_t> ct = build_compress_tables(mask.data(), N);
                if (ct.size() != 256) return 1;
                total += ct[0] & 1u;
            }
            for (int w : {1, 8, 24, 32}) {
                if (w > K) continue;
                std::vector<uint32_t> E, S;
                build_crc_tables((1ull << w) | 0x7ull | (w > 4 ? 0x10ull : 0ull), w, K, E, S);
                if ((int)S.size() != K) return 1;
                total += E.size();
            }
        }
    }
    // the dispatch table: every query in and around its ranges selects a row of the table or none
    int rows = 0;
    const DispatchRow *tab = dispatch_table(&rows);
    for (int op = -1; op <= 4; op++)
        for (int n = -1; n <= 18; n++)
            for (int dt = -1; dt <= 2; dt++)
                for (int L : {-1, 0, 1, 2, 3, 8, 16, 32, 33, 64})
                    for (int outputs : {0, 1, 3, 7, 9, 15, 255})
                        for (int flags : {0, 1, 3, 7, 64, 128, 511}) {
                            const int N = n >= 0 ? 1 << n : 0;
                            for (int K : {-1, 0, 1, N / 2, N}) {
                                const int r = dispatch_select(DispatchQuery{op, n, K, dt, L, outputs, flags});
                                if (r < -1 || r >= rows) return 1;
                                if (r >= 0 && tab[r].op != op) return 1;
                                total += (size_t)(r + 1);
                            }
                        }
    printf("host tables sanitizer run ok (%zu)\n", total);
    return 0;
}
