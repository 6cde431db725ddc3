// parity_driver.cpp -- native GPU-vs-oracle parity sweep (test infrastructure).
//
// Links liblapwarm_hip.so (the product, through its C ABI) and libjv_oracle.so (the CPU
// checker) and compares them bit for bit on seeded synthetic cases.  Used on the GPU box for
// fast iteration (no Python / torch start-up); tests/test_gpu_parity.py wraps it for pytest.
//
// usage: parity_driver [max_n] [reps] [verbose]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/lapwarm_hip.h"
#include "../../oracle/jv_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t next_u64()
{
    uint64_t x = rng_state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return rng_state = x;
}
static double uni() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
static double gauss()
{
    const double u1 = uni() + 1e-300, u2 = uni();
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static void make_family(const std::string &fam, int n, std::vector<double> &C)
{
    C.resize((size_t)n * n);
    if (fam == "uniform") {
        for (auto &c : C) c = uni();
    } else if (fam == "int9") {
        for (auto &c : C) c = 1.0 + (double)(next_u64() % 9);
    } else if (fam == "int100") {
        for (auto &c : C) c = 1.0 + (double)(next_u64() % 100);
    } else if (fam == "tie") {
        for (auto &c : C) c = (double)(next_u64() % 5) / 5.0 + 1e-6 * uni();
    } else if (fam == "sparse") {
        for (auto &c : C) c = (uni() < 0.3) ? uni() : 1e6;
        for (int i = 0; i < n; ++i) {  // keep a feasible permutation
            const int j = (i * 7 + 3) % n;
            if (C[(size_t)i * n + j] >= 1e6) C[(size_t)i * n + j] = uni();
        }
    } else if (fam == "metric") {
        std::vector<double> px(n), py(n);
        for (int i = 0; i < n; ++i) {
            px[i] = 100 * uni();
            py[i] = 100 * uni();
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                C[(size_t)i * n + j] = sqrt((px[i] - px[j]) * (px[i] - px[j]) + (py[i] - py[j]) * (py[i] - py[j]));
    } else if (fam == "clustered") {
        const int bs = n / 4 > 0 ? n / 4 : 1;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double c = uni();
                const int bi = (i / bs > 3) ? 3 : i / bs, bj = (j / bs > 3) ? 3 : j / bs;
                if (bi == bj) c -= 0.4;
                c += 0.1 * gauss();
                C[(size_t)i * n + j] = c < 0 ? 0 : c;
            }
    } else if (fam == "twozero") {
        for (auto &c : C) c = 1.0;
        for (int i = 0; i < n; ++i) {
            C[(size_t)i * n + i] = 0;
            C[(size_t)i * n + (i + 1) % n] = 0;
        }
    } else if (fam == "sparse_neg") {  // 1e6 fills and entries above LARGE in column 0
        for (auto &c : C) c = (uni() < 0.3) ? uni() : 1e6;
        for (int i = 0; i < n; ++i) {
            const int j = (i * 5 + 1) % n;
            if (C[(size_t)i * n + j] >= 1e6) C[(size_t)i * n + j] = uni();
            if (i % 3 == 0) C[(size_t)i * n] = 2e6 + uni();
        }
    } else if (fam == "uniform1e8") {
        for (auto &c : C) c = 1e8 * uni();
    }
}

static void make_seeds(const std::string &kind, int n, const std::vector<double> &C, std::vector<double> &u,
                       std::vector<double> &v)
{
    u.assign(n, 0.0);
    v.assign(n, 0.0);
    if (kind == "zeros") return;
    for (int i = 0; i < n; ++i) {
        double m = INFINITY;
        for (int j = 0; j < n; ++j) m = fmin(m, C[(size_t)i * n + j]);
        u[i] = m;
    }
    double scale = 1.0;
    if (kind == "noisy") {
        for (int i = 0; i < n; ++i) u[i] += 0.05 * scale * gauss();
    } else if (kind == "randu") {
        for (int i = 0; i < n; ++i) u[i] = 0.3 * gauss();
    } else if (kind == "arr") {
        for (int i = 0; i < n; ++i) u[i] += 0.02e8 * gauss();
    } else if (kind == "huge") {
        for (int i = 0; i < n; ++i) u[i] += 1e5 * gauss();
    }
    if (kind == "rowmin32") {
        for (int i = 0; i < n; ++i) u[i] = (double)(float)u[i];
        for (int j = 0; j < n; ++j) {
            float m = INFINITY;
            for (int i = 0; i < n; ++i) m = fminf(m, (float)C[(size_t)i * n + j] - (float)u[i]);
            v[j] = (double)m;
        }
    } else {
        for (int j = 0; j < n; ++j) {
            double m = INFINITY;
            for (int i = 0; i < n; ++i) m = fmin(m, C[(size_t)i * n + j] - u[i]);
            v[j] = m;
        }
    }
    if (kind == "noisy")
        for (int j = 0; j < n; ++j) v[j] += 0.05 * gauss();
    if (kind == "huge")
        for (int j = 0; j < n; ++j) v[j] += 1e5 * gauss();
}

int main(int argc, char **argv)
{
    const int max_n = argc > 1 ? atoi(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 2;
    const int verbose = argc > 3 ? atoi(argv[3]) : 0;
    const int gpu_repeat = getenv("PARITY_GPU_REPEAT") ? atoi(getenv("PARITY_GPU_REPEAT")) : 1;
    const int repeat_min_n = getenv("PARITY_REPEAT_MIN_N") ? atoi(getenv("PARITY_REPEAT_MIN_N")) : 0;
    const char *only = getenv("PARITY_ONLY");
    const char *fams[] = {"uniform", "int9", "int100", "tie", "sparse", "metric", "clustered", "twozero", "sparse_neg", "uniform1e8"};
    const char *kinds[] = {"zeros", "rowmin", "rowmin32", "noisy", "randu", "huge", "arr"};
    const int sizes[] = {1, 2, 3, 5, 8, 16, 33, 64, 100, 128, 200, 256, 400, 512, 777, 1024, 2048, 4096};
    int total = 0, bad = 0;
    long long branch_hist[5] = {0, 0, 0, 0, 0};
    int ret_hist_m3 = 0, proj_cases = 0, arr_cases = 0;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> C, u, v;
    for (int n : sizes) {
        if (n > max_n) break;
        for (const char *fam : fams) {
            for (const char *kind : kinds) {
                if (!strcmp(kind, "arr") && strcmp(fam, "uniform1e8")) continue;
                if (!strcmp(fam, "uniform1e8") && strcmp(kind, "arr") && strcmp(kind, "rowmin")) continue;
                if (!strcmp(kind, "huge") && strncmp(fam, "sparse", 6)) continue;
                const int r_eff = (n >= 1024) ? 1 : reps;
                for (int rep = 0; rep < r_eff; ++rep) {
                    make_family(fam, n, C);
                    make_seeds(kind, n, C, u, v);
                    if (only) {  // PARITY_ONLY=fam:kind:n -- same random stream, solve just that case
                        char key[96];
                        snprintf(key, sizeof(key), "%s:%s:%d", fam, kind, n);
                        if (strcmp(key, only)) continue;
                    }
                    std::vector<long long> xo(n, -1), yo(n, -1), xg(n, -1), yg(n, -1);
                    jvo_stats st;
                    const int ro = jvo_lapjv_seeded_ex(C.data(), n, n, xo.data(), yo.data(), u.data(), v.data(), 1e-12, &st, nullptr, nullptr);
                    int rg = lapjv_seeded(C.data(), n, n, xg.data(), yg.data(), u.data(), v.data(), 1e-12);
                    // PARITY_GPU_REPEAT=K (with PARITY_REPEAT_MIN_N): solve the same instance K times on
                    // the GPU and stop at the first run that differs from the oracle (race hunting)
                    for (int q = 1; q < ((n >= repeat_min_n) ? gpu_repeat : 1); ++q) {
                        const bool ok_q = (ro == rg) && (ro != 0 || (!memcmp(xo.data(), xg.data(), sizeof(long long) * n) &&
                                                                     !memcmp(yo.data(), yg.data(), sizeof(long long) * n)));
                        if (!ok_q) {
                            printf("  (differs at GPU repetition %d)\n", q - 1);
                            break;
                        }
                        rg = lapjv_seeded(C.data(), n, n, xg.data(), yg.data(), u.data(), v.data(), 1e-12);
                        if (q % 20 == 0) {
                            printf("  ... %s %s n=%d repetition %d\n", fam, kind, n, q);
                            fflush(stdout);
                        }
                    }
                    ++total;
                    if (ro == 0) branch_hist[st.branch]++;
                    if (ro == -3) ret_hist_m3++;
                    if (st.proj_events) proj_cases++;
                    if (st.arr_fired) arr_cases++;
                    bool ok = (ro == rg);
                    if (ok && ro == 0) ok = !memcmp(xo.data(), xg.data(), sizeof(long long) * n) && !memcmp(yo.data(), yg.data(), sizeof(long long) * n);
                    if (!ok) {
                        ++bad;
                        int first = -1;
                        for (int i = 0; i < n && first < 0; ++i)
                            if (xo[i] != xg[i]) first = i;
                        printf("MISMATCH seeded fam=%s n=%d kind=%s rep=%d ret oracle=%d gpu=%d branch=%lld first_diff_row=%d (%s)\n", fam, n,
                               kind, rep, ro, rg, st.branch, first, lapwarm_last_error());
                        if (bad > 40) goto done;
                    } else if (verbose) {
                        printf("ok seeded fam=%s n=%d kind=%s ret=%d branch=%lld proj=%lld paths=%lld\n", fam, n, kind, ro, st.branch,
                               st.proj_events, st.paths);
                    }
                }
            }
            // cold solve
            if (strcmp(fam, "uniform1e8")) {
                make_family(fam, n, C);
                if (only) continue;  // (the stream stays in step; only the selected seeded case is solved)
                std::vector<int> xo(n), yo(n), xg(n, -7), yg(n, -7);
                const int ro = jvo_lapjv_dense(C.data(), n, xo.data(), yo.data(), nullptr);
                const int rg = lapwarm_lapjv_dense(C.data(), n, xg.data(), yg.data());
                ++total;
                if (ro != rg || memcmp(xo.data(), xg.data(), sizeof(int) * n) || memcmp(yo.data(), yg.data(), sizeof(int) * n)) {
                    ++bad;
                    printf("MISMATCH cold fam=%s n=%d ret oracle=%d gpu=%d (%s)\n", fam, n, ro, rg, lapwarm_last_error());
                    if (bad > 40) goto done;
                } else if (verbose) {
                    printf("ok cold fam=%s n=%d\n", fam, n);
                }
            }
        }
        {
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("[n<=%d] cases=%d bad=%d elapsed=%.1fs\n", n, total, bad, secs);
            fflush(stdout);
        }
    }
done:
    printf("SUMMARY total=%d bad=%d branches ssp=%lld all_matched=%lld fallback=%lld ret-3=%d proj_cases=%d arr_cases=%d\n", total, bad,
           branch_hist[1], branch_hist[2], branch_hist[3], ret_hist_m3, proj_cases, arr_cases);
    return bad ? 1 : 0;
}
