// Sanitizer harness for the host-side code of libhmcgibbs.so that needs no GPU (tests/test_sanitizers.py builds it with
// g++ -fsanitize=address,undefined and with -fsanitize=thread and runs it on the CPU):
//   * hmcg_hostutil::ScatterPool -- the helper threads of the host entry's scatter: start / run / stop / start again with a
//     job posted at once (the generation race of ADVICE r2), many jobs, jobs that write disjoint parts of one array;
//   * hmcg_hostutil::partition_windows, plan_chunks -- invariants on random shapes;
//   * hmcg_csv.cpp -- hmcg_format_float, hmcg_write_table_csv and hmcg_save_results_csv with its thread pool.
// Exits 0 when every check holds; sanitizer reports make it exit non-zero by themselves.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <sys/stat.h>

#include "../../hmc.jl_amd/csrc/host_util.hpp"
#include "../../include/hmcg.h"

using namespace hmcg_hostutil;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

static int pool_checks()
{
    std::vector<double> src(1 << 18), dst(1 << 18);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (double)i * 0.5;
    for (int round = 0; round < 40; ++round) {
        ScatterPool pool;
        pool.start(round % 4);                              // 0..3 helpers
        const int jobs = round < 8 ? 1 : 25;                // the first rounds post their only job right after start()
        for (int j = 0; j < jobs; ++j) {
            std::fill(dst.begin(), dst.end(), -1.0);
            const std::function<void(int, int)> part = [&](int pi, int np) {
                const size_t n = src.size(), i0 = n * (size_t)pi / (size_t)np, i1 = n * (size_t)(pi + 1) / (size_t)np;
                memcpy(dst.data() + i0, src.data() + i0, 8 * (i1 - i0));
            };
            pool.run(part);
            CHECK(memcmp(dst.data(), src.data(), 8 * src.size()) == 0);       // every part done when run() returns
        }
        pool.stop();
        pool.start(3);                                      // same object again: generation counter has moved on
        int hits[4] = {0, 0, 0, 0};
        const std::function<void(int, int)> count = [&](int pi, int np) { (void)np; hits[pi] += 1; };
        pool.run(count);
        CHECK(hits[0] == 1 && hits[1] == 1 && hits[2] == 1 && hits[3] == 1);
        pool.stop();
    }
    return 0;
}

static int partition_checks()
{
    std::mt19937 g(7);
    for (int rep = 0; rep < 300; ++rep) {
        const int W = 1 + (int)(g() % 700), G = 1 + (int)(g() % 8);
        std::vector<int32_t> T((size_t)W);
        for (auto& t : T) t = 2 + (int32_t)(g() % 5000);
        const int Gu = std::min(G, W);
        const auto parts = partition_windows(T.data(), W, Gu);
        CHECK((int)parts.size() == Gu);
        std::vector<int> seen((size_t)W, 0);
        const int cap = (W + Gu - 1) / Gu;
        long long lo = 1LL << 60, hi = 0;
        for (const auto& p : parts) {
            CHECK((int)p.size() <= cap);
            long long load = 0;
            for (size_t i = 0; i < p.size(); ++i) { CHECK(p[i] >= 0 && p[i] < W); seen[(size_t)p[i]]++; load += T[(size_t)p[i]]; CHECK(i == 0 || p[i - 1] < p[i]); }
            lo = std::min(lo, load); hi = std::max(hi, load);
        }
        for (int s : seen) CHECK(s == 1);
        if (W >= 4 * Gu) CHECK(hi - lo <= 2 * 5002LL + (hi / 4));       // LPT under a count cap: loads stay close
    }
    return 0;
}

static int chunk_checks()
{
    std::mt19937 g(11);
    for (int rep = 0; rep < 2000; ++rep) {
        const int burnin = (int)(g() % 50), nrun = (int)(g() % 400), ns = 1 + (int)(g() % 4);
        const int per = std::max(1, burnin + nrun), total = ns * (burnin + nrun);
        int sb = total > 0 ? (int)(g() % (unsigned)(total + 1)) : 0, se = total;
        if (g() & 1) se = sb + (int)(g() % (unsigned)(total - sb + 1));
        const long long cap = 1 + (long long)(g() % 300);
        const bool stream = (g() % 4) != 0;
        const auto ch = plan_chunks(sb, se, per, burnin, nrun, cap, stream);
        CHECK(!ch.empty() && ch.front().s0 == sb && ch.back().s1 == se);
        CHECK(ch.front().d0 == kept_before(sb, per, burnin, nrun) && ch.back().d1 == kept_before(se, per, burnin, nrun));
        for (size_t i = 0; i < ch.size(); ++i) {
            CHECK(ch[i].s0 <= ch[i].s1 && ch[i].d0 <= ch[i].d1);
            if (i) CHECK(ch[i].s0 == ch[i - 1].s1 && ch[i].d0 == ch[i - 1].d1);
            CHECK(kept_before(ch[i].s1, per, burnin, nrun) - kept_before(ch[i].s0, per, burnin, nrun) == ch[i].d1 - ch[i].d0);
            if (stream && ch.size() > 1) CHECK(ch[i].d1 - ch[i].d0 <= cap);
        }
    }
    return 0;
}

static int csv_checks(const char* dir)
{
    char buf[64];
    const double xs[] = {0.0, 1.0, -1.0, 5.0, 0.1, 6.179999828338623, 24e-11, 46210605925778346e-23, 1e22, 1.7976931348623157e308,
                         4.9e-324, -2.5e-5, 123456.789, NAN, INFINITY, -INFINITY, 0.30000000000000004, 1e-4, 9.999e-5, 1e15, 1e16};
    for (double x : xs) {
        const int n = hmcg_format_float(x, buf);
        CHECK(n > 0 && n < 48 && (int)strlen(buf) == n);
        if (std::isfinite(x)) { double back = 0; CHECK(sscanf(buf, "%lf", &back) == 1 && back == x); }      // shortest round-trip text
    }
    const int K = 3, H = 2, W = 5;
    const int64_t nd = 700;
    const int32_t horizons[2] = {1, 12};
    std::mt19937_64 g(3);
    std::uniform_real_distribution<double> u(-10.0, 10.0);
    auto fill = [&](size_t n) { std::vector<double> v(n); for (auto& x : v) x = u(g); return v; };
    const auto mu = fill((size_t)W * K * nd), sig2 = fill((size_t)W * K * nd), pe = fill((size_t)W * K * nd), A = fill((size_t)W * K * K * nd),
               fc = fill((size_t)W * 2 * H * nd), sv = fill((size_t)W * 2 * 3);
    const char* dates[W] = {"1980-01-01", "1980-02-01", "1980-03-01", "1980-04-01", "1980-05-01"};
    CHECK(hmcg_save_results_csv(dir, W, dates, K, H, horizons, nd, mu.data(), sig2.data(), pe.data(), A.data(), fc.data(), nullptr, 0, 0, 0, 0, 4) == 0);
    // the signal layout: nd = n_samples * nrun, sigvals [W][n_samples][nsave_ld]
    const std::string d2 = std::string(dir) + "/sig";
    mkdir(d2.c_str(), 0755);
    CHECK(hmcg_save_results_csv(d2.c_str(), W, dates, K, H, horizons, nd, mu.data(), sig2.data(), pe.data(), A.data(), fc.data(), sv.data(), 2, 2, 3,
                                HMCG_CSV_LEGACY_TRANS_HEADER, 3) == 0);
    const char* names[3] = {"a", "b", "c"};
    const std::string p3 = std::string(dir) + "/table.csv";
    CHECK(hmcg_write_table_csv(p3.c_str(), "2000-01-01", 3, names, mu.data(), 100, nd, 5) == 0);
    CHECK(hmcg_write_table_csv("/nonexistent-dir/x.csv", "2000-01-01", 3, names, mu.data(), 100, nd, 5) != 0);
    FILE* f = fopen(p3.c_str(), "r");
    CHECK(f != nullptr);
    int lines = 0;
    for (int c; (c = fgetc(f)) != EOF;) lines += c == '\n';
    fclose(f);
    CHECK(lines == 101);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <scratch dir>\n", argv[0]); return 2; }
    if (pool_checks()) return 1;
    if (partition_checks()) return 1;
    if (chunk_checks()) return 1;
    if (csv_checks(argv[1])) return 1;
    printf("host harness ok\n");
    return 0;
}
