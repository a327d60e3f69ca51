// hmcg_csv.cpp -- native writers for the reference's per-draw CSV files (part of libhmcgibbs.so; host code only).
//
// What it replaces (joe5saia/Hmc.jl): basicsave (src/Hmc.jl:707-722) and saveresults (:724-748) -- five files per window,
//   filtered_means_<date>.csv, filtered_variances_<date>.csv, filtered_state_probs_<date>.csv,
//   filtered_trans_probs_<date>.csv, forecasts_<date>.csv
// each `date[,signalid],<columns>[,signal_1..]`, one row per kept draw, values rounded to 5 digits, written upstream by
// CSV.jl 0.5.16 from a Matrix{Any} (250 000 rows per file in production, code/run_hmm.jl:103-104).  The writers here take
// the C-ABI draw arrays as they are (Julia column-major blocks, include/hmcg.h) and produce the same bytes: LF line
// ends, CSV.jl 0.5.16's float text (shortest round-trip digits; integral values without a fraction; |x| < 1e-4 as
// <integer mantissa>e-<n>, e.g. 24e-11; plain decimals otherwise), so a Julia or C caller needs no formatter of its own.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hmcg.h"

namespace {

// CSV.jl 0.5.16 float text into buf (>= 40 bytes); returns the length.
int format_float(double x, char* buf)
{
    if (x != x) { memcpy(buf, "NaN", 3); return 3; }
    if (std::isinf(x)) { const char* s = x > 0 ? "Inf" : "-Inf"; const int n = (int)strlen(s); memcpy(buf, s, (size_t)n); return n; }
    if (x == 0.0) { buf[0] = '0'; return 1; }                       // (-0.0 prints as 0: it is integral)
    char* p = buf;
    if (x < 0) { *p++ = '-'; x = -x; }
    // shortest round-trip digits and decimal exponent: d.ddddde[+-]XX
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int e10 = 0;
    {
        const char* e = q + 1;
        const bool neg = *e == '-';
        if (*e == '-' || *e == '+') ++e;
        for (; e < r.ptr; ++e) e10 = e10 * 10 + (*e - '0');
        if (neg) e10 = -e10;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;                   // (to_chars gives none, kept for safety)
    const int point = e10 + 1;                                      // digits before the decimal point
    if (point >= nd && point <= 15) {                               // integral value: no fraction
        memcpy(p, digits, (size_t)nd); p += nd;
        for (int i = nd; i < point; ++i) *p++ = '0';
        return (int)(p - buf);
    }
    if (e10 < -4) {                                                 // |x| < 1e-4: integer mantissa, e.g. 24e-11
        memcpy(p, digits, (size_t)nd); p += nd;
        p += snprintf(p, 8, "e%d", e10 - (nd - 1));
        return (int)(p - buf);
    }
    if (point > 15) {                                               // large non-integral magnitudes: d.ddde+XX as repr would
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, (size_t)(nd - 1)); p += nd - 1; } else { *p++ = '.'; *p++ = '0'; }
        p += snprintf(p, 8, "e%d", e10);
        return (int)(p - buf);
    }
    if (point <= 0) {                                               // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = point; i < 0; ++i) *p++ = '0';
        memcpy(p, digits, (size_t)nd); p += nd;
        return (int)(p - buf);
    }
    memcpy(p, digits, (size_t)point); p += point;                   // ddd.ddd
    *p++ = '.';
    memcpy(p, digits + point, (size_t)(nd - point)); p += nd - point;
    return (int)(p - buf);
}

inline double round_digits(double v, double sc) { return std::isfinite(v) ? std::rint(v * sc) / sc : v; }   // round(x; digits) (:719)

struct Table {
    const char* path;
    const char* date;
    std::vector<std::string> cols;
    const double* data;        // column-major: element (d, c) at d + ld * c
    int64_t n, ld;
    int ncol;
    int precision;
    // signal columns (estimatesignals!, :715-717): row d belongs to noise sample d / nrun_s
    const double* sigvals;     // [n_samples][nsave_ld] or NULL
    int nsave, nsave_ld;
    int64_t nrun_s;
};

int write_table(const Table& t)
{
    FILE* f = fopen(t.path, "wb");
    if (!f) return HMCG_E_BADARG;
    std::string out;
    out.reserve(1 << 22);
    out += "date";
    if (t.sigvals) out += ",signalid";
    for (const auto& c : t.cols) { out += ','; out += c; }
    if (t.sigvals) for (int j = 0; j < t.nsave; ++j) { out += ",signal_"; out += std::to_string(j + 1); }
    out += '\n';
    const double sc = std::pow(10.0, t.precision);
    const size_t dlen = strlen(t.date);
    char buf[48];
    int rc = 0;
    for (int64_t d = 0; d < t.n; ++d) {
        out.append(t.date, dlen);
        const int64_t smp = t.sigvals ? d / (t.nrun_s > 0 ? t.nrun_s : 1) : 0;
        if (t.sigvals) { out += ','; out += std::to_string(smp + 1); }
        for (int c = 0; c < t.ncol; ++c) {
            out += ',';
            out.append(buf, (size_t)format_float(round_digits(t.data[d + t.ld * c], sc), buf));
        }
        if (t.sigvals)
            for (int j = 0; j < t.nsave; ++j) {
                out += ',';
                out.append(buf, (size_t)format_float(round_digits(t.sigvals[smp * t.nsave_ld + j], 1e5), buf));
            }
        out += '\n';
        if (out.size() > (1u << 22) - 4096) {
            if (fwrite(out.data(), 1, out.size(), f) != out.size()) rc = HMCG_E_BADARG;
            out.clear();
        }
    }
    if (fwrite(out.data(), 1, out.size(), f) != out.size()) rc = HMCG_E_BADARG;
    if (fclose(f) != 0) rc = HMCG_E_BADARG;
    return rc;
}

int save_window(const char* dir, const char* date, int K, int H, const int32_t* horizons, int64_t n, const double* mu,
                const double* sig2, const double* pi_end, const double* A, const double* fcast, const double* sigvals,
                int nsave, int nsave_ld, int64_t nrun_s, int flags)
{
    std::vector<std::string> h1, h2, h3;
    for (int i = 1; i <= K; ++i) h1.push_back("state_" + std::to_string(i));
    for (int j = 1; j <= K; ++j)
        for (int i = 1; i <= K; ++i)                      // vec of [i,j] column-major: i fastest (src/Hmc.jl:727)
            h2.push_back((flags & HMCG_CSV_LEGACY_TRANS_HEADER) ? "trans_" + std::to_string(j) + "_" + std::to_string(i)
                                                                 : "trans_" + std::to_string(i) + "_" + std::to_string(j));
    for (int h = 0; h < H; ++h) {
        h3.push_back("forecast_" + std::to_string(horizons[h]));
        h3.push_back("forecast_error_" + std::to_string(horizons[h]));
    }
    const std::string base = std::string(dir) + "/";
    struct Spec { const char* stem; const double* data; const std::vector<std::string>* cols; };
    const Spec specs[5] = { {"filtered_means_", mu, &h1}, {"filtered_variances_", sig2, &h1}, {"filtered_state_probs_", pi_end, &h1},
                            {"filtered_trans_probs_", A, &h2}, {"forecasts_", fcast, &h3} };
    for (const Spec& s : specs) {
        if (!s.data) continue;
        const std::string path = base + s.stem + date + ".csv";
        Table t{path.c_str(), date, *s.cols, s.data, n, n, (int)s.cols->size(), 5, sigvals, nsave, nsave_ld, nrun_s};
        const int rc = write_table(t);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int hmcg_format_float(double x, char* buf) { const int n = format_float(x, buf); buf[n] = 0; return n; }

int hmcg_write_table_csv(const char* path, const char* date, int32_t ncol, const char* const* colnames, const double* data,
                         int64_t n, int64_t ld, int32_t precision)
{
    if (!path || !date || !data || ncol < 0 || n < 0 || ld < n) return HMCG_E_BADARG;
    std::vector<std::string> cols;
    for (int c = 0; c < ncol; ++c) cols.emplace_back(colnames[c]);
    Table t{path, date, cols, data, n, ld, ncol, precision, nullptr, 0, 0, 0};
    return write_table(t);
}

int hmcg_save_results_csv(const char* dir, int32_t W, const char* const* dates, int32_t K, int32_t H, const int32_t* horizons,
                          int64_t nd, const double* mu, const double* sig2, const double* pi_end, const double* A,
                          const double* fcast, const double* sigvals, int32_t n_samples, int32_t nsave, int32_t nsave_ld,
                          int32_t flags, int32_t n_threads)
{
    if (!dir || !dates || W < 1 || K < 2 || K > HMCG_MAXK || H < 0 || H > HMCG_MAXH || nd < 0) return HMCG_E_BADARG;
    if (sigvals && (n_samples < 1 || nsave < 0 || nsave_ld < nsave)) return HMCG_E_BADARG;
    const int64_t nrun_s = (sigvals && n_samples > 0) ? nd / n_samples : nd;
    const size_t n = (size_t)nd, k = (size_t)K;
    int nthr = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nthr < 1) nthr = 1;
    if (nthr > W) nthr = W;
    std::vector<int> rcs((size_t)nthr, 0);
    auto work = [&](int tix) {
        for (int w = tix; w < W; w += nthr) {
            const size_t wz = (size_t)w;
            const int rc = save_window(dir, dates[w], K, H, horizons, nd, mu ? mu + wz * k * n : nullptr,
                                       sig2 ? sig2 + wz * k * n : nullptr, pi_end ? pi_end + wz * k * n : nullptr,
                                       A ? A + wz * k * k * n : nullptr, (fcast && H) ? fcast + wz * 2 * (size_t)H * n : nullptr,
                                       sigvals ? sigvals + wz * (size_t)n_samples * (size_t)nsave_ld : nullptr, nsave, nsave_ld,
                                       nrun_s, flags);
            if (rc && !rcs[(size_t)tix]) rcs[(size_t)tix] = rc;
        }
    };
    if (nthr == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthr; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int rc : rcs) if (rc) return rc;
    return 0;
}

}  // extern "C"
