// moments.hip -- correlations between the per-draw outputs of a window, accumulated on the device.
//
// What it replaces (joe5saia/Hmc.jl): calccorr (src/Hmc.jl:1094-1163) reads the five per-draw CSV files of every end
// date back from disk (250 000 rows each in production, code/run_hmm.jl:103-104), glues the columns
//   mu_1..K | sigma_1..K | pi_1..K | trans (column-major A[:]) | forecast_<first horizon>            (:1122)
// into one Nrun x (3K + K^2 + 1) matrix and takes `cor` of it (:1125).  The draws are already in HBM when a chunk of the
// chain ends, so the second moments are taken there: one pass over the chunk's draw block, values rounded to 5 digits
// exactly as the CSV cells are (round5), no per-draw file and no D2H of the draws needed for the workbook.
//
// Numerics: moments of (x - pivot), pivot = the window's first rounded draw, in fp64, each pair accumulated in draw order
// (tiles in order, chunks in order): the result does not depend on how the run is chunked.  A constant-one column rides
// along, so the pair table holds the sums and the count as well: NCp = NC + 1 columns, NCp (NCp + 1) / 2 pairs.
//
// Layout of the draw arrays (include/hmcg.h): window w, column q, draw d at base[(w * ncol + q) * nd_ld + d].
// HBM-bound by construction: every draw value is read once (coalesced along d), 8 (3K + K^2 + 1) bytes per draw.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "moments.hpp"
#include "round5.hpp"

namespace hmcg {

constexpr int MOM_TD = 64;          // draws per LDS tile
constexpr int MOM_NT = 256;

struct MomentsParams {
    const double* mu; const double* sig2; const double* pi_end; const double* A; const double* fcast;
    double* mom;                    // [W][NC pivots | NP pair accumulators]
    long long nd, nd_ld;            // draws in this block, leading dimension of the arrays
    int K, H, NC, first;            // first: this block starts the run (pivots are taken, accumulators start at zero)
};

__device__ __forceinline__ const double* column_base(const MomentsParams& p, int w, int c)
{
    const int K = p.K, KK = K * K;
    if (c < K) return p.mu + ((size_t)w * K + c) * p.nd_ld;
    if (c < 2 * K) return p.sig2 + ((size_t)w * K + (c - K)) * p.nd_ld;
    if (c < 3 * K) return p.pi_end + ((size_t)w * K + (c - 2 * K)) * p.nd_ld;
    if (c < 3 * K + KK) return p.A + ((size_t)w * KK + (c - 3 * K)) * p.nd_ld;
    return p.fcast + ((size_t)w * 2 * p.H) * p.nd_ld;                    // forecast of the first horizon (df4[!, [2]], :1122)
}

// pair number of (i, j), i <= j, in the row-major upper triangle of an n x n table
__device__ __host__ __forceinline__ int pair_index(int i, int j, int n) { return i * (2 * n - i + 1) / 2 + (j - i); }

template <int PPT>
__global__ __launch_bounds__(MOM_NT) void draw_moments_kernel(const MomentsParams p)
{
    extern __shared__ double tile[];            // [NCp][MOM_TD + 1]: padded rows (no bank conflicts between rows)
    const int w = blockIdx.x, tid = threadIdx.x;
    const int NC = p.NC, NCp = NC + 1, NP = NCp * (NCp + 1) / 2;
    constexpr int LDT = MOM_TD + 1;
    double* piv = tile + (size_t)NCp * LDT;     // [NC]
    double* mom = p.mom + (size_t)w * (NC + NP);
    for (int c = tid; c < NC; c += MOM_NT) {
        const double v = p.first ? round5(column_base(p, w, c)[0]) : mom[c];
        piv[c] = v;
        if (p.first) mom[c] = v;
    }
    // this thread's pairs
    int pi[PPT], pj[PPT];
    double acc[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pr = tid + q * MOM_NT;
        int i = 0, rem = pr < NP ? pr : 0;
        while (rem >= NCp - i) { rem -= NCp - i; ++i; }
        pi[q] = i; pj[q] = i + rem;
        acc[q] = (pr < NP && !p.first) ? mom[NC + pr] : 0.0;
    }
    __syncthreads();
    for (long long d0 = 0; d0 < p.nd; d0 += MOM_TD) {
        for (int e = tid; e < NCp * MOM_TD; e += MOM_NT) {
            const int c = e / MOM_TD, dd = e - c * MOM_TD;
            const long long d = d0 + dd;
            double v = 0.0;
            if (d < p.nd) v = c < NC ? round5(column_base(p, w, c)[d]) - piv[c] : 1.0;
            tile[c * LDT + dd] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double* a = tile + pi[q] * LDT;
            const double* b = tile + pj[q] * LDT;
            double s = acc[q];
#pragma unroll 8
            for (int dd = 0; dd < MOM_TD; ++dd) s = fma(a[dd], b[dd], s);
            acc[q] = s;
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pr = tid + q * MOM_NT;
        if (pr < NP) mom[NC + pr] = acc[q];
    }
}

// corr[w][i][j] from the pair table: cov_ij = P_ij - S_i S_j / n (the common 1/(n-1) cancels); diagonal exactly 1;
// a constant column gives NaN, as Statistics.cor does.  Clamped to [-1, 1] like cor's clampcor.
__global__ __launch_bounds__(MOM_NT) void corr_finalize_kernel(const double* mom_all, double* corr_all, int NC)
{
    const int w = blockIdx.x, NCp = NC + 1, NP = NCp * (NCp + 1) / 2;
    const double* P = mom_all + (size_t)w * (NC + NP) + NC;
    double* corr = corr_all + (size_t)w * NC * NC;
    const double n = P[pair_index(NC, NC, NCp)];
    for (int e = threadIdx.x; e < NC * NC; e += MOM_NT) {
        const int i = e / NC, j = e - i * NC;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const double Si = P[pair_index(i, NC, NCp)], Sj = P[pair_index(j, NC, NCp)];
        const double cij = P[pair_index(a, b, NCp)] - Si * Sj / n;
        const double cii = P[pair_index(i, i, NCp)] - Si * Si / n, cjj = P[pair_index(j, j, NCp)] - Sj * Sj / n;
        double r = cij / (sqrt(cii) * sqrt(cjj));
        r = r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
        if (i == j) r = (cii > 0.0) ? 1.0 : r;
        corr[e] = r;
    }
}

}  // namespace hmcg

namespace hmcg_host {

int corr_columns(int K) { return 3 * K + K * K + 1; }
size_t moments_stride(int K)
{
    const size_t NC = (size_t)corr_columns(K), NCp = NC + 1;
    return NC + NCp * (NCp + 1) / 2;
}

hipError_t launch_moments(const MomentsArgs& a, hipStream_t stream)
{
    if (a.W <= 0 || a.nd <= 0) return hipSuccess;
    hmcg::MomentsParams p{};
    p.mu = a.mu; p.sig2 = a.sig2; p.pi_end = a.pi_end; p.A = a.A; p.fcast = a.fcast;
    p.mom = a.mom; p.nd = a.nd; p.nd_ld = a.nd_ld; p.K = a.K; p.H = a.H; p.NC = corr_columns(a.K); p.first = a.first ? 1 : 0;
    const int NCp = p.NC + 1, NP = NCp * (NCp + 1) / 2;
    const size_t lds = sizeof(double) * ((size_t)NCp * (hmcg::MOM_TD + 1) + (size_t)p.NC);
    const int ppt = (NP + hmcg::MOM_NT - 1) / hmcg::MOM_NT;
    const dim3 grid((unsigned)a.W), block(hmcg::MOM_NT);
#define HMCG_MOM(PPT_)                                                                                                    \
    do {                                                                                                                  \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(hmcg::draw_moments_kernel<PPT_>),               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        if (e_ != hipSuccess) return e_;                                                                                   \
        hipLaunchKernelGGL(hmcg::draw_moments_kernel<PPT_>, grid, block, lds, stream, p);                                  \
    } while (0)
    if (ppt <= 1) HMCG_MOM(1);
    else if (ppt <= 2) HMCG_MOM(2);
    else if (ppt <= 4) HMCG_MOM(4);
    else if (ppt <= 8) HMCG_MOM(8);
    else if (ppt <= 16) HMCG_MOM(16);
    else return hipErrorInvalidValue;
#undef HMCG_MOM
    return hipGetLastError();
}

hipError_t launch_corr_finalize(const double* mom, double* corr, int W, int K, hipStream_t stream)
{
    if (W <= 0) return hipSuccess;
    hipLaunchKernelGGL(hmcg::corr_finalize_kernel, dim3((unsigned)W), dim3(hmcg::MOM_NT), 0, stream, mom, corr, corr_columns(K));
    return hipGetLastError();
}

}  // namespace hmcg_host
