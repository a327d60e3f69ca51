// gibbs_device.hpp -- device side of libhmcgibbs: one persistent workgroup per
// window runs every Gibbs sweep of that window's chain with the whole chain state
// (Y, X, filtered probabilities) resident in registers/LDS.  gfx950 (wave64) only.
//
// Reference behaviour restated here (joe5saia/Hmc.jl, src/Hmc.jl):
//   gibbssweep!  :486-515   update_mu_sigma! :231-336   update_beta! :338-348
//   update_rho!  :350-356   update_A!  :358-369          forwardupdate_P! :371-440
//   update_X!    :459-484   forecast   :658-667          makeParams :161-195
//   HyperParams(Y,D) :132-142
//
// Parallel decomposition (not in the reference, which is sequential in t):
//   * thread j owns the L consecutive time steps t = j*L .. j*L+L-1;
//   * forward filter  = prefix scan of the (scaled) K x K matrices A*diag(f_t):
//     per-thread product of L matrices, Kogge-Stone scan over the 64 lanes of a
//     wave, wave totals through LDS, then a per-thread replay of the normalised
//     recursion from the scanned prefix vector;
//   * backward sampling = suffix scan of the random maps g_t : X_{t+1} -> X_t
//     (each map is the inverse-CDF draw for a fixed pre-drawn uniform), composed
//     as 4-bit-per-entry tables;
//   * sufficient statistics and transition counts = wave ballots / butterflies.
// Only per-draw outputs (3K + K^2 + 2H doubles) leave the chip per sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hmcg.h"

namespace hmcg {

enum { SITE_SIG2 = 0, SITE_MU = 1, SITE_RHO = 2, SITE_A = 3, SITE_X = 4 };
constexpr int GAMMA_MAX_ATTEMPTS = 64;
constexpr double INVSQRT2PI = 0.3989422804014327;
constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double EPS64 = 2.220446049250313e-16;

struct KernelParams {
    const double* Y;        // [W][ldY]
    const int32_t* T;       // [W]
    const double* yreal;    // [W][H] or null
    int32_t ldY, W, H, nrun;
    int32_t sweep_begin;    // global index of first sweep of this launch
    int32_t sweep_end;      // one past the last sweep of this launch
    int32_t keep_from;      // global sweep index of the first kept draw (= burnin)
    int32_t resume;         // 1: load chain from xstate, 0: makeParams init
    int32_t final_launch;   // 1: write summary
    int32_t horizons[HMCG_MAXH];
    uint32_t seed_lo, seed_hi, window_base;
    double alpha, nu;
    double* mu; double* sig2; double* A; double* pi_end; double* fcast; double* summary;
    int32_t* status;
    const int32_t* x_init; int32_t* x_final; double* pif_final; uint8_t* xstate; double* sumacc;
    const uint32_t* window_ids;
};

// ------------------------------------------------------------------ RNG ----
// Philox4x32-10, counter = (index, site<<16|element, sweep, window), key = seed.

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct Rng {
    uint32_t k0, k1, window, sweep;
    __device__ __forceinline__ void block(uint32_t site, uint32_t elem, uint32_t idx, uint32_t (&out)[4]) const
    {
        out[0] = idx; out[1] = (site << 16) | elem; out[2] = sweep; out[3] = window;
        philox4x32_10(out, k0, k1);
    }
};

__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    const uint64_t x = ((uint64_t)a << 21) | (uint64_t)(b >> 11);
    return (double)x * 0x1.0p-53;
}

__device__ __forceinline__ double box_muller(const uint32_t (&r)[4])
{
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    return sqrt(-2.0 * log(1.0 - u1)) * cos(TWO_PI * u2);
}

// Gamma(shape,1): shape==1 -> exponential; shape>1 -> Marsaglia-Tsang; shape<1 -> boost.
__device__ inline double gamma_draw(const Rng& g, uint32_t site, uint32_t elem, double shape, int& status)
{
    uint32_t r[4];
    if (shape == 1.0) {
        g.block(site, elem, 0, r);
        return -log(1.0 - u53(r[0], r[1]));
    }
    const double a = shape < 1.0 ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double out = d;
    int j = 0;
    for (; j < GAMMA_MAX_ATTEMPTS; ++j) {
        g.block(site, elem, 2u * (uint32_t)j, r);
        const double x = box_muller(r);
        g.block(site, elem, 2u * (uint32_t)j + 1u, r);
        const double u = u53(r[0], r[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(1.0 - u) < 0.5 * x * x + d - d * v + d * log(v)) { out = d * v; break; }
    }
    if (j == GAMMA_MAX_ATTEMPTS) status |= HMCG_ST_GAMMA_CAP;
    if (shape < 1.0) {
        g.block(site, elem, 0xFFFFFFFFu, r);
        out *= pow(1.0 - u53(r[0], r[1]), 1.0 / shape);
    }
    return out;
}

// ------------------------------------------------------- wave helpers ----

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}

// Scale every entry by the power of two that brings the largest one into [0.5,1).
// Exact (no rounding) away from the subnormal range; a zero matrix stays zero.
template <int N>
__device__ __forceinline__ void rescale_pow2(double (&q)[N])
{
    double m = q[0];
#pragma unroll
    for (int i = 1; i < N; ++i) m = fmax(m, q[i]);
    const int e = (m > 0.0 && m < 1.0e300) ? -ilogb(m) - 1 : 0;
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = ldexp(q[i], e);
}

// 4-bit-per-entry state maps (K <= 8).  entry s of map m: (m >> 4s) & 15.
template <int K>
__device__ __forceinline__ uint32_t map_identity()
{
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < K; ++s) m |= (uint32_t)s << (4 * s);
    return m;
}
template <int K>
__device__ __forceinline__ uint32_t map_const(int v)
{
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < K; ++s) m |= (uint32_t)v << (4 * s);
    return m;
}
// (a o b)[s] = a[b[s]] : apply b first, then a.
template <int K>
__device__ __forceinline__ uint32_t map_compose(uint32_t a, uint32_t b)
{
    uint32_t m = 0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const uint32_t bs = (b >> (4 * s)) & 15u;
        m |= ((a >> (4u * bs)) & 15u) << (4 * s);
    }
    return m;
}
__device__ __forceinline__ int map_apply(uint32_t m, int s) { return (int)((m >> (4 * s)) & 15u); }

// Julia round(x; digits=5) (basicsave, src/Hmc.jl:719)
__device__ __forceinline__ double round5(double x)
{
    const double r = rint(x * 1e5) / 1e5;
    return isfinite(r) ? r : x;
}

// forecast (src/Hmc.jl:658-667): (pi' A^h) . mu, A^h on Julia's power_by_squaring schedule.
template <int K>
__device__ inline void matmul_small(double (&out)[K][K], const double (&a)[K][K], const double (&b)[K][K])
{
    double t[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) acc += a[i][k] * b[k][j];
            t[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) out[i][j] = t[i][j];
}

template <int K>
__device__ inline double forecast_value(const double (&mu)[K], const double (&A)[K][K], const double (&pe)[K], int h)
{
    double x[K][K], y[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) { x[i][j] = A[i][j]; y[i][j] = (i == j) ? 1.0 : 0.0; }
    if (h > 0) {
        unsigned p = (unsigned)h;
        int t = __ffs((int)p);          // trailing_zeros + 1
        p >>= t;
        while (--t > 0) matmul_small<K>(x, x, x);
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) y[i][j] = x[i][j];
        while (p > 0) {
            t = __ffs((int)p);
            p >>= t;
            while (--t >= 0) matmul_small<K>(x, x, x);
            matmul_small<K>(y, y, x);
        }
    }
    double f = 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        double s1 = 0.0;
#pragma unroll
        for (int r = 0; r < K; ++r) s1 += pe[r] * y[r][s];
        f += s1 * mu[s];
    }
    return f;
}

// ---------------------------------------------------------------- LDS ----

template <int K, int NT>
struct SweepShared {
    static constexpr int NW = NT / 64;
    int red_cnt[NW][K + K * K];   // per-wave state counts and transition counts
    double red_sum[NW][K];        // per-wave sums of Y by state
    double red_sse[NW][K];        // per-wave sums of squared deviations by state
    double th_mu[K], th_sig2[K], th_isd[K], th_coef[K], th_rho[K];
    double th_A[K][K];
    double wtot[NW][K * K];       // forward scan: wave totals
    double pfirst[NT + 1][K];     // filtered probs at each thread's first step
    double pi_end[K];             // unsorted pif[T-1,:]
    uint32_t wmap[NW];            // backward scan: wave totals
    int xlast;                    // X[T-1]
    int xfirst[NT + 1];           // init only: first state of each thread's chunk
    double bred[NW];              // generic block reductions (init)
    double med[2];
};

template <int NW>
__device__ __forceinline__ double block_sum(double v, double* bred, int wave, int lane)
{
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) bred[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += bred[w];
    return t;
}
template <int NW>
__device__ __forceinline__ double block_minmax(double v, double* bred, int wave, int lane, bool is_max)
{
    v = is_max ? wave_max(v) : wave_min(v);
    __syncthreads();
    if (lane == 0) bred[wave] = v;
    __syncthreads();
    double t = bred[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = is_max ? fmax(t, bred[w]) : fmin(t, bred[w]);
    return t;
}

// ------------------------------------------------------------- kernel ----

template <int K, int L, int NT>
__global__ __launch_bounds__(NT) void gibbs_sweeps_kernel(const KernelParams p)
{
    static_assert(K >= 2 && K <= 7, "small-K kernel: all parameter draws fit one wave");
    static_assert(NT % 64 == 0 && NT >= 64, "whole waves");
    constexpr int NW = NT / 64;
    constexpr int KK = K * K;
    constexpr int ND = 2 * K + KK;       // parameter-draw roles: sig2/mu (K), rho (K), A (K*K)
    static_assert(ND <= 64, "draw roles must fit wave 0");

    extern __shared__ double dyn_lds[];  // init only: Y staged for the median (NT*L doubles)
    __shared__ SweepShared<K, NT> sh;

    const int w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T[w];
    const int t0 = tid * L;
    int st = 0;

    if (T < 2 || T > NT * L) {           // uniform per block
        if (tid == 0) atomicOr(&p.status[w], HMCG_ST_BAD_T);
        return;
    }

    // ---- load the window's observations (once per launch) ----
    double y[L];
    int x[L];
    bool valid[L];
    bool bad = false;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        valid[l] = (t0 + l) < T;
        y[l] = valid[l] ? p.Y[(size_t)w * p.ldY + t0 + l] : 0.0;
        bad |= valid[l] && !isfinite(y[l]);
        x[l] = 0;
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (tid == 0) atomicOr(&p.status[w], HMCG_ST_NONFINITE);
        return;
    }

    // ---- HyperParams(Y,D): xi = mean(Y)  (src/Hmc.jl:136) ----
    double part = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) part += y[l];
    const double ymean = block_sum<NW>(part, sh.bred, wave, lane) / (double)T;
    const double xi = ymean;

    if (p.resume) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (valid[l]) x[l] = p.xstate[(size_t)w * p.ldY + t0 + l];
    } else if (p.x_init) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (valid[l]) x[l] = p.x_init[(size_t)w * p.ldY + t0 + l];
    } else {
        // ---- makeParams (src/Hmc.jl:161-195): mu spread around the median, X = argmax pdf.
        // The initial "sigma" (= std(Y), :177) is the same for every state and is overwritten by the
        // first sweep before any other use, so argmax pdf == nearest initial mean (first index on
        // ties) and std(Y) itself is never needed.  Distances, not pdf values, are compared so the
        // decision does not depend on the exp implementation (see oracle/hmc_oracle.c chain_init).
        double lmin = 1.0e308, lmax = -1.0e308;
#pragma unroll
        for (int l = 0; l < L; ++l) if (valid[l]) { lmin = fmin(lmin, y[l]); lmax = fmax(lmax, y[l]); }
        const double ymin = block_minmax<NW>(lmin, sh.bred, wave, lane, false);
        const double ymax = block_minmax<NW>(lmax, sh.bred, wave, lane, true);
        // median by rank counting over the LDS-staged window
#pragma unroll
        for (int l = 0; l < L; ++l) dyn_lds[t0 + l] = y[l];
        __syncthreads();
        int rank[L];
#pragma unroll
        for (int l = 0; l < L; ++l) rank[l] = 0;
        for (int j = 0; j < T; ++j) {
            const double yj = dyn_lds[j];
#pragma unroll
            for (int l = 0; l < L; ++l) rank[l] += (yj < y[l] || (yj == y[l] && j < t0 + l)) ? 1 : 0;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (valid[l] && rank[l] == (T - 1) / 2) sh.med[0] = y[l];
            if (valid[l] && rank[l] == T / 2) sh.med[1] = y[l];
        }
        __syncthreads();
        const double med = (T & 1) ? sh.med[0] : sh.med[0] / 2 + sh.med[1] / 2;
        const double R = ymax - ymin;
        const double lo = med - 0.25 * R, hi = med + 0.25 * R;
        double mu0[K];
#pragma unroll
        for (int k = 0; k < K; ++k) mu0[k] = lo + (hi - lo) * ((double)k / (double)(K - 1));
        mu0[K - 1] = hi;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            int best = 0;
            double bd = fabs(y[l] - mu0[0]);
#pragma unroll
            for (int k = 1; k < K; ++k) {
                const double d = fabs(y[l] - mu0[k]);
                if (d < bd) { bd = d; best = k; }
            }
            x[l] = valid[l] ? best : 0;
        }
    }

    // first state of the next thread's chunk (X at t0+L)
    int xnext = 0;
    sh.xfirst[tid] = x[0];
    if (tid == 0) sh.xfirst[NT] = 0;
    __syncthreads();
    xnext = sh.xfirst[tid + 1];

    // running sums behind `summary`
    constexpr int NSMAX = 3 * K + KK + 2 * HMCG_MAXH;
    const int NS = 3 * K + KK + 2 * p.H;
    const int orole = tid - (NT - 64);          // output roles live in the last wave
    double sum_acc = 0.0;
    if (orole >= 0 && orole < NS && p.resume && p.sumacc) sum_acc = p.sumacc[(size_t)w * NS + orole];
    (void)NSMAX;

    Rng rng{p.seed_lo, p.seed_hi, p.window_ids ? p.window_ids[w] : p.window_base + (uint32_t)w, 0u};

    // per-wave partial statistics of the current X (pass 1): counts, sums, transitions
    auto publish_stats = [&]() {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            int c = 0;
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const bool hit = valid[l] && x[l] == i;
                c += __popcll(__ballot(hit));
                s += hit ? y[l] : 0.0;
            }
            s = wave_sum(s);
            if (lane == 0) { sh.red_cnt[wave][i] = c; sh.red_sum[wave][i] = s; }
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                int c = 0;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int xn = (l + 1 < L) ? x[(l + 1 < L) ? l + 1 : l] : xnext;
                    const bool hit = (t0 + l + 1 < T) && x[l] == i && xn == j;
                    c += __popcll(__ballot(hit));
                }
                if (lane == 0) sh.red_cnt[wave][K + i * K + j] = c;
            }
    };
    publish_stats();

    double pf[L][K];     // unsorted filtered probabilities of this thread's steps

    for (int sweep = p.sweep_begin; sweep < p.sweep_end; ++sweep) {
        rng.sweep = (uint32_t)sweep;
        __syncthreads();                                                     // B0
        // ---- totals of pass 1 ----
        int Ni[K];
        double Si[K], ybar[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            int c = 0;
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { c += sh.red_cnt[ww][i]; s += sh.red_sum[ww][i]; }
            Ni[i] = c; Si[i] = s;
            ybar[i] = c > 0 ? s / (double)c : 0.0;                           // :259-265
        }
        // ---- pass 2: squared deviations about the state mean (:291-294) ----
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double s2 = 0.0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const double dlt = y[l] - ybar[i];
                s2 += (valid[l] && x[l] == i) ? dlt * dlt : 0.0;
            }
            s2 = wave_sum(s2);
            if (lane == 0) sh.red_sse[wave][i] = s2;
        }
        __syncthreads();                                                     // B1
        // ---- parameter draws on wave 0 (sites 0..3) ----
        if (wave == 0) {
            double val = 0.0;     // this lane's gamma variate
            const int role = lane;
            const bool is_sig = role < K, is_rho = role >= K && role < 2 * K, is_A = role >= 2 * K && role < ND;
            double shape = 1.0, bpar = 1.0, Neff = 0.0;
            uint32_t site = SITE_RHO, elem = 0;
            if (is_sig) {
                const int i = role;
                int c = 0; double s = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) if (k == i) { c = Ni[k]; s = Si[k]; }
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s2 += sh.red_sse[ww][i];
                Neff = (double)c;
                const double tb = c > 0 ? s / (double)c : 0.0;               // totalbar (:282-288, Mi=0)
                const double beta = (sweep == 0) ? 1.0 : 2.0;                // quirk 2 (:179, :347)
                const double dm = tb - xi;
                shape = p.alpha + 0.5 * Neff;                                // :313
                bpar = beta + 0.5 * s2 + 0.5 * Neff * p.nu / (Neff + p.nu) * (dm * dm);   // :314
                site = SITE_SIG2; elem = (uint32_t)i;
            } else if (is_rho) {
                site = SITE_RHO; elem = (uint32_t)(role - K);
            } else if (is_A) {
                const int e = role - 2 * K;
                int c = 0;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) c += sh.red_cnt[ww][K + e];
                shape = (double)(c + 1);                                     // :362-365
                site = SITE_A; elem = (uint32_t)e;
            }
            if (role < ND) val = gamma_draw(rng, site, elem, shape, st);
            // normalise the Dirichlet rows: sum the group's variates in element order
            const int gbase = is_A ? 2 * K + ((role - 2 * K) / K) * K : K;
            double gs = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) gs += __shfl(val, gbase + j, 64);
            const double ginv = 1.0 / gs;
            if (is_sig) {
                const int i = role;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) if (k == i) s = Si[k];
                const double sig2 = 1.0 / (val * (1.0 / bpar));              // :320 InverseGamma(a,b)
                const double m = (s + p.nu * xi) / (Neff + p.nu);            // :331
                const double sdev = sqrt(sig2 / (Neff + p.nu));              // :332
                uint32_t r[4];
                rng.block(SITE_MU, (uint32_t)i, 0, r);
                const double mu = m + sdev * box_muller(r);                  // :334
                const double sd = sqrt(sig2);                                // :381
                sh.th_mu[i] = mu; sh.th_sig2[i] = sig2;
                sh.th_isd[i] = 1.0 / sd; sh.th_coef[i] = INVSQRT2PI / sd;
            } else if (is_rho) {
                sh.th_rho[role - K] = val * ginv;                            // :355
            } else if (is_A) {
                const int e = role - 2 * K;
                sh.th_A[e / K][e % K] = val * ginv;                          // :367
            }
        }
        __syncthreads();                                                     // B2
        // ---- everyone: parameters to registers, label order (sortperm, :501) ----
        double mu[K], isd[K], coef[K], rho[K], A[K][K];
        int order[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            mu[i] = sh.th_mu[i]; isd[i] = sh.th_isd[i]; coef[i] = sh.th_coef[i]; rho[i] = sh.th_rho[i];
#pragma unroll
            for (int j = 0; j < K; ++j) A[i][j] = sh.th_A[i][j];
        }
        {
            int pos[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                int c = 0;
#pragma unroll
                for (int j = 0; j < K; ++j) c += (mu[j] < mu[i] || (mu[j] == mu[i] && j < i)) ? 1 : 0;
                pos[i] = c;
            }
#pragma unroll
            for (int q = 0; q < K; ++q) {
                int o = 0;
#pragma unroll
                for (int i = 0; i < K; ++i) o += (pos[i] == q) ? i : 0;
                order[q] = o;
            }
        }
        // ---- uniforms for the state draws (site 4, index t) ----
        double ux[L];
        if constexpr (L % 2 == 0) {
#pragma unroll
            for (int b = 0; b < L / 2; ++b) {
                uint32_t r[4];
                rng.block(SITE_X, 0, (uint32_t)(t0 / 2 + b), r);
                ux[2 * b] = u53(r[0], r[1]);
                ux[2 * b + 1] = u53(r[2], r[3]);
            }
        } else {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                uint32_t r[4];
                const uint32_t t = (uint32_t)(t0 + l);
                rng.block(SITE_X, 0, t >> 1, r);
                ux[l] = (t & 1u) ? u53(r[2], r[3]) : u53(r[0], r[1]);
            }
        }
        // ---- forward filter (:371-440) as a scan of M_t = A diag(f_t) ----
        double f[L][K];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            double fm = 0.0;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const double z = (y[l] - mu[s]) * isd[s];
                f[l][s] = exp(-(z * z) / 2.0) * coef[s];
                fm = fmax(fm, f[l][s]);
            }
            if (!(fm >= 1e-300)) {
                // every pdf underflowed: treat the observation as missing (f = 1) and flag the window
                if (valid[l]) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                for (int s = 0; s < K; ++s) f[l][s] = 1.0;
            } else {
                const int e = -ilogb(fm) - 1;        // exact power-of-two scaling of the step
#pragma unroll
                for (int s = 0; s < K; ++s) f[l][s] = ldexp(f[l][s], e);
            }
        }
        // local product Q = M_{t0} ... M_{t0+L-1} (identity for padded steps)
        double Q[KK];
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s) Q[r * K + s] = (r == s) ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (valid[l]) {
                double N[KK];
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc += Q[r * K + k] * A[k][s];
                        N[r * K + s] = acc * f[l][s];
                    }
#pragma unroll
                for (int i = 0; i < KK; ++i) Q[i] = N[i];
            }
        }
        rescale_pow2<KK>(Q);
        // Kogge-Stone inclusive scan over the wave (earlier lanes multiply on the left)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            double O[KK];
#pragma unroll
            for (int i = 0; i < KK; ++i) O[i] = __shfl_up(Q[i], d, 64);
            if (lane >= d) {
                double N[KK];
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc += O[r * K + k] * Q[k * K + s];
                        N[r * K + s] = acc;
                    }
#pragma unroll
                for (int i = 0; i < KK; ++i) Q[i] = N[i];
            }
            rescale_pow2<KK>(Q);
        }
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < KK; ++i) sh.wtot[wave][i] = Q[i];
        }
        __syncthreads();                                                     // B3
        // prefix vector: rho' * (totals of earlier waves) * (exclusive lane prefix)
        double av[K];
#pragma unroll
        for (int s = 0; s < K; ++s) av[s] = rho[s];
        for (int ww = 0; ww < wave; ++ww) {
            double nv[K];
#pragma unroll
            for (int s = 0; s < K; ++s) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < K; ++r) acc += av[r] * sh.wtot[ww][r * K + s];
                nv[s] = acc;
            }
            rescale_pow2<K>(nv);
#pragma unroll
            for (int s = 0; s < K; ++s) av[s] = nv[s];
        }
        {
            double E[KK];
#pragma unroll
            for (int i = 0; i < KK; ++i) E[i] = __shfl_up(Q[i], 1, 64);
            if (lane > 0) {
                double nv[K];
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < K; ++r) acc += av[r] * E[r * K + s];
                    nv[s] = acc;
                }
#pragma unroll
                for (int s = 0; s < K; ++s) av[s] = nv[s];
            }
            rescale_pow2<K>(av);
        }
        // replay the normalised recursion over this thread's steps (:413-432)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (valid[l]) {
                double nv[K], total = 0.0;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < K; ++r) acc += av[r] * A[r][s];
                    nv[s] = acc * f[l][s];
                    total += nv[s];
                }
                if (!(total > 0.0)) {
                    st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                    for (int s = 0; s < K; ++s) nv[s] = 1.0 / K;
                    total = 1.0;
                }
                const double inv = 1.0 / total;
#pragma unroll
                for (int s = 0; s < K; ++s) av[s] = nv[s] * inv;
            }
#pragma unroll
            for (int s = 0; s < K; ++s) pf[l][s] = av[s];
        }
#pragma unroll
        for (int s = 0; s < K; ++s) sh.pfirst[tid][s] = pf[0][s];
        // owner of the last step: pib[end,:] and the draw of X[T-1] in SORTED labels (:464)
        if (t0 <= T - 1 && T - 1 < t0 + L) {
            double pe[K];
            double ulast = 0.0;
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (t0 + l == T - 1) {
                    ulast = ux[l];
#pragma unroll
                    for (int s = 0; s < K; ++s) pe[s] = pf[l][s];
                }
            double cp = 0.0;
            int idx = 0;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                double pq = 0.0;
#pragma unroll
                for (int s = 0; s < K; ++s) pq = (order[q] == s) ? pe[s] : pq;
                cp += pq;
                if (q < K - 1) idx += (cp <= ulast) ? 1 : 0;
            }
#pragma unroll
            for (int s = 0; s < K; ++s) sh.pi_end[s] = pe[s];
            sh.xlast = idx;
        }
        __syncthreads();                                                     // B4
        // ---- per-draw outputs (last wave), d = index of the kept draw ----
        if (sweep >= p.keep_from && orole >= 0 && orole < NS) {
            const int d = sweep - p.keep_from;
            double smu[K], ssig[K], spe[K], sA[K][K];
#pragma unroll
            for (int q = 0; q < K; ++q) {
#pragma unroll
                for (int s = 0; s < K; ++s)
                    if (order[q] == s) { smu[q] = mu[s]; ssig[q] = sh.th_sig2[s]; spe[q] = sh.pi_end[s]; }
            }
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int q2 = 0; q2 < K; ++q2) {
                    double v = 0.0;
#pragma unroll
                    for (int i = 0; i < K; ++i)
#pragma unroll
                        for (int j = 0; j < K; ++j) v = (order[q] == i && order[q2] == j) ? A[i][j] : v;
                    sA[q][q2] = v;
                }
            double val = 0.0;
            double* dst = nullptr;
            const size_t nrun = (size_t)p.nrun;
            if (orole < K) {
#pragma unroll
                for (int q = 0; q < K; ++q) if (orole == q) val = smu[q];
                if (p.mu) dst = p.mu + nrun * ((size_t)orole + (size_t)K * w) + d;
            } else if (orole < 2 * K) {
                const int q0 = orole - K;
#pragma unroll
                for (int q = 0; q < K; ++q) if (q0 == q) val = ssig[q];
                if (p.sig2) dst = p.sig2 + nrun * ((size_t)q0 + (size_t)K * w) + d;
            } else if (orole < 3 * K) {
                const int q0 = orole - 2 * K;
#pragma unroll
                for (int q = 0; q < K; ++q) if (q0 == q) val = spe[q];
                if (p.pi_end) dst = p.pi_end + nrun * ((size_t)q0 + (size_t)K * w) + d;
            } else if (orole < 3 * K + KK) {
                const int e = orole - 3 * K;          // column-major: e = i + K*j
                const int i0 = e % K, j0 = e / K;
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int j = 0; j < K; ++j) if (i0 == i && j0 == j) val = sA[i][j];
                if (p.A) dst = p.A + nrun * ((size_t)e + (size_t)KK * w) + d;
            } else {
                const int e = orole - 3 * K - KK;     // 2h + {0: forecast, 1: error}
                const int h = e >> 1;
                const double fv = forecast_value<K>(smu, sA, spe, p.horizons[h]);
                const double yr = p.yreal ? p.yreal[(size_t)w * p.H + h] : __builtin_nan("");
                val = (e & 1) ? fv - yr : fv;
                if (p.fcast) dst = p.fcast + nrun * ((size_t)e + (size_t)(2 * p.H) * w) + d;
            }
            if (dst) *dst = val;
            sum_acc += round5(val);
        }
        // ---- backward sampling (:459-484) as a suffix scan of state maps ----
        const int xlast = sh.xlast;
        double pfn_last[K];
#pragma unroll
        for (int s = 0; s < K; ++s) pfn_last[s] = sh.pfirst[(tid + 1 < NT) ? tid + 1 : tid][s];
        uint32_t gmap[L];
        uint32_t G = map_identity<K>();
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const int t = t0 + l;
            uint32_t m = map_identity<K>();
            if (t == T - 1) {
                m = map_const<K>(xlast);
            } else if (t < T - 1) {
                m = 0;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    // p[r] = Pf[t+1,r,s] is proportional to pif[t,r] * A[r,s]; its sum over r equals
                    // the unsorted pif[t+1,s], which drives the eps() guard (:472, quirk 7)
                    double wr[K], tot = 0.0;
#pragma unroll
                    for (int r = 0; r < K; ++r) { wr[r] = pf[l][r] * A[r][s]; tot += wr[r]; }
                    const double guard = (l + 1 < L) ? pf[(l + 1 < L) ? l + 1 : l][s] : pfn_last[s];
                    int idx = 0;
                    if (guard > EPS64) {
                        const double thr = ux[l] * tot;
                        double cp = 0.0;
#pragma unroll
                        for (int r = 0; r < K - 1; ++r) { cp += wr[r]; idx += (cp <= thr) ? 1 : 0; }
                    } else {
                        double cp = 0.0;
#pragma unroll
                        for (int r = 0; r < K - 1; ++r) { cp += 1.0 / K; idx += (cp <= ux[l]) ? 1 : 0; }
                    }
                    m |= (uint32_t)idx << (4 * s);
                }
            }
            gmap[l] = m;
            G = map_compose<K>(m, G);      // G = g_{t0+l} o (g_{t0+l+1} o ...)
        }
        // inclusive suffix scan over lanes: H_lane = G_lane o G_{lane+1} o ... o G_63
        uint32_t Hm = G;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t O = __shfl_down(Hm, d, 64);
            if (lane + d < 64) Hm = map_compose<K>(Hm, O);
        }
        if (lane == 0) sh.wmap[wave] = Hm;
        __syncthreads();                                                     // B5
        uint32_t Rw = map_identity<K>();
        for (int ww = NW - 1; ww > wave; --ww) Rw = map_compose<K>(sh.wmap[ww], Rw);
        uint32_t Hx = __shfl_down(Hm, 1, 64);
        if (lane == 63) Hx = map_identity<K>();
        const uint32_t Sfx = map_compose<K>(Hx, Rw);   // everything after this thread's chunk
        int sin = map_apply(Sfx, 0);                   // constant map below T-1: evaluate anywhere
        xnext = sin;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            if (valid[l]) { sin = map_apply(gmap[l], sin); x[l] = sin; }
        }
        publish_stats();
    }

    // ---- epilogue: checkpoint / debug outputs ----
    if (p.xstate) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (valid[l]) p.xstate[(size_t)w * p.ldY + t0 + l] = (uint8_t)x[l];
    }
    if (p.x_final) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (valid[l]) p.x_final[(size_t)w * p.ldY + t0 + l] = x[l];
    }
    if (p.pif_final && p.sweep_end > p.sweep_begin) {
#pragma unroll
        for (int l = 0; l < L; ++l)
            if (valid[l])
#pragma unroll
                for (int s = 0; s < K; ++s) p.pif_final[((size_t)w * p.ldY + t0 + l) * K + s] = pf[l][s];
    }
    if (orole >= 0 && orole < NS) {
        if (p.sumacc) p.sumacc[(size_t)w * NS + orole] = sum_acc;
        if (p.summary && p.final_launch)
            p.summary[(size_t)w * NS + orole] = p.nrun > 0 ? sum_acc / (double)p.nrun : __builtin_nan("");
    }
    if (st) atomicOr(&p.status[w], st);
}

}  // namespace hmcg
