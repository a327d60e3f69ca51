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
//   * forward filter  = prefix scan of the (power-of-two scaled) K x K matrices
//     A*diag(f_t): per-thread product of L matrices, a DPP scan over the 64 lanes of a
//     wave (row_shr 1/2/4/8, row_bcast 15/31), wave totals through LDS, then a
//     per-thread replay of the normalised recursion from the scanned prefix vector;
//   * backward sampling = suffix scan of the random maps g_t : X_{t+1} -> X_t
//     (each map is the inverse-CDF draw for a fixed pre-drawn uniform), composed
//     as 4-bit-per-entry tables;
//   * sufficient statistics: one pass, pivoted on the state means, wave ballots for the
//     counts and DPP reductions for the sums.
//   * wave specialisation: wave 0 draws the parameters (short serial phase) while the
//     other waves, in its shadow, (a) write the previous sweep's per-draw outputs and
//     forecast, (b) generate this sweep's uniforms for the state draws, and (c) prepare
//     the state-independent parts of the NEXT sweep's parameter draws (Philox blocks,
//     Box-Muller normals, log-uniforms, the rho Dirichlet) -- the RNG is counter-based,
//     so none of that depends on the chain.
// Only per-draw outputs (3K + K^2 + 2H doubles) leave the chip per sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hmcg.h"
#include "round5.hpp"

namespace hmcg {

enum { SITE_SIG2 = 0, SITE_MU = 1, SITE_RHO = 2, SITE_A = 3, SITE_X = 4, SITE_NOISE = 5 };
constexpr int GAMMA_MAX_ATTEMPTS = 64;
constexpr double INVSQRT2PI = 0.3989422804014327;
constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double EPS64 = 2.220446049250313e-16;

struct KernelParams {
    const double* Y;        // [W][ldY]
    const int32_t* T;       // [W]
    const double* yreal;    // [W][H] or null
    int32_t ldY, W, H;
    int32_t sweep_begin;    // global index of first sweep of this launch
    int32_t sweep_end;      // one past the last sweep of this launch
    int32_t resume;         // 1: load chain from xstate, 0: makeParams init
    int32_t final_launch;   // 1: write summary
    int32_t horizons[HMCG_MAXH];
    uint32_t seed_lo, seed_hi, window_base;
    double alpha, nu;
    double* mu; double* sig2; double* A; double* pi_end; double* fcast; double* summary;
    int32_t* status;
    const int32_t* x_init; int32_t* x_final; double* pif_final; uint8_t* xstate; double* sumacc;
    const uint32_t* window_ids;
    unsigned long long* dbg;   // diagnostic (HMCG_STAMPS) builds only: per-wave phase cycle sums
    // sampling schedule: n_samples consecutive blocks of (burnin_s discarded + nrun_s kept) sweeps; kept draw
    // d of global sweep g is (g / per_sample) * nrun_s + (g % per_sample - burnin_s); nd = n_samples * nrun_s
    // is the leading dimension of the per-draw outputs.  n_samples == 1 is the estimatemodel case.
    int32_t per_sample, burnin_s, nrun_s, n_samples, nd;
    // The per-draw output arrays of THIS launch hold draws [draw_off, draw_off + nd_ld): leading dimension nd_ld, kept
    // draw d is stored at column index d - draw_off.  (nd_ld = nd, draw_off = 0 when one launch covers the whole run; the
    // host entry runs long chains in chunks whose outputs stream to the caller while the next chunk samples.)
    int32_t nd_ld, draw_off;
    // signal Monte-Carlo path (estimatesignals!, src/Hmc.jl:868-914); all null/zero for estimatemodel
    double kappa;                   // hp.kappa: relative noise of a signal observation
    const int32_t* sig_range;       // [W][2] signal positions [begin, end) (end == T)
    const int32_t* save_range;      // [W][2] positions whose noisy values are reported (signalSave)
    const double* sigma_signal;     // [W] sd of the noise added to the signal positions of each sample
    double* sigvals;                // [W][n_samples][nsave_ld]
    int32_t nsave_ld;
    // signals past the end date (sigLen = T-1-end_pos > 0, src/Hmc.jl:888): pi_end reports the SMOOTHED
    // probabilities at end_pos (:900); horizons in blend_mask go through forecastsignal (:908-909)
    const int32_t* end_pos;         // [W] 0-based, T-1-HMCG_MAXTAIL <= end_pos <= T-1; null = T-1
    int32_t blend_mask;
    // smoothed probabilities (backwardupdate_P!, src/Hmc.jl:442-457): running sum over the kept draws of
    // P(X_t | Y_1:T, theta) in SORTED labels, [W][ldY][K]; divided by nd at the final launch
    double* pi_smooth_mean;
    // the same running sum for the FILTERED probabilities pif[t,:] (sorted labels), [W][ldY][K]; SMOOTH variants
    double* pi_filter_mean;
    // every kept draw's smoothed probabilities (sorted labels), [W][K][ldY][nd_ld] with the draw index fastest = the
    // reference's samples.pib[Nrun, N, D] (:552,558); this launch's draws at column d - draw_off, as the other outputs
    double* pi_smooth_draws;
    // signal path: per noise sample the mean over its nrun_s kept draws of the rounded outputs, [W][n_samples][NS] (one row
    // of upstream's runaggregate over a signal run, src/Hmc.jl:1025-1057); the raw running sums while a sample is incomplete
    double* sample_summary;
    // LDS-resident kernel (gibbs_big.hpp) only: scratch that hands each step's K pdfs from the product phase to the replay,
    // [W][L][KP][NT][2] with KP = big_scratch_pairs(K) (a pair of values per thread, thread index next: coalesced 16-byte
    // accesses); library-owned
    double* fscr;
    // ... and its HBM-streaming variant (windows whose per-step state does not fit the CU's LDS): per window the observations,
    // the sweep's uniforms, the state maps and the states, [W][stream_stride] bytes; library-owned
    uint8_t* sscr;
    int64_t stream_stride;
    // host entry, first launch of a call: pinned HOST memory (device-addressable), one word per window, zeroed by the host --
    // a window the prologue skips puts its status bits here as well, so that the host knows whose block of a chunk buffer
    // holds nothing without a copy of the status words (a copy that small is a shader copy in the HIP runtime: it cannot
    // start while the sweep kernels hold every CU, and the chunk's SDMA copy queued behind it waited with it).  May be null.
    int32_t* skip_host;
    // length-bucketed dispatch (hmcg.hip, launch_kernel / bucket_lists_kernel): this launch runs the windows of ONE length class,
    // t_lo < T[w] <= t_hi; another launch of the same call, with the steps-per-thread variant that fits them, runs the others
    // beside it on its own stream.  order[0] = n, order[1] = t_lo, order[2] = t_hi, order[4..4+n) = the ids of the class's
    // windows, compacted: block b runs window order[4 + b] for b < n and leaves at once otherwise.  The live blocks of a
    // launch are then blocks 0..n-1 whatever the order of the caller's windows -- with the class's windows scattered over the
    // grid between blocks that leave, a shuffled production batch took 9.5 ms against 6.4 sorted
    // (profiles/r04/production_460_order.txt).  n < 0 (the lists found every class in one run of the caller's windows: a
    // batch sorted by length, as the reference's expanding windows are): block b runs window b if its length is in the
    // class and leaves otherwise.  Null: one launch for everything, block b runs window b.
    // (ONE kernel argument: more fields and a loop in the prologue cost the sweep loop 2 % through the register allocator.)
    const int32_t* order;
};

// a window the prologue refuses: its status bits, in HBM and (host entry) in the host's skip words
__device__ __forceinline__ void flag_skipped(const KernelParams& p, int w, int bits)
{
    atomicOr(&p.status[w], bits);
    if (p.skip_host) p.skip_host[w] = bits;
}

// pairs of pdf values per step in the scratch KernelParams::fscr
__host__ __device__ constexpr int big_scratch_pairs(int K) { return (K + 1) / 2; }

// bytes of one window's slab of KernelParams::sscr for `cap` = NT * L steps (observations | uniforms | maps | states + 8)
__host__ __device__ inline size_t stream_slab_bytes(size_t cap) { return (cap * (8 + 8 + 4) + cap + 8 + 255) & ~(size_t)255; }

// index of the kept draw produced by global sweep g, or -1 during burn-in
__device__ __forceinline__ int kept_index(const KernelParams& p, int g)
{
    const int smp = g / p.per_sample, i = g - smp * p.per_sample;
    return i >= p.burnin_s ? smp * p.nrun_s + (i - p.burnin_s) : -1;
}

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

// exp(x), x <= 0, with a table of 2^(j/N) held in LDS: x = (N e + j) ln2/N + r, |r| <= ln2/(2N), exp(r) by a
// Taylor polynomial, result 2^e * tab[j] * poly: ~1 ulp.  N = 64 with degree 5 (truncation 3.5e-17) or N = 256 with
// degree 4 (3.8e-17).
// The table is REPLICATED: EXPTAB_C copies, entry j of copy c at tab[j * EXPTAB_C + c], lane l reading copy l % EXPTAB_C.
// Every lane looks up its own random j: out of ONE copy the 64 lanes of a ds_read_b64 collide ~6 deep on the 64 banks, and
// all four waves of a window evaluate their pdfs at the same time -- the phase ran at 1.65x its issue floor on the LDS
// pipe (round 4, tools/phase_table.py).  With 32 copies of a 64-entry table (16 KB) copy c owns banks 2c, 2c+1 whatever
// j is, and lanes l, l + 32 go through the pipe in different halves: no conflict at all.  (256 entries x 32 copies would
// be 64 KB; the 64-entry table costs one FMA more per exponential.)  A translation unit may define HMCG_EXPTAB_N /
// HMCG_EXPTAB_COPIES ahead of this header (A/B builds only: every shipped kernel uses the same table).
#ifndef HMCG_EXPTAB_N
#define HMCG_EXPTAB_N 64
#endif
#ifndef HMCG_EXPTAB_COPIES
#define HMCG_EXPTAB_COPIES 32
#endif
constexpr int EXPTAB_N = HMCG_EXPTAB_N, EXPTAB_C = HMCG_EXPTAB_COPIES;
static_assert(EXPTAB_N == 64 || EXPTAB_N == 256, "exp table: 64 or 256 entries");
static_assert(EXPTAB_C >= 1 && EXPTAB_C <= 64 && (EXPTAB_C & (EXPTAB_C - 1)) == 0, "copies: a power of two");
// fills tab[EXPTAB_N * EXPTAB_C] (all threads of the block; the caller's next barrier publishes it)
__device__ __forceinline__ void exptab_fill(double* tab, int tid, int nthreads)
{
    for (int i = tid; i < EXPTAB_N * EXPTAB_C; i += nthreads)
        tab[i] = exp2((double)(i / EXPTAB_C) * (1.0 / EXPTAB_N));      // correctly rounded enough (OCML exp2, < 1 ulp)
}
// the table entry of exponent index ni for a lane whose copy is c = lane % EXPTAB_C (byte address = entry << log2(8 C) | 8 c:
// one v_and and one v_lshl_or)
__device__ __forceinline__ double exptab_at(const double* tab, int ni, int c)
{
    const unsigned off = ((unsigned)(ni & (EXPTAB_N - 1)) * (unsigned)(8 * EXPTAB_C)) | ((unsigned)c * 8u);
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + off);
}
// MAGIC: n = x N/ln2 rounded to an integer by ONE fused add of 1.5 * 2^52 -- the integer is then the low word of the sum as
// it stands (no v_rndne, no v_cvt) and n one subtraction away.  One instruction less per value where the table address
// costs the same either way: +0.7 % on the K = 8 kernel (the LDS-resident kernels use it), -0.4 % on the K = 3 headline
// kernel, whose compiler-chosen address arithmetic grows by one instruction (the register-resident kernels do not).
constexpr double EXP_MAGIC = 6755399441055744.0;      // 1.5 * 2^52: x + EXP_MAGIC holds round(x) in its low word, |x| < 2^31
template <bool MAGIC = false>
__device__ __forceinline__ double exp_tab(double x, const double* tab, int c)
{
    x = fmax(x, -746.0);
    double n, r, p;
    int ni;
    if constexpr (MAGIC) {
        const double t2 = fma(x, EXPTAB_N == 64 ? 92.332482616893657 : 369.3299304675746, EXP_MAGIC);
        ni = __double2loint(t2);
        n = t2 - EXP_MAGIC;
    } else {
        n = rint(x * (EXPTAB_N == 64 ? 92.332482616893657 : 369.3299304675746));      // N / ln 2
        ni = (int)n;
    }
    if constexpr (EXPTAB_N == 64) {
        r = fma(-n, 1.0830424693267560e-02, x);               // ln2/64 high part
        r = fma(-n, 2.9815858269852933e-12, r);               // ln2/64 low part
        p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
        p = fma(p, r, 1.0 / 6.0);
    } else {
        r = fma(-n, 2.70760617331689e-03, x);                 // ln2/256 high part
        r = fma(-n, 7.453964567463233e-13, r);                // ln2/256 low part
        p = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    }
    const double tj = exptab_at(tab, ni, c);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(tj * p, ni >> (EXPTAB_N == 64 ? 6 : 8));
}

// 1/x: hardware reciprocal + one Newton step.  v_rcp_f64 / v_rsq_f64 are 2^-24-grade seeds on gfx950 (measured:
// tools/ubench/rcp_prec.hip, max relative error 4.5e-8 / 5.2e-8), so one step leaves ~2e-15 relative error -- the
// source of the ~1e-14 GPU-vs-oracle differences the parity tests observe, five orders inside their 1e-9 tolerance.
// (sqrt_fast / rsqrt_fast below take two steps and are accurate to the last bit or two.)
__device__ __forceinline__ double rcp_fast(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

// log(x) for positive normal x (fdlibm-style: x = 2^e * m, m in [sqrt(.5), sqrt(2)), atanh series), < 1 ulp
__device__ __forceinline__ double log_fast(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * rcp_fast(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// cos(2*pi*u) for u in [0,1): exact reduction to a quadrant, fdlibm kernels on [-pi/4, pi/4]
__device__ __forceinline__ double cos2pi_fast(double u)
{
    const double k = rint(4.0 * u);                       // 0..4
    const double r = u - 0.25 * k;                        // exact, |r| <= 1/8
    const double phi = TWO_PI * r;
    const double z = phi * phi;
    const int q = (int)k & 3;
    // cos kernel
    const double c = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                     -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03),
                     4.16666666666666019037e-02), -0.5);
    const double cosv = fma(z, c, 1.0);
    // sin kernel
    const double sp = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                      2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                      -1.66666666666666324348e-01);
    const double sinv = fma(phi * z, sp, phi);
    // cos(k*pi/2 + phi): k=0 cos, 1 -sin, 2 -cos, 3 sin
    const double v = (q & 1) ? sinv : cosv;
    return (q == 1 || q == 2) ? -v : v;
}

__device__ __forceinline__ double box_muller(const uint32_t (&r)[4])
{
    const double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    return sqrt(-2.0 * log_fast(1.0 - u1)) * cos2pi_fast(u2);
}

// Marsaglia-Tsang acceptance for one attempt: returns the variate or a negative number.  `u` is the uniform behind
// logu = log(1 - u).  The squeeze 1 - u < 1 - 0.0331 x^4 implies the exact test (Marsaglia & Tsang 2000, step 3: for
// d >= 1/3 the bound holds with room to spare), so it decides identically and skips the logarithm; the exact test runs
// only when some active lane fails the squeeze (a wave-uniform branch, about one parameter phase in five).
__device__ __forceinline__ double mt_try(double d, double c, double x, double logu, double u)
{
    const double v1 = 1.0 + c * x;
    const double v = v1 * v1 * v1;
    const double x2 = x * x;
    bool ok = (v1 > 0.0) && (u > 0.0331 * (x2 * x2));
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull)
        ok = ok || ((v1 > 0.0) && (logu < 0.5 * x2 + d - d * v + d * log_fast(v1 > 0.0 ? v : 1.0)));
    return ok ? d * v : -1.0;
}

// ------------------------------------------------------------- math ------

// exp(x) for x <= ~0 (emission pdfs): Cody-Waite reduction by ln2 and a degree-13 Taylor
// polynomial on |r| <= ln2/2 (truncation < 2e-17), scaled by ldexp (gradual underflow).
__device__ __forceinline__ double exp_fast(double x)
{
    x = fmax(x, -746.0);                       // exp(-746) rounds to 0 in binary64
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// ------------------------------------------------------- DPP helpers -----
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// lanes with no source (row edge, masked row) receive `old`
template <int CTRL, int RMASK>
__device__ __forceinline__ int dpp_i32(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, RMASK, 0xF, false);
}
template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_f64(double old, double v)
{
    const int lo = dpp_i32<CTRL, RMASK>(__double2loint(old), __double2loint(v));
    const int hi = dpp_i32<CTRL, RMASK>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// full-row-mask variant whose source-less lanes receive 0.0 (bound_ctrl zero fill: no `old` register to initialise)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_zero(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// ... and 1.0 (only the high word needs an `old`)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_one(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// wave-wide sum, fixed association order (deterministic), broadcast to every lane
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<DPP_ROW_SHR1, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_SHR2, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_SHR4, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_SHR8, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xA>(0.0, v);
    v += dpp_f64<DPP_ROW_BCAST31, 0xC>(0.0, v);
    return readlane_f64(v, 63);
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// v(lane) + v(lane ^ 32) in every lane (v_permlane32_swap: upper half of one operand <-> lower half of the other)
__device__ __forceinline__ double xor32_sum(double v)
{
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// v(lane) + v(lane ^ 16) in every lane (v_permlane16_swap: odd 16-lane rows of one operand <-> even rows of the other)
__device__ __forceinline__ double xor16_sum(double v)
{
    const u32x2_t lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL>
__device__ __forceinline__ double quadperm_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Wave-wide sums of 8 per-lane quantities in ~1/3 of the instructions of 8 separate reductions:
// a butterfly that halves the number of live quantities at each of the first two levels
// (xor 1, xor 2 inside a quad), then sums the 16 quads (row_shr 4, 8; xor 16; xor 32).
// On return the lanes with (lane & 12) == 12 hold, in (o0, o1), the totals of slots
// 4*(lane&1) + (lane&2) + {0, 1}.  Fixed association order: deterministic.
__device__ __forceinline__ void wave_sum8_transposed(const double (&v)[8], int lane, double& o0, double& o1)
{
    const bool p = (lane & 1) != 0, p2 = (lane & 2) != 0;
    double a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = p ? v[4 + i] : v[i];
        const double send = p ? v[i] : v[4 + i];
        a[i] = keep + quadperm_f64<0xB1>(send);          // quad_perm:[1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = p2 ? a[2 + i] : a[i];
        const double send = p2 ? a[i] : a[2 + i];
        b[i] = keep + quadperm_f64<0x4E>(send);          // quad_perm:[2,3,0,1]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        b[i] += dpp_f64<DPP_ROW_SHR4, 0xF>(0.0, b[i]);
        b[i] += dpp_f64<DPP_ROW_SHR8, 0xF>(0.0, b[i]);
        b[i] = xor16_sum(b[i]);
        b[i] = xor32_sum(b[i]);
    }
    o0 = b[0]; o1 = b[1];
}

// wave-wide sum of packed non-negative integer fields (no field may overflow into its neighbour);
// one v_add_u32_dpp per level; the total lands in lane 63
__device__ __forceinline__ unsigned wave_sum_u32_lane63(unsigned v)
{
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}

// the same for N words at once: the N independent chains are interleaved in one block, so a word's next level is
// N - 1 instructions behind its previous one and the DPP read-after-write wait (two states) needs no s_nop from N = 3 on
#define HMCG_DPP_ADD3(CTRL) \
    "v_add_u32_dpp %0, %0, %0 " CTRL "\n\tv_add_u32_dpp %1, %1, %1 " CTRL "\n\tv_add_u32_dpp %2, %2, %2 " CTRL "\n\t"
#define HMCG_DPP_ADD2(CTRL) "v_add_u32_dpp %0, %0, %0 " CTRL "\n\tv_add_u32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
template <int N>
__device__ __forceinline__ void wave_sum_words_lane63(unsigned (&v)[N])
{
    if constexpr (N == 3) {
        asm volatile("s_nop 1\n\t" HMCG_DPP_ADD3("row_shr:1 row_mask:0xf bank_mask:0xf") HMCG_DPP_ADD3("row_shr:2 row_mask:0xf bank_mask:0xf")
                         HMCG_DPP_ADD3("row_shr:4 row_mask:0xf bank_mask:0xf") HMCG_DPP_ADD3("row_shr:8 row_mask:0xf bank_mask:0xf")
                             HMCG_DPP_ADD3("row_bcast:15 row_mask:0xa bank_mask:0xf") HMCG_DPP_ADD3("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    } else if constexpr (N == 2) {
        asm volatile("s_nop 1\n\t" HMCG_DPP_ADD2("row_shr:1 row_mask:0xf bank_mask:0xf") HMCG_DPP_ADD2("row_shr:2 row_mask:0xf bank_mask:0xf")
                         HMCG_DPP_ADD2("row_shr:4 row_mask:0xf bank_mask:0xf") HMCG_DPP_ADD2("row_shr:8 row_mask:0xf bank_mask:0xf")
                             HMCG_DPP_ADD2("row_bcast:15 row_mask:0xa bank_mask:0xf") HMCG_DPP_ADD2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     : "+v"(v[0]), "+v"(v[1]));
    } else {
#pragma unroll
        for (int d = 0; d < N; ++d) v[d] = wave_sum_u32_lane63(v[d]);
    }
}
#undef HMCG_DPP_ADD3
#undef HMCG_DPP_ADD2

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

// Scale every (non-negative) entry by the power of two that brings the largest one into [0.5,1).
// Exact (no rounding) away from the subnormal range; a zero matrix stays zero.  The largest
// exponent is found on the high words as integers (entries are >= 0, so the order is the same).
template <int N>
__device__ __forceinline__ void rescale_pow2(double (&q)[N])
{
    unsigned m = (unsigned)__double2hiint(q[0]);
#pragma unroll
    for (int i = 1; i < N; ++i) m = max(m, (unsigned)__double2hiint(q[i]));
    const int be = (int)(m >> 20);                       // biased exponent of the largest entry
    int e = (be > 0 && be < 2040) ? 1022 - be : 0;
    asm volatile("" : "+v"(e));                          // one select on the exponent -- not one per entry on ldexp's results
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = ldexp(q[i], e);
}

// sqrt(x), x > 0 normal: hardware rsq + two Newton steps on the product form (last-bit accurate)
__device__ __forceinline__ double sqrt_fast(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);                   // refine 1/sqrt(x)
    double g = x * r;
    return fma(fma(-g, g, x), 0.5 * r, g);
}

// 1/sqrt(x), x > 0 normal: hardware rsq + two Newton steps (last-bit accurate)
__device__ __forceinline__ double rsqrt_fast(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return fma(0.5 * r, fma(-x * r, r, 1.0), r);
}

// one level of the wave-wide inclusive matrix scan: Q <- (Q of the source lane) * Q
template <int K, int CTRL, int RMASK>
__device__ __forceinline__ void scan_level(double (&Q)[K * K])
{
    double O[K * K], N[K * K];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if constexpr (RMASK == 0xF) O[r * K + s] = (r == s) ? dpp_f64_one<CTRL>(Q[r * K + s]) : dpp_f64_zero<CTRL>(Q[r * K + s]);
            else O[r * K + s] = dpp_f64<CTRL, RMASK>((r == s) ? 1.0 : 0.0, Q[r * K + s]);
        }
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            double acc = O[r * K] * Q[s];
#pragma unroll
            for (int k = 1; k < K; ++k) acc = fma(O[r * K + k], Q[k * K + s], acc);
            N[r * K + s] = acc;
        }
#pragma unroll
    for (int i = 0; i < K * K; ++i) Q[i] = N[i];
}

// one level of the inclusive scan with the product order flipped: Q <- Q * (Q of the source lane)
template <int K, int CTRL, int RMASK>
__device__ __forceinline__ void scan_level_rev(double (&Q)[K * K])
{
    double O[K * K], N[K * K];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if constexpr (RMASK == 0xF) O[r * K + s] = (r == s) ? dpp_f64_one<CTRL>(Q[r * K + s]) : dpp_f64_zero<CTRL>(Q[r * K + s]);
            else O[r * K + s] = dpp_f64<CTRL, RMASK>((r == s) ? 1.0 : 0.0, Q[r * K + s]);
        }
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            double acc = Q[r * K] * O[s];
#pragma unroll
            for (int k = 1; k < K; ++k) acc = fma(Q[r * K + k], O[k * K + s], acc);
            N[r * K + s] = acc;
        }
#pragma unroll
    for (int i = 0; i < K * K; ++i) Q[i] = N[i];
}

// Same, for large K: one source row at a time (K doubles live instead of K*K); the result goes to N (the caller
// ping-pongs between two matrices from level to level instead of copying back)
template <int K, int CTRL, int RMASK>
__device__ __forceinline__ void scan_level_rowwise(const double (&Q)[K * K], double (&N)[K * K])
{
#pragma unroll
    for (int r = 0; r < K; ++r) {
        double o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (RMASK == 0xF) o[k] = (r == k) ? dpp_f64_one<CTRL>(Q[r * K + k]) : dpp_f64_zero<CTRL>(Q[r * K + k]);
            else o[k] = dpp_f64<CTRL, RMASK>((r == k) ? 1.0 : 0.0, Q[r * K + k]);
        }
#pragma unroll
        for (int s = 0; s < K; ++s) {
            double acc = o[0] * Q[s];
#pragma unroll
            for (int k = 1; k < K; ++k) acc = fma(o[k], Q[k * K + s], acc);
            N[r * K + s] = acc;
        }
    }
}

// The flipped order for large K: N = Q * (Q of the source lane), one source COLUMN at a time
template <int K, int CTRL, int RMASK>
__device__ __forceinline__ void scan_level_colwise(const double (&Q)[K * K], double (&N)[K * K])
{
#pragma unroll
    for (int s = 0; s < K; ++s) {
        double o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (RMASK == 0xF) o[k] = (s == k) ? dpp_f64_one<CTRL>(Q[k * K + s]) : dpp_f64_zero<CTRL>(Q[k * K + s]);
            else o[k] = dpp_f64<CTRL, RMASK>((s == k) ? 1.0 : 0.0, Q[k * K + s]);
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            double acc = Q[r * K] * o[0];
#pragma unroll
            for (int k = 1; k < K; ++k) acc = fma(Q[r * K + k], o[k], acc);
            N[r * K + s] = acc;
        }
    }
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

// Byte-per-entry state maps for K <= 4 (the register-resident kernel): entry s of map m is byte s.  Composition is ONE
// instruction -- v_perm_b32 selects, for every byte of b (a state 0..3), that byte of a: (a o b)[s] = a[b[s]].
// Unused high bytes carry the identity so that they stay valid selectors.
constexpr uint32_t BMAP_IDENTITY = 0x03020100u;
__device__ __forceinline__ uint32_t bmap_const(int v) { return (uint32_t)v * 0x01010101u; }
__device__ __forceinline__ uint32_t bmap_compose(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(a, a, b); }
__device__ __forceinline__ int bmap_apply(uint32_t m, int s) { return (int)((m >> (8 * s)) & 0xFFu); }
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108, DPP_WAVE_SHL1 = 0x130;

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

// (pi' A^h) . mu as pi' (A^h mu): binary exponentiation carrying the vector, log2(h) squarings
template <int K>
__device__ inline double forecast_value(const double (&mu)[K], const double (&A)[K][K], const double (&pe)[K], int h)
{
    double M[K][K], v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        v[i] = mu[i];
#pragma unroll
        for (int j = 0; j < K; ++j) M[i][j] = A[i][j];
    }
    for (unsigned hh = (unsigned)h; hh != 0; hh >>= 1) {        // h differs between horizon lane pairs: per-lane trip count
        if (hh & 1u) {
            double nv[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) acc = fma(M[i][j], v[j], acc);
                nv[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = nv[i];
        }
        if (hh > 1u) matmul_small<K>(M, M, M);
    }
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) f = fma(pe[i], v[i], f);
    return f;
}

// ---------------------------------------------------------------- LDS ----

template <int K>
struct ThetaBuf {                 // one sweep's parameters, UNSORTED labels
    double mu[K], sig2[K], isd[K], coef[K], rho[K];
    double A[K][K];
    double pi_end[K];             // unsorted pif[T-1,:] of that sweep
};

template <int K>
struct RngBuf {                   // state-independent parts of one sweep's parameter draws
    static constexpr int NG = K + K * K;      // gamma roles: sig2_i (K) then A_ij (K*K, row-major)
    double x[NG][2];              // Box-Muller normal of attempts 0,1
    double lu[NG][2];             // log(1-u) of attempts 0,1
    double uu[NG][2];             // ... and u itself (Marsaglia-Tsang squeeze)
    double z[K];                  // normals for mu_i
    double rho[K];                // the complete rho ~ Dirichlet(1) draw
};

template <int K, int L, int NT, bool SIG>
struct SweepShared {
    static constexpr int NW = NT / 64;
    static constexpr int NCNT = K + K * K;
    static constexpr int NF = K * K + (SIG ? K : 0);     // count fields: C_ij, then (signal path) M_i = signal steps in state i
    static constexpr int FW = (64 * L < 1024) ? 10 : 16;  // bits per wave-total field: a wave total is at most 64 L
    static constexpr int FPK = 32 / FW;                  // fields per word: three 10-bit ones while L <= 15, else two
    static constexpr int NPK = (NF + FPK - 1) / FPK;
    unsigned red_pk[NW][NPK];     // per-wave counts, FPK fields per word (field e = i*K+j; K*K+i for M_i)
    double red_s1[SIG ? NW : 1][K];   // signal path: the same pivoted sums over the signal positions
    double red_s2[SIG ? NW : 1][K];
    int x_end;                    // X[T-1] of the chain state the statistics describe
    double red_d1[NW][K];         // per-wave sums of (y - pivot_i) by state
    double red_d2[NW][K];         // per-wave sums of (y - pivot_i)^2 by state
    double pivot[K];              // pivots the partial sums above were taken about
    ThetaBuf<K> th[2];            // parity = sweep & 1
    RngBuf<K> rb[2];
    double wtot[NW][K * K];       // forward scan: wave totals
    double pfirst[NT + 1][K];     // filtered probs at each thread's first step
    uint32_t wmap[NW];            // backward scan: wave totals
    double ulast;                 // the uniform that draws X[T-1]
    int xfirst[NT + 1];           // init only: first state of each thread's chunk
    double bred[NW];              // generic block reductions (init)
    int selcnt[2][8][3];          // block_select (init): by round parity, wave, digit
    double ux[NT * L];            // this sweep's uniforms for the state draws (init: Y staged for the median)
    double exptab[EXPTAB_N * EXPTAB_C];   // 2^(j/N), j = 0..N-1, EXPTAB_C copies (entry j of copy c at [j * EXPTAB_C + c])
    // decoded output role of every lane of the (at most two) output waves, staged once per launch: slot 0 = the
    // parameter-output wave, slot 1 = the forecast wave when it is a different one
    struct OutConst { double* base; double yr; int packed; int h; };
    OutConst oc[2][64];
    // signal path with signals past the end date: the (scaled) emission values of the last `tail` steps, the
    // filtered probabilities at end_pos, and the current noise sample's last observation (by sample parity)
    double ftail[2][SIG ? HMCG_MAXTAIL : 1][K];      // by sweep parity: the outputs job of sweep s-1 runs beside the pdf phase of sweep s
    double pf_rep[2][K];
    double y_last[2];
};

template <int NW>
__device__ __forceinline__ double block_sum(double v, double* bred, int wave, int lane)
{
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0 && wave < NW) bred[wave] = v;      // helper waves (wave >= NW) only keep the barriers
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
    if (lane == 0 && wave < NW) bred[wave] = v;
    __syncthreads();
    double t = bred[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = is_max ? fmax(t, bred[w]) : fmin(t, bred[w]);
    return t;
}

// ---- the window's median without a sort: binary radix selection on order-preserving integer keys ----
// order_key: unsigned integers whose order is the doubles' order (finite values; -0.0 is folded into +0.0 first, so that
// equal doubles have equal keys); key_value is its inverse.
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const long long b = __double_as_longlong(v + 0.0);
    return (unsigned long long)(b ^ ((b >> 63) | (long long)0x8000000000000000ull));
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k));
}
// The key of rank r (0-based) among the block's live elements: 32 rounds, two key bits each, most significant first.  A
// round counts, among the elements that agree with the bits chosen so far, those whose next two bits are 00, 01, 10 (ballot
// and popcount per wave, three LDS words per wave, one barrier); the rank falls into one of the four digit classes, whose
// digit is appended and whose predecessors' counts leave the rank.  `elem(i, key)` hands out the thread's i-th key and
// whether it is live.  Every wave of the block must call it (barriers); waves >= NW have no elements and only keep the
// barriers.  cnt: LDS, [2][8][3] ints (by round parity, wave, digit).
// (Round 4: replaces counting every element's rank against every other -- T^2 / NT compares per thread, 0.18 ms of a
//  fresh launch at T = 1000 and 13 ms at T = 5000.  The median VALUE is the same: exact.)
template <int NW, int NE, typename ElemFn>          // NE > 0: that many elements per thread (unrolled); 0: n of them
__device__ __forceinline__ unsigned long long block_select(int n, ElemFn elem, int r, int (*cnt)[8][3], int wave, int lane)
{
    unsigned long long prefix = 0ull, mask = 0ull;
    const bool mine = __builtin_amdgcn_readfirstlane(wave) < NW;
    for (int b = 62; b >= 0; b -= 2) {
        const int par = (b >> 1) & 1;
        if (mine) {
            int c0 = 0, c1 = 0, c2 = 0;
            auto one = [&](int i) __attribute__((always_inline)) {
                unsigned long long key;
                const bool live = elem(i, key);
                const bool cand = live && ((key ^ prefix) & mask) == 0ull;
                const unsigned digit = (unsigned)(key >> b) & 3u;
                c0 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand && digit == 0u));
                c1 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand && digit == 1u));
                c2 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand && digit == 2u));
            };
            if constexpr (NE > 0) {
#pragma unroll
                for (int i = 0; i < NE; ++i) one(i);
            } else {
                for (int i = 0; i < n; ++i) one(i);
            }
            if (lane == 0) { cnt[par][wave][0] = c0; cnt[par][wave][1] = c1; cnt[par][wave][2] = c2; }
        }
        __syncthreads();
        int t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t0 += cnt[par][w][0]; t1 += cnt[par][w][1]; t2 += cnt[par][w][2]; }
        t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1); t2 = __builtin_amdgcn_readfirstlane(t2);
        unsigned long long digit = 0ull;
        if (r >= t0) { r -= t0; digit = 1ull;
            if (r >= t1) { r -= t1; digit = 2ull;
                if (r >= t2) { r -= t2; digit = 3ull; } } }
        prefix |= digit << b;
        mask |= 3ull << b;
    }
    return prefix;
}

// sortperm(mu) (src/Hmc.jl:501): stable rank of each entry
template <int K>
__device__ __forceinline__ void sort_order(const double (&mu)[K], int (&order)[K])
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

// In-kernel phase stamps: compiled only into the diagnostic build (-DHMCG_STAMPS, see
// csrc/Makefile `stamps`); the shipped kernel executes none.  Values leave the kernel only
// through p.dbg, which no other code reads.
#ifdef HMCG_STAMPS
#define HMCG_NSTAMP 20
#define HMCG_NSTAMP_ALL (HMCG_NSTAMP + 2)   // + the sweep loop's total in s_memtime ticks and in s_memrealtime (100 MHz) ticks
#define STAMP(i)                                                          \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        asm volatile("; HMCG_STAMP_MARK %0" :: "n"(i));                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
        __builtin_amdgcn_s_waitcnt(0xC07F);                               \
        __builtin_amdgcn_sched_barrier(0);                                \
        stamp_acc[i] += t_ - stamp_prev;                                  \
        stamp_prev = t_;                                                  \
    } while (0)
#else
#define STAMP(i)
#endif

// ------------------------------------------------------------- kernel ----

// NH > 0 adds NH "helper" waves (threads NT .. NT+64*NH-1) that own no time steps: they carry the per-draw
// outputs, the forecasts, the next sweep's RNG preparation and a share of the Philox uniforms while wave 0
// draws the parameters, sharing the SIMDs' issue slots with the primary waves (one helper per SIMD).
// OCC = 2 caps the registers so that two blocks of a plain 256-thread variant fit one CU (for batches with more
// windows than CUs); OCC = 1 gives a lone block the whole register file.  (A helped block is eight waves, two per
// SIMD, by itself.)
template <int K, int L, int NT, bool SIG = false, bool SMOOTH = false, int NH = 0, int OCC = 1>
__global__ __launch_bounds__(NT + 64 * NH) __attribute__((amdgpu_waves_per_eu(OCC)))
void gibbs_sweeps_kernel(const KernelParams p)
{
    static_assert(NH == 0 || (NH == 4 && NT == 256), "helper waves: 256 + 256 threads");
    static_assert(K >= 2 && K <= 4, "small-K kernel: byte-per-entry state maps (v_perm_b32 composition)");
    static_assert(NT % 64 == 0 && NT >= 128, "at least two whole waves (wave 0 draws, the others work in its shadow)");
    constexpr int NW = NT / 64;
    constexpr int KK = K * K;
    constexpr int NG = K + KK;           // gamma roles: sig2_i, then A_ij row-major
    static_assert(NG <= 64, "draw roles must fit wave 0");
    using Sh = SweepShared<K, L, NT, SIG>;
    __shared__ Sh sh;

    int w_ = blockIdx.x;
    int t_lo_ = INT32_MIN, t_hi_ = INT32_MAX;
    if (p.order) {
        const int n_live = p.order[0];
        t_lo_ = p.order[1]; t_hi_ = p.order[2];
        if (n_live >= 0) {                           // (uniform) some class is scattered over the grid: run the compacted list
            if (w_ >= n_live) __builtin_amdgcn_endpgm();
            w_ = p.order[4 + w_];
        }
    }
    const int w = w_;
#ifdef HMCG_VECTOR_WAVE                                     // (A/B only: the wave id as a per-lane value, as until round 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#else
    // the wave id is a SCALAR: everything that depends on it branches (s_cbranch_scc) instead of opening an exec-masked
    // region -- fewer lane masks parked in SGPRs, fewer join blocks (the places the allocator fault of DESIGN 5a strikes)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int T = p.T[w];
    const int t0 = tid * L;                                // helper threads: t0 >= NT*L >= T, so they own no step
    const int owner = (T - 1) / L, l_last = (T - 1) % L;   // thread and slot that hold the last time step
    const bool helper = NH > 0 && __builtin_amdgcn_readfirstlane(wave) >= NW;   // wave-uniform
    int st = 0;

    if (T <= t_lo_ || T > t_hi_) __builtin_amdgcn_endpgm();   // another length bucket's window: this block leaves at once
    if (T < 2 || T > NT * L || T > p.ldY) {   // uniform per block
        if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_T);
        return;
    }

    exptab_fill(sh.exptab, tid, NT + 64 * NH);
    // ---- load the window's observations (once per launch) ----
    // Steps at or beyond T ("padded": t0 + l >= T) carry the pseudo-state XPAD.  For K <= 3 that is the out-of-range value K,
    // which no `x == i` test matches and which the backward pass keeps in place by itself (the maps beyond T-1 are the
    // constant map to K): the statistics then need no `t < T` predicates -- eight lane masks that used to live in SGPRs
    // across the sweep loop and came back through v_readlane.  (K = 4 fills the byte-map selector range, so it keeps 0
    // and the predicates.)
    constexpr int XPAD = K < 4 ? K : 0;
    constexpr bool PADMARK = K < 4;
    double y[L];
    int x[L];
    bool bad = false;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const bool v = (t0 + l) < T;
        y[l] = v ? p.Y[(size_t)w * p.ldY + t0 + l] : 0.0;
        bad |= v && !isfinite(y[l]);
        x[l] = XPAD;
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (tid == 0) flag_skipped(p, w, HMCG_ST_NONFINITE);
        return;
    }
    // signal path: positions [sb, se) are "signals" (noisy observations); y[] then holds the current
    // noise sample's Yfake and yreal[] the data
    int sb = T, se = T;
    double yreal[SIG ? L : 1];
    const double kfac = SIG ? 1.0 / (1.0 + p.kappa) : 1.0;
    if constexpr (SIG) {
        if (p.sig_range) { sb = p.sig_range[2 * w]; se = p.sig_range[2 * w + 1]; }
        // caller data: ranges must lie inside the window and the saved range must fit its slab (uniform per block)
        bool bad_range = sb < 0 || se > T || (sb < se && se != T);
        if (p.save_range) {
            const int svb = p.save_range[2 * w], sve = p.save_range[2 * w + 1];
            bad_range |= svb < 0 || sve > T || (svb < sve && p.sigvals && sve - svb > p.nsave_ld);
        }
        if (bad_range) {
            if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_RANGE);
            return;
        }
        if (sb >= se) { sb = T; se = T; }
#pragma unroll
        for (int l = 0; l < L; ++l) yreal[l] = y[l];
    }
    int tail = 0;                         // steps after the position whose smoothed probabilities are reported
    if constexpr (SIG) {
        if (p.end_pos) tail = (T - 1) - p.end_pos[w];
        if (tail < 0 || tail > HMCG_MAXTAIL || tail > T - 1) {      // uniform per block
            if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_T);
            return;
        }
    }
    (void)kfac; (void)yreal; (void)sb; (void)se; (void)tail;

    // ---- HyperParams(Y,D): xi = mean(Y)  (src/Hmc.jl:136); always the mean of the REAL window ----
    double part = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) part += y[l];
    const double xi = block_sum<NW>(part, sh.bred, wave, lane) / (double)T;

    double pivot[K];                     // pivots of the one-pass statistics (any value is exact in exact arithmetic)
#pragma unroll
    for (int k = 0; k < K; ++k) pivot[k] = xi;

    const int NCK = 3 * K + K * K + 2 * p.H + K;      // checkpoint block: summary sums + pivots
    if (p.resume) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (t0 + l < T) x[l] = min((int)p.xstate[(size_t)w * p.ldY + t0 + l], K - 1);
        if (p.sumacc) {
#pragma unroll
            for (int k = 0; k < K; ++k) pivot[k] = p.sumacc[(size_t)w * NCK + (NCK - K) + k];
        }
    } else if (p.x_init) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (t0 + l < T) x[l] = min(max(p.x_init[(size_t)w * p.ldY + t0 + l], 0), K - 1);   // caller data: clamp
    } else {
        // ---- makeParams (src/Hmc.jl:161-195): mu spread around the median, X = argmax pdf.
        // The initial "sigma" (= std(Y), :177) is the same for every state and is overwritten by the
        // first sweep before any other use, so argmax pdf == nearest initial mean (first index on
        // ties) and std(Y) itself is never needed.  Distances, not pdf values, are compared so the
        // decision does not depend on the exp implementation (see oracle/hmc_oracle.c chain_init).
        double lmin = 1.0e308, lmax = -1.0e308;
#pragma unroll
        for (int l = 0; l < L; ++l) if (t0 + l < T) { lmin = fmin(lmin, y[l]); lmax = fmax(lmax, y[l]); }
        const double ymin = block_minmax<NW>(lmin, sh.bred, wave, lane, false);
        const double ymax = block_minmax<NW>(lmax, sh.bred, wave, lane, true);
        // median: the lower middle order statistic by radix selection (block_select); for an even T the upper one is the same
        // value when more than (T-1)/2 + 1 elements are <= it, else the smallest value above it
        unsigned long long key[L];
#pragma unroll
        for (int l = 0; l < L; ++l) key[l] = order_key(y[l]);
        const int rlo = (T - 1) / 2;
        const double med_lo = key_value(block_select<NW, L>(L, [&](int l, unsigned long long& k) __attribute__((always_inline)) { k = key[l]; return t0 + l < T; },
                                                            rlo, sh.selcnt, wave, lane));
        double med_hi = med_lo;
        if (!(T & 1)) {                  // uniform
            double nle = 0.0, above = 1.0e308;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (!helper && t0 + l < T) {
                    if (y[l] <= med_lo) nle += 1.0;
                    else above = fmin(above, y[l]);
                }
            }
            const double cle = block_sum<NW>(nle, sh.bred, wave, lane);
            const double nxt = block_minmax<NW>(above, sh.bred, wave, lane, false);
            med_hi = (cle > (double)(rlo + 1)) ? med_lo : nxt;
        }
        {
            // bit-identical to the oracle: no FMA contraction in this block (tie rule, see above)
#pragma clang fp contract(off)
            const double med = (T & 1) ? med_lo : med_lo / 2 + med_hi / 2;
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
                x[l] = (t0 + l < T) ? best : XPAD;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) pivot[k] = mu0[k];
        }
        __syncthreads();                 // sh.ux is reused by the sweeps
    }

    // first state of the next thread's chunk (X at t0+L), and X[T-1]
    int xnext = 0;
    if (!helper) sh.xfirst[tid] = x[0];
    if (tid == 0) sh.xfirst[NT] = XPAD;
    if (tid == owner) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (l == l_last) sh.x_end = x[l];
    }
    __syncthreads();
    xnext = sh.xfirst[helper ? NT : tid + 1];
    int x_end = sh.x_end;

    Rng rng{p.seed_lo, p.seed_hi, p.window_ids ? p.window_ids[w] : p.window_base + (uint32_t)w, 0u};
    const int NS = 3 * K + KK + 2 * p.H;

    // ======================= shadow jobs (waves 1..NW-1) ======================
    const int shadow_wave = wave - 1;                  // -1 on wave 0
    constexpr int NSH = NW - 1;

    // State-independent parts of sweep `sw`'s parameter draws -> sh.rb[sw & 1].  One task per lane:
    //   [0, K)                rho exponentials (site 2), normalised across these K lanes
    //   [K, K+2NG)            normal of gamma role g>>1, attempt g&1           (block index 2j)
    //   [K+2NG, 2K+2NG)       normal for mu_i                                  (site 1)
    //   [2K+2NG, 2K+4NG)      log(1-u) of gamma role, attempt                  (block index 2j+1)
    // Every task is one Philox block and one log; the normals add sqrt*cos.
    auto job_prep = [&](int sw) __attribute__((always_inline)) {
        Rng g = rng;
        g.sweep = (uint32_t)sw;
        RngBuf<K>& rb = sh.rb[sw & 1];
        constexpr int NTASK = 4 * NG + 2 * K;
        for (int base = 0; base < NTASK; base += 64) {
            const int task = base + lane;
            const bool live = task < NTASK;
            const bool t_rho = task < K;
            const bool t_gx = !t_rho && task < K + 2 * NG;
            const bool t_z = !t_rho && !t_gx && task < 2 * K + 2 * NG;
            const bool t_gl = !t_rho && !t_gx && !t_z;
            const int gt = t_gx ? task - K : task - (2 * K + 2 * NG);      // gamma task id for gx / gl
            const int role = gt >> 1, j = gt & 1;
            uint32_t site = SITE_RHO, elem = (uint32_t)task, idx = 0;
            if (t_gx || t_gl) {
                site = role < K ? SITE_SIG2 : SITE_A;
                elem = role < K ? (uint32_t)role : (uint32_t)(role - K);
                idx = 2u * (uint32_t)j + (t_gl ? 1u : 0u);
            } else if (t_z) {
                site = SITE_MU; elem = (uint32_t)(task - (K + 2 * NG));
            }
            uint32_t r[4];
            g.block(site, elem, idx, r);
            const double uraw = u53(r[0], r[1]);
            const double lg = log_fast(1.0 - uraw);
            double val = lg;
            if (t_gx || t_z) val = sqrt_fast(-2.0 * lg) * cos2pi_fast(u53(r[2], r[3]));   // Box-Muller
            // rho ~ Dirichlet(ones(K)) (:350-356): K exponentials normalised in element order (lanes 0..K-1 of pass 0)
            double rs = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) rs += -__shfl(lg, i, 64);
            if (live) {
                if (t_rho) rb.rho[task] = -lg * (1.0 / rs);
                else if (t_gx) rb.x[role][j] = val;
                else if (t_z) rb.z[task - (K + 2 * NG)] = val;
                else if (t_gl) { rb.lu[role][j] = val; rb.uu[role][j] = uraw; }
            }
        }
    };
    static_assert(NW > 2 || 3 * K + KK <= 64 - 2 * HMCG_MAXH, "parameter and forecast output lanes share a wave when NW == 2");
    // uniforms for the state draws of sweep `sw` (site 4, index t): block b covers t = 2b, 2b+1.
    // Blocks [b0, b1) are dealt round-robin to the calling wave's lanes.
    auto job_uniforms = [&](int sw, int b0, int b1) __attribute__((always_inline)) {
        Rng g = rng;
        g.sweep = (uint32_t)sw;
        for (int b = b0 + lane; b < b1; b += 128) {       // two independent blocks per trip (ILP across the mul chains)
            const int bb = b + 64;
            uint32_t r[4], q[4];
            g.block(SITE_X, 0, (uint32_t)b, r);
            g.block(SITE_X, 0, (uint32_t)bb, q);
            sh.ux[2 * b] = u53(r[0], r[1]);
            if (2 * b + 1 < NT * L) sh.ux[2 * b + 1] = u53(r[2], r[3]);
            if (bb < b1) {
                sh.ux[2 * bb] = u53(q[0], q[1]);
                if (2 * bb + 1 < NT * L) sh.ux[2 * bb + 1] = u53(q[2], q[3]);
            }
        }
    };

    constexpr int FC_SPLIT_MAX = 26;      // beyond this, binary exponentiation (27 log2 h fma) beats 9 h/2
    double sum_acc = 0.0;                 // running sum behind `summary` (meaningful on the output lanes only)
    double smp_acc = 0.0;                 // signal path: the same sum over the running noise sample (extras.sample_summary)
    (void)smp_acc;
    constexpr int OUT_WAVE = NH > 0 ? NW + 3 : 1;        // owns the 3K + K^2 parameter output lanes
    constexpr int FC_WAVE = NH > 0 ? NW + 2 : NW - 1;    // owns the 2H forecast lanes (== OUT_WAVE when NW == 2)
    constexpr int PREP_WAVE = NH > 0 ? NW + 1 : (NW - 1 >= 3 ? 2 : 1);   // prepares the next sweep's RNG parts
    const int NP = 3 * K + KK;
    // output role of a lane: [0, 3K) mu | sig2 | pi_end by sorted position, [3K, NP) A(:) column-major, [NP, NP + 2H)
    // forecast / forecast error per horizon; -1: none.  The role and what follows from it (output column pointer,
    // horizon, realised value) are decoded ONCE per launch into LDS (sh.oc) and read back on every call: as per-lane
    // registers they would stay live across the whole sweep loop in every wave -- the long-lived values the register
    // allocator splits, with the copies it then places around divergent blocks being where the backend fault of
    // DESIGN.md section 5a strikes (and ~10 registers on every variant); decoded afresh on every call they put ~40
    // dependent integer instructions and two memory loads at the head of the forecast job, which bounds the parameter phase.
    auto role_of = [&](int ln) __attribute__((always_inline)) -> int {
        int r = -1;
        if (wave == OUT_WAVE && ln < NP) r = ln;
        if (wave == FC_WAVE && ln >= 64 - 2 * HMCG_MAXH && ln - (64 - 2 * HMCG_MAXH) < 2 * p.H) r = NP + ln - (64 - 2 * HMCG_MAXH);
        return r;
    };
    const int oslot = (wave == FC_WAVE && FC_WAVE != OUT_WAVE) ? 1 : 0;
    if (wave == OUT_WAVE || wave == FC_WAVE) {
        const int orole = role_of(lane);
        double* out_base = nullptr;           // element (d = draw_off) of this lane's output column
        int o_which = 0, o_q = 0, o_i = 0, o_j = 0, fc_h = 0, fc_blend = 0;
        double fc_yr = 0.0;
        const size_t nrun = (size_t)p.nd_ld;
        if (orole >= 0 && orole < 3 * K) {
            o_q = orole % K; o_which = orole / K;          // 0 mu, 1 sig2, 2 pi_end; sorted position q
            double* base = o_which == 0 ? p.mu : (o_which == 1 ? p.sig2 : p.pi_end);
            if (base) out_base = base + nrun * ((size_t)o_q + (size_t)K * w);
        } else if (orole >= 3 * K && orole < NP) {
            const int e = orole - 3 * K;                   // column-major: e = i + K*j (src/Hmc.jl:745)
            o_which = 3; o_i = e % K; o_j = e / K;
            if (p.A) out_base = p.A + nrun * ((size_t)e + (size_t)KK * w);
        } else if (orole >= NP) {
            const int e = orole - NP;                      // 2h + {0: forecast, 1: error}
            o_which = 4 + (e & 1);
            fc_h = p.horizons[e >> 1];
            fc_blend = (SIG && ((p.blend_mask >> (e >> 1)) & 1)) ? 1 : 0;
            fc_yr = p.yreal ? p.yreal[(size_t)w * p.H + (e >> 1)] : __builtin_nan("");
            if (p.fcast) out_base = p.fcast + nrun * ((size_t)e + (size_t)(2 * p.H) * w);
        }
        typename Sh::OutConst c;
        c.base = out_base; c.yr = fc_yr; c.h = fc_h;
        c.packed = o_which | (o_q << 4) | (o_i << 8) | (o_j << 12) | (fc_blend << 16) | ((orole >= 0 ? orole : 0) << 20) | (orole >= 0 ? (1 << 30) : 0);
        sh.oc[oslot][lane] = c;                            // (read back by the same lane only; the prologue's barriers follow)
        if (orole >= 0 && p.resume && p.sumacc) sum_acc = p.sumacc[(size_t)w * NCK + orole];
        if constexpr (SIG) {
            // a launch that resumes inside a noise sample picks that sample's running sums up from its row
            if (orole >= 0 && p.resume && p.sample_summary) {
                const int smp = p.sweep_begin / p.per_sample, kb = p.sweep_begin - smp * p.per_sample - p.burnin_s;
                if (kb > 0 && kb < p.nrun_s && smp < p.n_samples) smp_acc = p.sample_summary[((size_t)w * p.n_samples + smp) * NS + orole];
            }
        }
    }
    // per-draw outputs of sweep `sw` (whose parameters sit in sh.th[sw & 1])
    auto job_outputs = [&](int sw) __attribute__((always_inline)) {
        // (without the signal path a launch is one sample: no division needed to find the kept-draw index)
        const int d = SIG ? kept_index(p, sw) : (sw >= p.burnin_s ? sw - p.burnin_s : -1);
        if (d < 0 || !(wave == OUT_WAVE || wave == FC_WAVE)) return;
        const typename Sh::OutConst c = sh.oc[oslot][lane];
        if (!(c.packed >> 30)) return;
        double* const out_base = c.base;
        const int o_which = c.packed & 15, o_q = (c.packed >> 4) & 15, o_i = (c.packed >> 8) & 15, o_j = (c.packed >> 12) & 15;
        const int fc_h = c.h;
        const bool fc_blend = SIG && ((c.packed >> 16) & 1);       // signal path: this horizon is a forecastsignal blend
        const double fc_yr = c.yr;
        const ThetaBuf<K>& th = sh.th[sw & 1];
        double val;
        if (o_which >= 4) {
            // forecast (src/Hmc.jl:658-667).  (pi' A^h) . mu is invariant under the label permutation, so the
            // unsorted parameters are used as they are.
            double fv;
            if (SIG && fc_blend) {
                // forecastsignal (src/Hmc.jl:670-681, :908-909): the horizon equals the number of signal steps past the
                // end date; blend of the last noisy observation and each state mean, weighted by the last step's
                // probabilities.  noise = sigma_signal, tau = 1/noise, a = tau / (1 + tau).
                const double tau = 1.0 / (p.sigma_signal ? p.sigma_signal[w] : 0.0);
                const double a = tau / (1.0 + tau);
                const double ysig = sh.y_last[(sw / p.per_sample) & 1];
                fv = 0.0;
#pragma unroll
                for (int i = 0; i < K; ++i) fv += th.pi_end[i] * (a * ysig + (1.0 - a) * th.mu[i]);
            } else if (fc_h <= FC_SPLIT_MAX) {
                // short horizons: the lane pair of a horizon (forecast, forecast error) meets in the middle --
                // the even lane carries pi' A^(h/2) (as (A')^(h/2) pi), the odd lane A^(h - h/2) mu, one
                // exchange and a dot product finish.  Half the dependent chain of the power iteration.
                const int side = o_which - 4;
                const int n_own = side ? fc_h - (fc_h >> 1) : (fc_h >> 1);
                const double* Af = &th.A[0][0];
                double M[K][K], v[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    v[i] = side ? th.mu[i] : th.pi_end[i];
#pragma unroll
                    for (int j = 0; j < K; ++j) M[i][j] = Af[side ? i * K + j : j * K + i];
                }
                for (int n = 0; n < n_own; ++n) {           // per-lane trip count (differs by at most one in a pair)
                    double nv[K];
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        double acc = M[i][0] * v[0];
#pragma unroll
                        for (int j = 1; j < K; ++j) acc = fma(M[i][j], v[j], acc);
                        nv[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < K; ++i) v[i] = nv[i];
                }
                fv = 0.0;
#pragma unroll
                for (int i = 0; i < K; ++i) fv = fma(v[i], __shfl_xor(v[i], 1, 64), fv);
            } else {
                double mu_u[K], pe_u[K], A_u[K][K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    mu_u[i] = th.mu[i]; pe_u[i] = th.pi_end[i];
#pragma unroll
                    for (int j = 0; j < K; ++j) A_u[i][j] = th.A[i][j];
                }
                fv = forecast_value<K>(mu_u, A_u, pe_u, fc_h);
            }
            val = (o_which == 5) ? fv - fc_yr : fv;
        } else {
            double mu_u[K];
            int order[K];
#pragma unroll
            for (int i = 0; i < K; ++i) mu_u[i] = th.mu[i];
            sort_order<K>(mu_u, order);
            int si = 0, sj = 0;
#pragma unroll
            for (int qq = 0; qq < K; ++qq) {
                si = (qq == (o_which == 3 ? o_i : o_q)) ? order[qq] : si;
                sj = (qq == o_j) ? order[qq] : sj;
            }
            // sorted views: mu[order[q]], sig2[order[q]], pib_end[order[q]], A[order[i], order[j]] (:502-513)
            const double* src = o_which == 0 ? &th.mu[0] : (o_which == 1 ? &th.sig2[0] : (o_which == 2 ? &th.pi_end[0] : &th.A[0][0]));
            val = src[o_which == 3 ? si * K + sj : si];
            if constexpr (SIG) {
                if (tail > 0 && o_which == 2) {
                    // signals past the end date: report pib[end_pos,:] (:900) = pif[end_pos,:] o b, b = M_{end_pos+1} ... M_{T-1} 1
                    // by the backward recursion b <- A (f_t o b) (backwardupdate_P!, :442-457) over the `tail` last steps
                    double b[K];
#pragma unroll
                    for (int r = 0; r < K; ++r) b[r] = 1.0;
                    for (int j = tail - 1; j >= 0; --j) {
                        double g[K], nb[K];
#pragma unroll
                        for (int c = 0; c < K; ++c) g[c] = sh.ftail[sw & 1][j][c] * b[c];
#pragma unroll
                        for (int r = 0; r < K; ++r) {
                            double acc = 0.0;
#pragma unroll
                            for (int c = 0; c < K; ++c) acc = fma(th.A[r][c], g[c], acc);
                            nb[r] = acc;
                        }
                        rescale_pow2<K>(nb);
#pragma unroll
                        for (int r = 0; r < K; ++r) b[r] = nb[r];
                    }
                    double tot = 0.0, mine = 0.0;
#pragma unroll
                    for (int c = 0; c < K; ++c) {
                        const double g = sh.pf_rep[sw & 1][c] * b[c];
                        tot += g;
                        mine = (c == si) ? g : mine;
                    }
                    val = mine / tot;
                }
            }
        }
        // (the pointer came back from LDS as a generic one: say that it is global, or the store would be a flat_store)
        if (out_base) ((__attribute__((address_space(1))) double*)out_base)[d - p.draw_off] = val;
        const double r5 = round5(val);
        sum_acc += r5;
        if constexpr (SIG) {
            if (p.sample_summary) {
                smp_acc += r5;
                const int smp = sw / p.per_sample;
                if (sw + 1 == (smp + 1) * p.per_sample) {       // the sample's last sweep: its row is complete
                    const int orole = (c.packed >> 20) & 63;
                    p.sample_summary[((size_t)w * p.n_samples + smp) * NS + orole] = p.nrun_s > 0 ? smp_acc / (double)p.nrun_s : __builtin_nan("");
                    smp_acc = 0.0;
                }
            }
        }
    };

#ifdef HMCG_STAMPS
    unsigned long long stamp_acc[HMCG_NSTAMP];
    unsigned long long stamp_prev = 0;
#endif
    // per-wave partial statistics of the current X: counts, pivoted sums, transitions
    constexpr int PB = L <= 1 ? 1 : (L <= 3 ? 2 : (L <= 7 ? 3 : (L <= 15 ? 4 : 5)));   // bits holding a per-thread count <= L
    constexpr int FPW = 32 / PB;                       // per-thread fields per 32-bit word
    constexpr int NF = Sh::NF;
    constexpr int NWORD = (NF + FPW - 1) / FPW;
    constexpr int NPK = Sh::NPK;
    constexpr bool PADCNT = PADMARK && !SIG && NWORD == 1 && NF < FPW;    // counts without `t + 1 < T` predicates
    static_assert(64 * L < 65536, "wave totals fit their fields (10 bits while 64 L < 1024, else 16)");
    auto publish_stats = [&]() __attribute__((always_inline)) {
        // ---- transition counts C_ij: per-thread PB-bit fields -> 16-bit fields -> one DPP integer sum per word
        unsigned acc[NWORD];
#pragma unroll
        for (int wd = 0; wd < NWORD; ++wd) acc[wd] = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int xn = (l + 1 < L) ? x[(l + 1 < L) ? l + 1 : l] : xnext;
            const int code = PADCNT ? x[l] + K * xn : x[l] * K + xn;
            const bool pv = (t0 + l + 1) < T;
            if constexpr (PADCNT) {
                // field j*K+i for the pair i -> j: a pair whose successor is padded (xn = K; a padded step is never
                // followed by a real one) has code >= K*K, and all of those land in the spare field K*K
                acc[0] += 1u << (PB * min(code, KK));
            } else if constexpr (NWORD == 1) {
                acc[0] += pv ? (1u << (PB * code)) : 0u;
            } else {
#pragma unroll
                for (int wd = 0; wd < NWORD; ++wd) {
                    const int rel = code - wd * FPW;
                    acc[wd] += (pv && rel >= 0 && rel < FPW) ? (1u << (PB * rel)) : 0u;
                }
            }
            if constexpr (SIG) {                       // M_i: signal steps in state i
                const int t = t0 + l;
                const bool sv = t >= sb && t < se;
                const int c2 = KK + x[l];
#pragma unroll
                for (int wd = 0; wd < NWORD; ++wd) {
                    const int rel = c2 - wd * FPW;
                    acc[wd] += (sv && rel >= 0 && rel < FPW) ? (1u << (PB * rel)) : 0u;
                }
            }
        }
        unsigned pk[NPK];
#pragma unroll
        for (int d = 0; d < NPK; ++d) {
            unsigned v = 0;
#pragma unroll
            for (int q = 0; q < Sh::FPK; ++q) {
                const int e = Sh::FPK * d + q;
                if (e < NF) v |= ((acc[e / FPW] >> (PB * (e % FPW))) & ((1u << PB) - 1u)) << (Sh::FW * q);
            }
            pk[d] = v;
        }
        wave_sum_words_lane63<NPK>(pk);
        // ---- pivoted sums by state: d1_i = sum (y - pivot_i), d2_i = sum (y - pivot_i)^2 over the observation
        // positions (and, on the signal path, the same two sums over the signal positions)
        double d1[K], d2[K], s1[SIG ? K : 1], s2[SIG ? K : 1];
#pragma unroll
        for (int i = 0; i < K; ++i) { d1[i] = 0.0; d2[i] = 0.0; }
        if constexpr (SIG) {
#pragma unroll
            for (int i = 0; i < K; ++i) { s1[i] = 0.0; s2[i] = 0.0; }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            double pvt = pivot[0];
#pragma unroll
            for (int k = 1; k < K; ++k) pvt = (x[l] == k) ? pivot[k] : pvt;
            const double dl = y[l] - pvt;
            const int t = t0 + l;
            const bool issig = SIG && t >= sb && t < se;
            const int xs = PADMARK ? (issig ? -1 : x[l]) : ((t < T && !issig) ? x[l] : -1);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double dm = (xs == i) ? dl : 0.0;
                d1[i] += dm;
                d2[i] = fma(dm, dm, d2[i]);
            }
            if constexpr (SIG) {
                const int xg = issig ? x[l] : -1;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const double dm = (xg == i) ? dl : 0.0;
                    s1[i] += dm;
                    s2[i] = fma(dm, dm, s2[i]);
                }
            }
        }
        if constexpr (SIG) {
            static_assert(!SIG || K <= 4, "signal path: transposed reduce carries 2 x 4 slots");
            double v8[8], o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v8[i] = i < K ? s1[i < K ? i : 0] : 0.0; v8[4 + i] = i < K ? s2[i < K ? i : 0] : 0.0; }
            wave_sum8_transposed(v8, lane, o0, o1);
            if ((lane & 0x3C) == 12) {
                const int i0 = lane & 2, i1 = i0 + 1;
                double* dst = (lane & 1) ? &sh.red_s2[wave][0] : &sh.red_s1[wave][0];
                if (i0 < K) dst[i0] = o0;
                if (i1 < K) dst[i1] = o1;
            }
        }
        if constexpr (K <= 4) {
            double v8[8], o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v8[i] = i < K ? d1[i] : 0.0; v8[4 + i] = i < K ? d2[i] : 0.0; }
            wave_sum8_transposed(v8, lane, o0, o1);
            if ((lane & 0x3C) == 12) {                 // lanes 12..15 hold the totals
                const int i0 = lane & 2, i1 = i0 + 1;  // state index of o0 / o1; lane&1 picks d1 or d2
                double* dst = (lane & 1) ? &sh.red_d2[wave][0] : &sh.red_d1[wave][0];
                if (i0 < K) dst[i0] = o0;
                if (i1 < K) dst[i1] = o1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) { d1[i] = wave_sum(d1[i]); d2[i] = wave_sum(d2[i]); }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < K; ++i) { sh.red_d1[wave][i] = d1[i]; sh.red_d2[wave][i] = d2[i]; }
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int d = 0; d < NPK; ++d) sh.red_pk[wave][d] = pk[d];
        }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) sh.pivot[k] = pivot[k];
            sh.x_end = x_end;
        }
    };
    // a new noise sample (src/Hmc.jl:892): Yfake = Yreal + N(0,1) * sigma_signal on the signal range; the chain
    // state carries over (:889-895) and the statistics are retaken on the new data
    auto regen_y = [&](int smp, bool report) __attribute__((always_inline)) {
        if constexpr (SIG) {
            const double ssig = p.sigma_signal ? p.sigma_signal[w] : 0.0;
            const bool noisy = ssig != 0.0 || p.n_samples > 1;
            int svb = 0, sve = 0;
            if (p.save_range) { svb = p.save_range[2 * w]; sve = p.save_range[2 * w + 1]; }
            Rng gn = rng;
            gn.sweep = (uint32_t)smp;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = t0 + l;
                if (noisy && t >= sb && t < se) {
                    uint32_t r[4];
                    gn.block(SITE_NOISE, 0, (uint32_t)t, r);
                    y[l] = yreal[l] + box_muller(r) * ssig;
                }
                if (report && p.sigvals && t >= svb && t < sve && t < T && (t - svb) < p.nsave_ld)
                    p.sigvals[((size_t)w * p.n_samples + smp) * p.nsave_ld + (t - svb)] = y[l];
                if (t == T - 1) sh.y_last[smp & 1] = y[l];       // forecastsignal's `signal` (:909)
            }
        }
    };
    if constexpr (SIG) {
        if (p.sweep_begin % p.per_sample != 0) regen_y(p.sweep_begin / p.per_sample, false);   // resumed inside a sample
    }
    if (!helper) publish_stats();

    // prologue: the first sweep's state-independent RNG parts
    if (p.sweep_begin < p.sweep_end && shadow_wave == 0) job_prep(p.sweep_begin);

    // Philox blocks of the state-draw uniforms when helper waves exist: dealt in trips of 128 blocks (two per
    // lane, ~1.1k cycles whatever the fill) to the waves with time to spare while wave 0 draws (~2.4k cycles):
    // primary waves 1..3 carry nothing else (two trips each fit), helper 0 has no other job (it shares wave 0's
    // SIMD, whose older wave keeps issue priority), then helper 3 (parameter outputs), the forecast and the
    // RNG-preparation helpers; beyond that round-robin over the primaries.  Trip k covers blocks [128k, 128k+128).
    uint32_t trips = 0;
    if constexpr (NH > 0) {
        const int ntrip = (((T + 1) >> 1) + 127) >> 7;
        constexpr int trip_wave[10] = {1, 2, 3, NW, 1, 2, 3, NW + 3, NW + 2, NW + 1};
        for (int k = 0; k < ntrip && k < 32; ++k) {
            const int wv = k < 10 ? trip_wave[k] : 1 + (k - 10) % 3;
            trips |= (wv == wave) ? (1u << k) : 0u;
        }
        trips = __builtin_amdgcn_readfirstlane(trips);
    }
    auto job_uniform_trips = [&](int sw) __attribute__((always_inline)) {
        const int nblk = (T + 1) >> 1;
        for (uint32_t m = trips; m != 0; m &= m - 1) {
            const int k = __builtin_ctz(m);
            job_uniforms(sw, k << 7, min((k << 7) + 128, nblk));
        }
    };

    double pf[L][K];     // unsorted filtered probabilities of this thread's steps
    double sm_acc[SMOOTH ? L : 1][K];   // running sums of the smoothed probabilities of this thread's steps
    double fm_acc[SMOOTH ? L : 1][K];   // ... and of the filtered ones
    if constexpr (SMOOTH) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int q = 0; q < K; ++q)
            {
                sm_acc[l][q] = (p.resume && p.pi_smooth_mean && t0 + l < T) ? p.pi_smooth_mean[((size_t)w * p.ldY + t0 + l) * K + q] : 0.0;
                fm_acc[l][q] = (p.resume && p.pi_filter_mean && t0 + l < T) ? p.pi_filter_mean[((size_t)w * p.ldY + t0 + l) * K + q] : 0.0;
            }
    }
    (void)sm_acc; (void)fm_acc;
#ifdef HMCG_STAMPS
    for (int i = 0; i < HMCG_NSTAMP; ++i) stamp_acc[i] = 0;
    stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    if constexpr (NH > 0) {
        if (helper) {
            // ---- helper waves: the draw-phase jobs, then only the sweep's barriers ----
            for (int sweep = p.sweep_begin; sweep < p.sweep_end; ++sweep) {
                __syncthreads();                                             // Ba
                STAMP(0);
                if (wave == PREP_WAVE && sweep + 1 < p.sweep_end) job_prep(sweep + 1);
                STAMP(15);
                job_uniform_trips(sweep);
                STAMP(1);
                __syncthreads();                                             // Bb
                STAMP(2);
                // the PREVIOUS sweep's per-draw outputs and forecasts (theta[par ^ 1] stays untouched until the next
                // parameter phase), in the shadow of the primaries' pdf / product / scan phases (~5k cycles) -- not in the
                // parameter phase, which the forecast job (2.3k cycles) would bound together with wave 0.  (Measured: in
                // the shadow of the shorter backward pass instead, the helpers delayed barrier Be by ~0.9k cycles.)
                if (sweep > p.sweep_begin) job_outputs(sweep - 1);
                STAMP(14);
                __syncthreads();                                             // Bc
                __syncthreads();                                             // Bd
                __syncthreads();                                             // Be
                STAMP(11);
            }
        }
    }

    for (int sweep = helper ? p.sweep_end : p.sweep_begin; sweep < p.sweep_end; ++sweep) {
        rng.sweep = (uint32_t)sweep;
        const int par = sweep & 1;
        ThetaBuf<K>& th = sh.th[par];
        if constexpr (SIG) {
            const int smp = sweep / p.per_sample;
            if (sweep == smp * p.per_sample) { regen_y(smp, true); publish_stats(); }
        }
        __syncthreads();                                                     // Ba: statistics + rb[par] ready
        STAMP(0);
        if (wave == 0) {
            // ---- parameter draws (sites 0,1,3; site 2 = rho comes ready-made from the shadow) ----
            const RngBuf<K>& rb = sh.rb[par];
            // Lanes 0..K-1 draw sig2_i / mu_i; row i of A sits in its own quad, lanes 4+4i .. 4+4i+K-1 (K <= 4): the
            // row sums (N_i, the Dirichlet normaliser) are then quad reductions by DPP instead of LDS shuffles.
            const int qi = (lane - 4) >> 2, qj = (lane - 4) & 3;
            const bool is_sig = lane < K, is_A = lane >= 4 && qi < K && qj < K;
            const bool is_g = is_sig || is_A;
            const int role = is_sig ? lane : (is_A ? K + qi * K + qj : NG);
            double shape = 1.0, bpar = 1.0, Neff = 0.0, Ssum = 0.0, rnn = 0.0;
            // transition count C_e of this lane's A role (e = role - K), summed over the waves' packed words
            int cT = 0;
            {
                const int e = is_A ? (PADCNT ? qj * K + qi : role - K) : 0;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) cT += (int)((sh.red_pk[ww][e / Sh::FPK] >> (Sh::FW * (e % Sh::FPK))) & ((1u << Sh::FW) - 1u));
                cT = is_A ? cT : 0;
            }
            // N_i = sum_j C_ij + [X[T-1] == i]: quad sums of the A lanes, handed to the sig2 lanes as scalars
            int rowsum = 0;
            {
                int qs = cT + __builtin_amdgcn_mov_dpp(cT, 0xB1, 0xF, 0xF, false);        // quad_perm:[1,0,3,2]
                qs += __builtin_amdgcn_mov_dpp(qs, 0x4E, 0xF, 0xF, false);                // quad_perm:[2,3,0,1]
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const int ri = __builtin_amdgcn_readlane(qs, 4 + 4 * i);
                    rowsum = (lane == i) ? ri : rowsum;
                }
            }
            STAMP(16);
            if (is_g) {
                const int c = is_sig ? rowsum + ((sh.x_end == role) ? 1 : 0) : cT;
                if (is_sig) {
                    double d1 = 0.0, d2 = 0.0;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) { d1 += sh.red_d1[ww][role]; d2 += sh.red_d2[ww][role]; }
                    const double piv = sh.pivot[role];
                    const double beta = (sweep == 0) ? 1.0 : 2.0;                // quirk 2 (:179, :347)
                    if constexpr (!SIG) {
                        Neff = (double)c;
                        const double rn = c > 0 ? rcp_fast(Neff) : 0.0;
                        const double ybar = c > 0 ? piv + d1 * rn : 0.0;             // :259-265, :282-288
                        const double S2 = c > 0 ? fmax(d2 - d1 * d1 * rn, 0.0) : 0.0;  // sum (y-ybar)^2 (:291-294)
                        Ssum = piv * Neff + d1;                                      // sum of y in the state
                        const double dm = ybar - xi;
                        shape = p.alpha + 0.5 * Neff;                                // :313
                        rnn = rcp_fast(Neff + p.nu);
                        bpar = beta + 0.5 * S2 + 0.5 * Neff * p.nu * rnn * (dm * dm);   // :314
                    } else {
                        // observation set and signal set (src/Hmc.jl:254-314)
                        int Mi = 0;
                        {
                            const int e = KK + role;
#pragma unroll
                            for (int ww = 0; ww < NW; ++ww) Mi += (int)((sh.red_pk[ww][e / Sh::FPK] >> (Sh::FW * (e % Sh::FPK))) & ((1u << Sh::FW) - 1u));
                        }
                        const int Ni = c - Mi;
                        double g1 = 0.0, g2 = 0.0;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) { g1 += sh.red_s1[ww][role]; g2 += sh.red_s2[ww][role]; }
                        const double dNi = (double)Ni, dMi = (double)Mi;
                        const double S = piv * dNi + d1, Sm = piv * dMi + g1;                       // sums of y
                        const double S2 = Ni > 0 ? fmax(d2 - d1 * d1 / dNi, 0.0) : 0.0;            // :291-294
                        const double Sm2 = Mi > 0 ? fmax(g2 - g1 * g1 / dMi, 0.0) : 0.0;           // :296-300
                        const double totalbar = (Ni + Mi) > 0 ? (S + Sm) / (dNi + dMi) : 0.0;      // :282-288
                        Neff = dNi + dMi * kfac;                                                    // :302-303
                        Ssum = S + Sm;                                                              // :331 (Sm unscaled: quirk 4)
                        const double dm = totalbar - xi;
                        shape = p.alpha + 0.5 * dNi + 0.5 * dMi;                                    // :313
                        rnn = rcp_fast(Neff + p.nu);
                        bpar = beta + 0.5 * S2 + (0.5 * kfac) * Sm2 + 0.5 * Neff * p.nu * rnn * (dm * dm);   // :314
                    }
                } else {
                    shape = (double)(c + 1);                                     // :362-365
                }
            }
            STAMP(17);
            double val = 1.0;
            if (is_g) {
                const uint32_t site = role < K ? SITE_SIG2 : SITE_A;
                const uint32_t elem = role < K ? (uint32_t)role : (uint32_t)(role - K);
                if (shape == 1.0) {
                    // Gamma(1) = exponential of block index 0: only for a never-visited state pair; inline
                    uint32_t r[4];
                    rng.block(site, elem, 0, r);
                    val = -log_fast(1.0 - u53(r[0], r[1]));
                } else {
                    const double a = shape < 1.0 ? shape + 1.0 : shape;
                    const double dd = a - 1.0 / 3.0;
                    const double cc = rsqrt_fast(dd) * (1.0 / 3.0);               // 1 / (3 sqrt(d))
                    val = mt_try(dd, cc, rb.x[role][0], rb.lu[role][0], rb.uu[role][0]);
                    if (val < 0.0) val = mt_try(dd, cc, rb.x[role][1], rb.lu[role][1], rb.uu[role][1]);
                    if (val < 0.0) {
                        // third and later attempts (rare): inline Philox
                        int j = 2;
                        for (; j < GAMMA_MAX_ATTEMPTS && val < 0.0; ++j) {
                            uint32_t r[4];
                            rng.block(site, elem, 2u * (uint32_t)j, r);
                            const double xx = box_muller(r);
                            rng.block(site, elem, 2u * (uint32_t)j + 1u, r);
                            const double u3 = u53(r[0], r[1]);
                            val = mt_try(dd, cc, xx, log_fast(1.0 - u3), u3);
                        }
                        if (val < 0.0) { val = dd; st |= HMCG_ST_GAMMA_CAP; }
                    }
                    if (shape < 1.0) {
                        uint32_t r[4];
                        rng.block(site, elem, 0xFFFFFFFFu, r);
                        val *= pow(1.0 - u53(r[0], r[1]), 1.0 / shape);
                    }
                }
            }
            STAMP(18);
            // normalise the Dirichlet rows of A (:367): quad sum of the row's variates ((v0 + v1) + (v2 + v3), absent
            // columns contribute an exact 0: for K <= 3 this is the column-order sum)
            double gs = is_A ? val : 0.0;
            gs += quadperm_f64<0xB1>(gs);
            gs += quadperm_f64<0x4E>(gs);
            if (is_sig) {
                const double sig2 = bpar * rcp_fast(val);                        // :320 InverseGamma(a,b) = b / Gamma(a,1)
                const double m = (Ssum + p.nu * xi) * rnn;                       // :331
                const double sdev = sqrt_fast(sig2 * rnn);                       // :332 sqrt(sig2/(Neff+nu))
                const double mu = m + sdev * rb.z[role];                         // :334
                const double isd = rsqrt_fast(sig2);                             // 1/sd (:381)
                th.mu[role] = mu; th.sig2[role] = sig2;
                th.isd[role] = isd * 0.70710678118654752440; th.coef[role] = INVSQRT2PI * isd;
                th.rho[role] = rb.rho[role];                                     // :355
            } else if (is_A) {
                th.A[qi][qj] = val * rcp_fast(gs);
            }
        } else {
            // ---- shadow of the parameter draws ----
            // Work list: parameter outputs of the previous sweep (wave 1), its forecasts (last wave), the
            // next sweep's RNG preparation (wave 1), and this sweep's (T+1)/2 Philox blocks of state-draw
            // uniforms, dealt to the shadow waves in proportion to what else they carry.
            const int nblk = (T + 1) >> 1;
            if (sweep > p.sweep_begin) job_outputs(sweep - 1);
            STAMP(14);
            if constexpr (NH > 0) {
                job_uniform_trips(sweep);
            } else if (NSH >= 3) {
                // fixed jobs: wave 1 parameter outputs (~1.0k cycles), wave 2 RNG preparation (~1.5k), last
                // wave forecasts (~2.1k); the Philox blocks (~0.5k cycles per lane-block) fill them up to an
                // even finish: shares in 1/16ths = 8 (wave 1), 5 (wave 2), 3 (last), any further waves take
                // equal parts of wave 1's and wave 2's share
                const int extra = NSH - 3;
                const int c_last = (nblk * 3) / 16;
                const int rest = nblk - c_last;
                // weights 8 : 5 : (6.5 each for extra waves, which carry no fixed job)
                const int den = 2 * (8 + 5) + 13 * extra;
                const int c0 = (rest * 16) / den, c1 = (rest * 10) / den;
                int b0, b1;
                if (shadow_wave == 0) { b0 = 0; b1 = c0; }
                else if (shadow_wave == 1) { b0 = c0; b1 = extra > 0 ? c0 + c1 : rest; }   // no gap when no extra waves
                else if (shadow_wave == NSH - 1) { b0 = rest; b1 = nblk; }
                else {
                    const int per = extra > 0 ? (rest - c0 - c1 + extra - 1) / extra : 0;
                    b0 = c0 + c1 + (shadow_wave - 2) * per;
                    b1 = min(b0 + per, rest);
                }
                if (shadow_wave == 1 && sweep + 1 < p.sweep_end) job_prep(sweep + 1);
                STAMP(15);
                job_uniforms(sweep, b0, b1);
            } else {
                if (shadow_wave == 0 && sweep + 1 < p.sweep_end) job_prep(sweep + 1);
                const int per = (nblk + NSH - 1) / NSH;
                job_uniforms(sweep, shadow_wave * per, min((shadow_wave + 1) * per, nblk));
            }
        }
        STAMP(1);
        __syncthreads();                                                     // Bb: theta[par], ux ready
        STAMP(2);
        // ---- everyone: parameters to registers ----
        double mu[K], isd[K], coef[K], rho[K], A[K][K];
        int order[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            mu[i] = th.mu[i]; isd[i] = th.isd[i]; coef[i] = th.coef[i]; rho[i] = th.rho[i];
#pragma unroll
            for (int j = 0; j < K; ++j) A[i][j] = th.A[i][j];
        }
        sort_order<K>(mu, order);
        double ux[L];
#pragma unroll
        for (int l = 0; l < L; ++l) ux[l] = sh.ux[t0 + l];
        // ---- forward filter (:371-440) as a scan of M_t = A diag(f_t) ----
        double f[L][K];
        unsigned und = 0;                            // bit l: every pdf of step l underflowed
        if constexpr (L >= 8) {
            // eight and more steps per thread: the registers are the scarce resource (the capped flavours spill), so the values
            // are evaluated step by step and the scheduler is left alone (the staged form below cost the 16-step variant 30 %)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                unsigned hm = 0;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double z = (y[l] - mu[s]) * isd[s];          // isd holds 1/(sd*sqrt(2)): exp(-z^2) = exp(-((y-mu)/sd)^2/2)
                    double cf = coef[s];
                    if constexpr (SIG) {                 // signal positions: sd scaled by (1 + kappa) (:382, quirk 4)
                        const bool issig = (t0 + l) >= sb && (t0 + l) < se;
                        z = issig ? z * kfac : z;
                        cf = issig ? cf * kfac : cf;
                    }
                    f[l][s] = exp_tab(-(z * z), sh.exptab, lane & (EXPTAB_C - 1)) * cf;
                    hm = max(hm, (unsigned)__double2hiint(f[l][s]));
                }
                // exact power-of-two scaling of the step: largest pdf into [0.5,1)
                const int e = 1022 - (int)(hm >> 20);
#pragma unroll
                for (int s = 0; s < K; ++s) f[l][s] = ldexp(f[l][s], e);
                und |= (hm < 0x01A56E1Fu) ? (1u << l) : 0u;     // largest pdf < 1e-300 (high word of 1e-300 is 0x01A56E1F)
            }
        } else {
            // The L K emission values in blocks of up to 16, ONE OPERATION AT A TIME across the block (a sched_barrier after
            // each): the instances of an operation are independent, so they issue back to back, and all the block's table reads
            // are in flight together.  Same operations per value as exp_tab: same bits.
            constexpr int LB = (16 / K) < L ? (16 / K) : L;          // steps per block
#define HMCG_EACH for (int g = 0; g < LB; ++g) _Pragma("unroll") for (int s = 0; s < K; ++s)
#define HMCG_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
            for (int l0 = 0; l0 < L; l0 += LB) {
                double x[LB][K], nn[LB][K], tj[LB][K], r[LB][K], pp[LB][K];
                int ni[LB][K];
                auto LL = [&](int g) __attribute__((always_inline)) { return l0 + g < L ? l0 + g : L - 1; };
                auto issig = [&](int g) __attribute__((always_inline)) { return SIG && (t0 + LL(g)) >= sb && (t0 + LL(g)) < se; };
                HMCG_SB;
#pragma unroll
                HMCG_EACH x[g][s] = y[LL(g)] - mu[s];
                HMCG_SB;
#pragma unroll
                HMCG_EACH x[g][s] = x[g][s] * isd[s];
                if constexpr (SIG) {
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH x[g][s] = issig(g) ? x[g][s] * kfac : x[g][s];
                }
                HMCG_SB;
#pragma unroll
                HMCG_EACH x[g][s] = -(x[g][s] * x[g][s]);
                HMCG_SB;
#pragma unroll
                HMCG_EACH x[g][s] = fmax(x[g][s], -746.0);
                HMCG_SB;
#pragma unroll
                HMCG_EACH nn[g][s] = x[g][s] * (EXPTAB_N == 64 ? 92.332482616893657 : 369.3299304675746);
                HMCG_SB;
#pragma unroll
                HMCG_EACH nn[g][s] = rint(nn[g][s]);
                HMCG_SB;
#pragma unroll
                HMCG_EACH ni[g][s] = (int)nn[g][s];
                HMCG_SB;
#pragma unroll
                HMCG_EACH tj[g][s] = exptab_at(sh.exptab, ni[g][s], lane & (EXPTAB_C - 1));
                HMCG_SB;
#pragma unroll
                HMCG_EACH r[g][s] = fma(-nn[g][s], EXPTAB_N == 64 ? 1.0830424693267560e-02 : 2.70760617331689e-03, x[g][s]);
                HMCG_SB;
#pragma unroll
                HMCG_EACH r[g][s] = fma(-nn[g][s], EXPTAB_N == 64 ? 2.9815858269852933e-12 : 7.453964567463233e-13, r[g][s]);
                HMCG_SB;
                if constexpr (EXPTAB_N == 64) {
#pragma unroll
                    HMCG_EACH pp[g][s] = fma(r[g][s], 1.0 / 120.0, 1.0 / 24.0);
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH pp[g][s] = fma(pp[g][s], r[g][s], 1.0 / 6.0);
                } else {
#pragma unroll
                    HMCG_EACH pp[g][s] = fma(r[g][s], 1.0 / 24.0, 1.0 / 6.0);
                }
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = fma(pp[g][s], r[g][s], 0.5);
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = fma(pp[g][s], r[g][s], 1.0);
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = fma(pp[g][s], r[g][s], 1.0);
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = tj[g][s] * pp[g][s];
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = ldexp(pp[g][s], ni[g][s] >> (EXPTAB_N == 64 ? 6 : 8));
                HMCG_SB;
#pragma unroll
                HMCG_EACH pp[g][s] = pp[g][s] * (issig(g) ? coef[s] * kfac : coef[s]);
                HMCG_SB;
#pragma unroll
                for (int g = 0; g < LB; ++g) {
                    if (l0 + g < L) {
                        unsigned hm = 0;
#pragma unroll
                        for (int s = 0; s < K; ++s) hm = max(hm, (unsigned)__double2hiint(pp[g][s]));
                        const int e = 1022 - (int)(hm >> 20);
#pragma unroll
                        for (int s = 0; s < K; ++s) f[l0 + g][s] = ldexp(pp[g][s], e);
                        und |= (hm < 0x01A56E1Fu) ? (1u << (l0 + g)) : 0u;
                    }
                }
                HMCG_SB;
            }
#undef HMCG_EACH
#undef HMCG_SB
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(und != 0u) != 0ull, 0)) {
            // (rare, wave-uniform branch) every pdf of some step underflowed: treat the observation as missing (f = 1)
            // and flag the window -- the reference would produce NaN and throw (src/Hmc.jl:435)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if ((und >> l) & 1u) {
                    if (t0 + l < T) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                    for (int s = 0; s < K; ++s) f[l][s] = 1.0;
                }
            }
        }
        if constexpr (SIG) {
            if (tail > 0) {                          // emission values of the steps after end_pos, for the outputs job
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int j = t0 + l - (T - tail);
                    if (j >= 0 && j < tail) {
#pragma unroll
                        for (int s = 0; s < K; ++s) sh.ftail[par][j][s] = f[l][s];
                    }
                }
            }
        }
        STAMP(3);
        // local product Q = M_{t0} ... M_{t0+L-1}
        // (padded steps t >= T take part with y = 0: everything they influence lies at or beyond T,
        //  where no result is consumed -- prefixes only flow forward in time)
        double Q[KK];
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s) {
                Q[r * K + s] = A[r][s] * f[0][s];
                if constexpr (SMOOTH) Q[r * K + s] = (t0 < T) ? Q[r * K + s] : ((r == s) ? 1.0 : 0.0);
            }
#pragma unroll
        for (int l = 1; l < L; ++l) {
            double N[KK];
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double acc = Q[r * K] * A[0][s];
#pragma unroll
                    for (int k = 1; k < K; ++k) acc = fma(Q[r * K + k], A[k][s], acc);
                    N[r * K + s] = acc * f[l][s];
                }
            if constexpr (SMOOTH) {                  // the suffix products must not see padded steps
                const bool v = (t0 + l) < T;
#pragma unroll
                for (int i = 0; i < KK; ++i) Q[i] = v ? N[i] : Q[i];
            } else {
#pragma unroll
                for (int i = 0; i < KK; ++i) Q[i] = N[i];
            }
        }
        // (every step was scaled so that its largest pdf lies in [0.5, 1): up to four steps per thread and two scan levels
        //  -- sixteen factors -- stay far inside the fp64 range without help; longer chunks are brought back here)
        if constexpr (L > 4 || SMOOTH) rescale_pow2<KK>(Q);
        double Qloc[SMOOTH ? KK : 1];                // this thread's own product, for the backward (suffix) scan
        if constexpr (SMOOTH) {
#pragma unroll
            for (int i = 0; i < KK; ++i) Qloc[i] = Q[i];
        }
        (void)Qloc;
        STAMP(4);
        // inclusive scan over the wave (earlier lanes multiply on the left); the per-step scaling
        // keeps every factor's largest entry in [0.5,1), so rescaling every other level is ample
        scan_level<K, DPP_ROW_SHR1, 0xF>(Q);
        scan_level<K, DPP_ROW_SHR2, 0xF>(Q);
        rescale_pow2<KK>(Q);
        scan_level<K, DPP_ROW_SHR4, 0xF>(Q);
        scan_level<K, DPP_ROW_SHR8, 0xF>(Q);
        rescale_pow2<KK>(Q);
        scan_level<K, DPP_ROW_BCAST15, 0xA>(Q);
        scan_level<K, DPP_ROW_BCAST31, 0xC>(Q);
        rescale_pow2<KK>(Q);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < KK; ++i) sh.wtot[wave][i] = Q[i];
        }
        STAMP(5);
        __syncthreads();                                                     // Bc
        STAMP(6);
        // prefix vector: rho' * (totals of earlier waves) * (exclusive lane prefix)
        double av[K];
#pragma unroll
        for (int s = 0; s < K; ++s) av[s] = rho[s];
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // scalar: the loops over other waves branch, not select
        // (the last wave runs NW-1 dependent vector-matrix products here and everyone waits for it at the next barrier:
        //  the next wave total is fetched from LDS while the current product runs)
        double Wn[KK];
#pragma unroll
        for (int i = 0; i < KK; ++i) Wn[i] = sh.wtot[0][i];
#pragma unroll
        for (int ww = 0; ww < NW - 1; ++ww) {
            if (ww < wave_u) {
                double Wc[KK], nv[K];
#pragma unroll
                for (int i = 0; i < KK; ++i) Wc[i] = Wn[i];
                if (ww + 1 < NW - 1) {
#pragma unroll
                    for (int i = 0; i < KK; ++i) Wn[i] = sh.wtot[ww + 1][i];
                }
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    double acc = av[0] * Wc[s];
#pragma unroll
                    for (int r = 1; r < K; ++r) acc = fma(av[r], Wc[r * K + s], acc);
                    nv[s] = acc;
                }
#pragma unroll
                for (int s = 0; s < K; ++s) av[s] = nv[s];
            }
        }
        {
            double nv[K];
#pragma unroll
            for (int s = 0; s < K; ++s) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < K; ++r)
                    acc = fma(av[r], dpp_f64<DPP_WAVE_SHR1, 0xF>((r == s) ? 1.0 : 0.0, Q[r * K + s]), acc);
                nv[s] = acc;
            }
            rescale_pow2<K>(nv);
#pragma unroll
            for (int s = 0; s < K; ++s) av[s] = nv[s];
        }
        // replay the normalised recursion over this thread's steps (:413-432)
        // (carrying the UNNORMALISED vector and normalising each step off the chain was measured: +-0)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            double nv[K], total = 0.0;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                double acc = av[0] * A[0][s];
#pragma unroll
                for (int r = 1; r < K; ++r) acc = fma(av[r], A[r][s], acc);
                nv[s] = acc * f[l][s];
                total += nv[s];
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(total > 0.0)) != 0ull, 0)) {
                // not reachable with finite positive parameters; kept as a guard (wave-uniform branch, never taken)
                if (!(total > 0.0)) {
                    if (t0 + l < T) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                    for (int s = 0; s < K; ++s) nv[s] = 1.0 / K;
                    total = 1.0;
                }
            }
            const double inv = rcp_fast(total);
#pragma unroll
            for (int s = 0; s < K; ++s) { av[s] = nv[s] * inv; pf[l][s] = av[s]; }
        }
#pragma unroll
        for (int s = 0; s < K; ++s) sh.pfirst[tid][s] = pf[0][s];
        // owner of the last step publishes pif[T-1,:] (unsorted) and its uniform; X[T-1] itself (sorted
        // labels, :464) is then drawn redundantly by every thread after the barrier (no divergent tail)
        if (tid == owner) {
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (l == l_last) {
#pragma unroll
                    for (int s = 0; s < K; ++s) th.pi_end[s] = pf[l][s];
                    sh.ulast = ux[l];
                }
        }
        if constexpr (SIG) {
            if (tail > 0) {
#pragma unroll
                for (int l = 0; l < L; ++l)
                    if (t0 + l == T - 1 - tail) {
#pragma unroll
                        for (int s = 0; s < K; ++s) sh.pf_rep[par][s] = pf[l][s];
                    }
            }
        }
        if constexpr (SMOOTH) {
            // ---- backwardupdate_P! (src/Hmc.jl:442-457) as the beta recursion b_{t-1} = A (f_t o b_t), b_{T-1} = 1:
            // pib[t,:] ~ pif[t,:] o b_t.  b at the end of a thread's chunk = (product of the later chunks' matrices) * 1:
            // a SUFFIX scan of the local products -- done as a prefix scan on lane-reversed data with the
            // multiplication order flipped -- then the later waves' totals (the forward wave totals, already in LDS).
            double R[KK];
#pragma unroll
            for (int i = 0; i < KK; ++i) R[i] = __shfl(Qloc[i], 63 - lane, 64);
            scan_level_rev<K, DPP_ROW_SHR1, 0xF>(R);
            scan_level_rev<K, DPP_ROW_SHR2, 0xF>(R);
            rescale_pow2<KK>(R);
            scan_level_rev<K, DPP_ROW_SHR4, 0xF>(R);
            scan_level_rev<K, DPP_ROW_SHR8, 0xF>(R);
            rescale_pow2<KK>(R);
            scan_level_rev<K, DPP_ROW_BCAST15, 0xA>(R);
            scan_level_rev<K, DPP_ROW_BCAST31, 0xC>(R);
            rescale_pow2<KK>(R);
            // vector entering this wave from the later ones: W_{wave+1} (W_{wave+2} (... 1))
            double bw[K];
#pragma unroll
            for (int r = 0; r < K; ++r) bw[r] = 1.0;
#pragma unroll
            for (int ww = NW - 1; ww >= 1; --ww) {
                double nb[K];
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int c = 0; c < K; ++c) acc = fma(sh.wtot[ww][r * K + c], bw[c], acc);
                    nb[r] = acc;
                }
                rescale_pow2<K>(nb);
#pragma unroll
                for (int r = 0; r < K; ++r) bw[r] = (ww > wave) ? nb[r] : bw[r];
            }
            // exclusive suffix of lane j = inclusive result held by reversed lane (63-j)-1, i.e. physical lane 62-j
            double b[K];
#pragma unroll
            for (int r = 0; r < K; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const double e = __shfl(R[r * K + c], lane < 63 ? 62 - lane : 0, 64);
                    acc = fma((lane < 63) ? e : ((r == c) ? 1.0 : 0.0), bw[c], acc);
                }
                b[r] = acc;
            }
            rescale_pow2<K>(b);
            const int dk = SIG ? kept_index(p, sweep) : (sweep >= p.burnin_s ? sweep - p.burnin_s : -1);     // kept-draw index
            const bool kept = dk >= 0;
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                if (t0 + l < T) {
                    double g[K], tot = 0.0;
#pragma unroll
                    for (int s = 0; s < K; ++s) { g[s] = pf[l][s] * b[s]; tot += g[s]; }
                    const double inv = rcp_fast(tot);
                    if (kept) {
#pragma unroll
                        for (int q = 0; q < K; ++q) {
                            double gq = 0.0;
#pragma unroll
                            for (int s = 0; s < K; ++s) gq = (order[q] == s) ? g[s] : gq;
                            sm_acc[l][q] = fma(gq, inv, sm_acc[l][q]);      // sorted labels (:513)
                            if (p.pi_smooth_draws)                          // samples.pib[d, t, q] itself (:558)
                                p.pi_smooth_draws[(size_t)p.nd_ld * ((size_t)q * p.ldY + (t0 + l) + (size_t)K * p.ldY * w) + (dk - p.draw_off)] = gq * inv;
                            double fq = 0.0;
#pragma unroll
                            for (int s = 0; s < K; ++s) fq = (order[q] == s) ? pf[l][s] : fq;
                            fm_acc[l][q] += fq;                              // sorted pif[t,:] (:512)
                        }
                    }
                    double nb[K];
#pragma unroll
                    for (int r = 0; r < K; ++r) {
                        double acc = 0.0;
#pragma unroll
                        for (int s = 0; s < K; ++s) acc = fma(A[r][s] * f[l][s], b[s], acc);
                        nb[r] = acc;
                    }
                    rescale_pow2<K>(nb);
#pragma unroll
                    for (int r = 0; r < K; ++r) b[r] = nb[r];
                }
            }
        }
        STAMP(7);
        __syncthreads();                                                     // Bd
        STAMP(8);
        // ---- backward sampling (:459-484) as a suffix scan of state maps ----
        int xlast = 0;
        {
            const double ulast = sh.ulast;
            double cp = 0.0;
#pragma unroll
            for (int q = 0; q < K - 1; ++q) {
                double pq = 0.0;
#pragma unroll
                for (int s = 0; s < K; ++s) pq = (order[q] == s) ? th.pi_end[s] : pq;
                cp += pq;
                xlast += (cp <= ulast) ? 1 : 0;
            }
        }
        double pfn_last[K];
#pragma unroll
        for (int s = 0; s < K; ++s) pfn_last[s] = sh.pfirst[(tid + 1 < NT) ? tid + 1 : tid][s];
        uint32_t gmap[L];
        uint32_t G = BMAP_IDENTITY;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const int t = t0 + l;
            uint32_t m = 0;
            // the draw under the uniform 1/K law (a failed guard, :476-480) does not depend on the successor state: once per step
            int idx_uni = 0;
            {
                double ucum = 0.0;
#pragma unroll
                for (int r = 0; r < K - 1; ++r) { ucum += 1.0 / K; idx_uni += (ucum <= ux[l]) ? 1 : 0; }
            }
#pragma unroll
            for (int s = 0; s < K; ++s) {
                // p[r] = Pf[t+1,r,s] is proportional to pif[t,r] * A[r,s]; its sum over r equals the
                // unsorted pif[t+1,s], which drives the eps() guard (:472, quirk 7).  A failed guard
                // (common for states far from y[t+1]) selects the uniform 1/K law (:476-480).
                const double guard = (l + 1 < L) ? pf[(l + 1 < L) ? l + 1 : l][s] : pfn_last[s];
                double tot = 0.0;
                int idx = 0;
                double cum[K];
#pragma unroll
                for (int r = 0; r < K; ++r) { tot = fma(pf[l][r], A[r][s], tot); cum[r] = tot; }
                const double thr = ux[l] * tot;
#pragma unroll
                for (int r = 0; r < K - 1; ++r) idx += (cum[r] <= thr) ? 1 : 0;
                idx = (guard > EPS64) ? idx : idx_uni;          // one integer select instead of selecting every threshold
                m |= (uint32_t)idx << (8 * s);
            }
            if constexpr (K < 4) m |= BMAP_IDENTITY & (0xFFFFFFFFu << (8 * K));       // unused bytes: identity
            m = (t == T - 1) ? bmap_const(xlast) : (t > T - 1 ? (PADMARK ? bmap_const(XPAD) : BMAP_IDENTITY) : m);
            gmap[l] = m;
            G = bmap_compose(m, G);      // G = g_{t0+l} o (g_{t0+l+1} o ...)
        }
        STAMP(9);
        // inclusive suffix scan over lanes: H_lane = G_lane o G_{lane+1} o ... o G_63.  Inside a 16-lane row by DPP
        // row_shl (a lane whose source lies beyond its row receives the identity); across the four rows through the
        // row totals (first lane of each row), read as scalars.
        uint32_t Hm = G;
        Hm = bmap_compose(Hm, (uint32_t)dpp_i32<DPP_ROW_SHL1, 0xF>((int)BMAP_IDENTITY, (int)Hm));
        Hm = bmap_compose(Hm, (uint32_t)dpp_i32<DPP_ROW_SHL2, 0xF>((int)BMAP_IDENTITY, (int)Hm));
        Hm = bmap_compose(Hm, (uint32_t)dpp_i32<DPP_ROW_SHL4, 0xF>((int)BMAP_IDENTITY, (int)Hm));
        Hm = bmap_compose(Hm, (uint32_t)dpp_i32<DPP_ROW_SHL8, 0xF>((int)BMAP_IDENTITY, (int)Hm));
        {
            const uint32_t R1 = (uint32_t)__builtin_amdgcn_readlane((int)Hm, 16);
            const uint32_t R2 = (uint32_t)__builtin_amdgcn_readlane((int)Hm, 32);
            const uint32_t R3 = (uint32_t)__builtin_amdgcn_readlane((int)Hm, 48);
            const uint32_t S1 = bmap_compose(R2, R3), S0 = bmap_compose(R1, S1);
            const int row = lane >> 4;
            const uint32_t after = row == 0 ? S0 : (row == 1 ? S1 : (row == 2 ? R3 : BMAP_IDENTITY));
            Hm = bmap_compose(Hm, after);
        }
        if (lane == 0) sh.wmap[wave] = Hm;
        STAMP(10);
        __syncthreads();                                                     // Be
        STAMP(11);
        uint32_t Rw = BMAP_IDENTITY;
#pragma unroll
        for (int ww = NW - 1; ww >= 1; --ww) if (ww > wave_u) Rw = bmap_compose(sh.wmap[ww], Rw);
        const uint32_t Hx = (uint32_t)dpp_i32<DPP_WAVE_SHL1, 0xF>((int)BMAP_IDENTITY, (int)Hm);   // lane 63: identity
        const uint32_t Sfx = bmap_compose(Hx, Rw);     // everything after this thread's chunk
        int sin = bmap_apply(Sfx, XPAD);               // constant map below T-1; the identity (-> XPAD) for the last thread
        xnext = sin;
        x_end = xlast;
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            sin = bmap_apply(gmap[l], sin);      // (slots at or beyond T: XPAD when PADMARK, else a value that is never read)
            x[l] = sin;
        }
        // pivots for the next sweep's one-pass statistics: this sweep's state means (X labels are unsorted)
#pragma unroll
        for (int k = 0; k < K; ++k) pivot[k] = mu[k];
        STAMP(12);
        publish_stats();
        STAMP(13);
    }
#ifdef HMCG_STAMPS
    if (lane == 0 && p.dbg) {
        unsigned long long* o = p.dbg + ((size_t)w * (NW + NH) + wave) * HMCG_NSTAMP_ALL;
        for (int i = 0; i < HMCG_NSTAMP; ++i) o[i] = stamp_acc[i];
        o[HMCG_NSTAMP] = __builtin_amdgcn_s_memtime() - stamp_t0;             // in-kernel clock = ticks / realtime ticks * 100 MHz
        o[HMCG_NSTAMP + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif

    // ---- epilogue: the last sweep's outputs, checkpoint / debug outputs ----
    __syncthreads();
    if (p.sweep_end > p.sweep_begin) job_outputs(p.sweep_end - 1);
    if (p.xstate) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (t0 + l < T) p.xstate[(size_t)w * p.ldY + t0 + l] = (uint8_t)x[l];
    }
    if (p.x_final) {
#pragma unroll
        for (int l = 0; l < L; ++l) if (t0 + l < T) p.x_final[(size_t)w * p.ldY + t0 + l] = x[l];
    }
    if (p.pif_final && p.sweep_end > p.sweep_begin) {
#pragma unroll
        for (int l = 0; l < L; ++l)
            if (t0 + l < T)
#pragma unroll
                for (int s = 0; s < K; ++s) p.pif_final[((size_t)w * p.ldY + t0 + l) * K + s] = pf[l][s];
    }
    if constexpr (SMOOTH) {
        if (p.pi_filter_mean) {
            const double sc = (p.final_launch && p.nd > 0) ? 1.0 / (double)p.nd : 1.0;
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (t0 + l < T)
#pragma unroll
                    for (int q = 0; q < K; ++q) p.pi_filter_mean[((size_t)w * p.ldY + t0 + l) * K + q] = fm_acc[l][q] * sc;
        }
        if (p.pi_smooth_mean) {
            const double sc = (p.final_launch && p.nd > 0) ? 1.0 / (double)p.nd : 1.0;
#pragma unroll
            for (int l = 0; l < L; ++l)
                if (t0 + l < T)
#pragma unroll
                    for (int q = 0; q < K; ++q) p.pi_smooth_mean[((size_t)w * p.ldY + t0 + l) * K + q] = sm_acc[l][q] * sc;
        }
    }
    {
        const int orole = role_of(lane);
        if (orole >= 0) {
            if (p.sumacc) p.sumacc[(size_t)w * NCK + orole] = sum_acc;
            if (p.summary && p.final_launch)
                p.summary[(size_t)w * NS + orole] = p.nd > 0 ? sum_acc / (double)p.nd : __builtin_nan("");
            if constexpr (SIG) {
                // a launch that stops inside a noise sample leaves that sample's running sums in its row (checkpoint)
                if (p.sample_summary) {
                    const int smp = p.sweep_end / p.per_sample, kb = p.sweep_end - smp * p.per_sample - p.burnin_s;
                    if (kb > 0 && kb < p.nrun_s && smp < p.n_samples) p.sample_summary[((size_t)w * p.n_samples + smp) * NS + orole] = smp_acc;
                }
            }
        }
    }
    if (p.sumacc) {
#pragma unroll
        for (int k = 0; k < K; ++k) if (tid == k) p.sumacc[(size_t)w * NCK + (NCK - K) + k] = pivot[k];
    }
    if (st) atomicOr(&p.status[w], st);
}

}  // namespace hmcg
