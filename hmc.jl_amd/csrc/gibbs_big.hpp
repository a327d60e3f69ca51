// gibbs_big.hpp -- large-K variant (5 <= K <= 8) of the persistent per-window Gibbs kernel, for
// BASELINE configs[3] (8 states, T = 5000).  Same phases, barriers and RNG spec as the small-K kernel
// (gibbs_device.hpp); what changes is where things live and how the loops are shaped:
//   * a K x K matrix is 64 doubles, so per-step data cannot sit in registers: the window's Y, X, the
//     sweep's uniforms and the per-step state maps live in (dynamic) LDS, and the loops over a
//     thread's L = ceil(T/256) consecutive steps are runtime loops;
//   * the transition matrix is read from LDS, one column at a time (stored transposed so a column is
//     contiguous); a step's K pdfs are evaluated once, in the product phase, and reach the replay (and the smoothing
//     pass) through a lane-contiguous HBM scratch, KernelParams::fscr;
//   * the state map g_{t-1} is built inside the filter replay at step t: the running sums
//     sum_{r' <= r} pif[t-1,r'] A[r',s] that give pif[t,s] are exactly the cumulative weights of the
//     draw X[t-1] | X[t] = s (src/Hmc.jl:468-481), and the eps() guard pif[t,s] is at hand -- so
//     the filtered probabilities never have to be stored (pif is neither kept in registers nor
//     written to HBM; only pif[T-1,:] leaves the replay);
//   * transition counts go through per-wave LDS histograms (integer atomics: order-independent),
//     parameter-draw roles (K + K^2 = 72 at K = 8) and output roles take more than one pass over a wave;
//   * the state maps are 4-bit per entry in LDS and widened to a byte per entry (two words, composed by two
//     v_perm_b32) for the backward pass; runtime loops whose iterations start with an LDS read are pipelined by hand.
// Reference lines: as in gibbs_device.hpp.
#pragma once
#include "gibbs_device.hpp"

namespace hmcg {

template <int K>
struct alignas(16) ThetaBufBig {       // (16-byte rows: the assembly product loop reads A by ds_read_b128)
    double mu[K], sig2[K], isd[K], coef[K], rho[K];
    double A[K][K];               // row-major, unsorted labels
    double At[K][K];              // At[s][r] = A[r][s]: column s contiguous
    double pi_end[K];
};

template <int K, int NT>
struct BigShared {
    static constexpr int NW = NT / 64;
    static constexpr int KK = K * K;
    static constexpr int NG = K + KK;
    unsigned cnt[NW][KK];         // per-wave transition histograms C_ij (field i*K+j)
    double red_d1[NW][K], red_d2[NW][K];
    double pivot[K];
    int x_end;
    ThetaBufBig<K> th[2];
    RngBuf<K> rb[2];
    double gval[NG];              // gamma variates of the running parameter phase
    double wtot[NW][KK];
    double rtot[NW][4][KK];       // cooperative scan: the four 16-lane row totals of every wave
    double ptmp[NW][2][KK];       // ... and the partial products on the way to the wave total
    uint32_t wmap[NW][2];         // per-wave composed state map, byte-per-entry in two words
    double ulast;
    double bred[NW];
    int selcnt[2][8][3];          // block_select (init)
    double fcM[2][K * K];         // forecast scratch (cooperative matrix power on the forecast wave)
    double fcv[2][K];
    double fcval[HMCG_MAXH];
    double exptab[EXPTAB_N * EXPTAB_C];   // 2^(j/N): the replicated table of exp_tab (gibbs_device.hpp)
    // signal path (SIG): signal steps by state, the pivoted sums over the signal positions, the last noisy observation of
    // the running sample (forecastsignal's `signal`) and the filtered probabilities at end_pos, by sweep parity
    unsigned cntm[NW][K];
    double red_s1[NW][K], red_s2[NW][K];
    double y_last[2];
    double pf_rep[2][K];
};

// SM: additionally run the backward pass (backwardupdate_P!, src/Hmc.jl:442-457) on every kept sweep and accumulate the
// smoothed and the filtered probabilities of every step (sorted labels) into p.pi_smooth_mean / p.pi_filter_mean -- the
// large-K / long-window counterpart of the SMOOTH variants of gibbs_device.hpp.  The filtered probabilities of the
// running sweep pass through p.pif_final (written by the forward replay, read back by the backward pass), the running
// sums live in HBM (T x K doubles per window do not fit on the chip: this variant genuinely streams them).
// Byte-per-entry state maps for up to eight states, in two words: entry s is byte s & 3 of (s < 4 ? lo : hi).
struct ByteMap { uint32_t lo, hi; };
__device__ __forceinline__ ByteMap bytemap_identity() { return ByteMap{0x03020100u, 0x07060504u}; }
// (a o b)[s] = a[b[s]]: v_perm_b32's selector values 0..3 take bytes of its second source, 4..7 of its first
__device__ __forceinline__ ByteMap bytemap_compose(const ByteMap a, const ByteMap b)
{
    return ByteMap{__builtin_amdgcn_perm(a.hi, a.lo, b.lo), __builtin_amdgcn_perm(a.hi, a.lo, b.hi)};
}
__device__ __forceinline__ uint32_t spread_nibbles(uint32_t x)      // 0x0000dcba -> 0x0d0c0b0a
{
    x = (x | (x << 8)) & 0x00FF00FFu;
    return (x | (x << 4)) & 0x0F0F0F0Fu;
}
__device__ __forceinline__ ByteMap bytemap_from_nibbles(uint32_t m) { return ByteMap{spread_nibbles(m & 0xFFFFu), spread_nibbles(m >> 16)}; }

// #{i < N : c[i] <= thr} for a non-decreasing c (cumulative sums of non-negative terms), N <= 7: a three-level bisection
// -- 3 compares and 8 selects instead of N compare + add-with-carry pairs.  The same predicate as the linear count
// (Categorical's CDF scan, src/Hmc.jl:481), so the draws do not change.
template <int N>
__device__ __forceinline__ int count_le_sorted(const double (&c)[N + 1], double thr)
{
    static_assert(N >= 1 && N <= 7, "three levels");
    constexpr double INF = __builtin_huge_val();
    auto C = [&](int i) __attribute__((always_inline)) { return i < N ? c[i < N ? i : 0] : INF; };
    const bool b1 = C(3) <= thr;
    const double m2 = b1 ? C(5) : C(1);
    const double lo3 = b1 ? C(4) : C(0), hi3 = b1 ? C(6) : C(2);
    const bool b2 = m2 <= thr;
    const double m3 = b2 ? hi3 : lo3;
    const bool b3 = m3 <= thr;
    return (b1 ? 4 : 0) + (b2 ? 2 : 0) + (b3 ? 1 : 0);
}

// The same for K searches at once, level by level: idx[s] = #{i < K-1 : c[s][i] <= thr[s]}.  Written so that the K
// compares of a level are independent instructions (each into its own condition register) instead of K serial
// compare -> select chains through VCC.
template <int K>
__device__ __forceinline__ void count_le_sorted_batch(const double (&c)[K][K], const double (&thr)[K], int (&idx)[K])
{
    constexpr int N = K - 1;
    static_assert(N >= 1 && N <= 7, "three levels");
    constexpr double INF = __builtin_huge_val();
    auto C = [&](int s, int i) __attribute__((always_inline)) { return i < N ? c[s][i < N ? i : 0] : INF; };
    bool b1[K], b2[K];
    double m2[K], lo3[K], hi3[K], m3[K];
#pragma unroll
    for (int s = 0; s < K; ++s) b1[s] = C(s, 3) <= thr[s];
#pragma unroll
    for (int s = 0; s < K; ++s) { m2[s] = b1[s] ? C(s, 5) : C(s, 1); lo3[s] = b1[s] ? C(s, 4) : C(s, 0); hi3[s] = b1[s] ? C(s, 6) : C(s, 2); }
#pragma unroll
    for (int s = 0; s < K; ++s) b2[s] = m2[s] <= thr[s];
#pragma unroll
    for (int s = 0; s < K; ++s) m3[s] = b2[s] ? hi3[s] : lo3[s];
#pragma unroll
    for (int s = 0; s < K; ++s) idx[s] = (b1[s] ? 4 : 0) + (b2[s] ? 2 : 0) + ((m3[s] <= thr[s]) ? 1 : 0);
}

// A value that is the same in every lane, moved to scalar registers: as an operand of a VALU instruction it then costs no
// vector register and no LDS read (one scalar operand per instruction: the constant-bus limit of gfx9).
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Where the per-step arrays (observations, uniforms, state maps, states) live: LDS, or -- STREAM -- the window's slab of an
// HBM scratch (global address space, so that the accesses are global_*, never flat_*): the same code either way.
template <bool GLOBAL, class T> struct StepPtr { using type = T*; };
template <class T> struct StepPtr<true, T> { using type = __attribute__((address_space(1))) T*; };

// STREAM: the window is longer than the CU's LDS holds (T beyond ~5 500 at K = 8, ~6 000 at K = 3): Y, the sweep's uniforms,
// the state maps and the states stream through HBM / L2 (each lane walks its own L consecutive steps, so a cache line
// serves eight of them); slower per step than the LDS-resident form, but no longer refused (the reference's loops are
// unbounded in N, src/Hmc.jl:406).
// SIG: the signal Monte-Carlo path (estimatesignals!, src/Hmc.jl:868-914) as in the SIG variants of gibbs_device.hpp -- noise
// samples chained inside the launch on Yfake = Yreal + N(0,1) sigma_signal over the signal range, two-population statistics
// (observation set / signal set, :254-314), signal emission sd (1 + kappa) sqrt(sigma) (:382), signals past the end date
// (smoothed probabilities at end_pos :900, forecastsignal :670-681) and the per-sample summaries.
template <int K, int NT, bool SM = false, bool STREAM = false, bool SIG = false>
__global__ __launch_bounds__(NT) void gibbs_sweeps_kernel_big(const KernelParams p, const int L)
{
    static_assert(K >= 2 && K <= 8, "4-bit map entries: K <= 8 (K <= 4 normally runs on the register-resident kernel; this one\n"
                                    "also serves small K when the window is too long for it)");
    constexpr int NW = NT / 64;
    constexpr int KK = K * K;
    constexpr int NG = K + KK;
    static_assert(NW >= 2, "needs shadow waves");
    using Sh = BigShared<K, NT>;
    __shared__ Sh sh;
    extern __shared__ double dyn_lds[];
    const int cap = NT * L;
    using DPtr = typename StepPtr<STREAM, double>::type;
    using UPtr = typename StepPtr<STREAM, uint32_t>::type;
    using BPtr = typename StepPtr<STREAM, uint8_t>::type;
    DPtr ylds;                                                      // [cap] observations
    if constexpr (STREAM) ylds = (DPtr)(p.sscr + (size_t)blockIdx.x * (size_t)p.stream_stride);
    else ylds = (DPtr)dyn_lds;
    const DPtr uxs = ylds + cap;                                    // [cap] uniforms of the running sweep
    const UPtr maps = (UPtr)(ylds + 2 * (size_t)cap);               // [cap] state maps g_t
    const BPtr xs = (BPtr)(maps + cap);                             // [cap + 8] states
    // HBM scratch of the window's per-step pdfs, p.fscr is [W][L][KP][NT][2]: the K values of a step as KP = ceil(K/2) PAIRS,
    // the thread index next (a wave's access to a pair is one 1024-byte dwordx4 instruction: half as many vector-memory
    // instructions as a value at a time -- each costs the lone wave ~50 ticks of issue)
    constexpr int KP = big_scratch_pairs(K);
    double* const fscr_w = p.fscr + (size_t)blockIdx.x * L * (2 * KP) * NT;
    auto fslot = [&](int l, int s, int thread) __attribute__((always_inline)) -> size_t {
        return (((size_t)l * KP + (size_t)(s >> 1)) * NT + (size_t)thread) * 2 + (size_t)(s & 1);
    };
    double* const fscr = fscr_w;          // (indexed through fslot(l, s, threadIdx.x))

    const int w = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T[w];
    const int t0 = tid * L;
    int st = 0;
    if (T < 2 || T > cap || T > p.ldY) {
        if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_T);
        return;
    }

    // signal path: positions [sb, se) are signals; ylds then holds the running noise sample's Yfake, p.Y the data
    int sb = T, se = T, tail = 0;
    const double kfac = SIG ? 1.0 / (1.0 + p.kappa) : 1.0;
    if constexpr (SIG) {
        if (p.sig_range) { sb = p.sig_range[2 * w]; se = p.sig_range[2 * w + 1]; }
        bool bad_range = sb < 0 || se > T || (sb < se && se != T);          // caller data; uniform per block
        if (p.save_range) {
            const int svb = p.save_range[2 * w], sve = p.save_range[2 * w + 1];
            bad_range |= svb < 0 || sve > T || (svb < sve && p.sigvals && sve - svb > p.nsave_ld);
        }
        if (bad_range) {
            if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_RANGE);
            return;
        }
        if (sb >= se) { sb = T; se = T; }
        if (p.end_pos) tail = (T - 1) - p.end_pos[w];       // steps after the position whose smoothed probabilities are reported
        if (tail < 0 || tail > HMCG_MAXTAIL || tail > T - 1) {
            if (tid == 0) flag_skipped(p, w, HMCG_ST_BAD_T);
            return;
        }
    }
    (void)sb; (void)se; (void)tail; (void)kfac;

    exptab_fill(sh.exptab, tid, NT);
    // ---- observations into LDS (coalesced), xi = mean(Y) (src/Hmc.jl:136; always the mean of the REAL window) ----
    bool bad = false;
    double part = 0.0;
    for (int t = tid; t < cap; t += NT) {
        const double v = t < T ? p.Y[(size_t)w * p.ldY + t] : 0.0;
        bad |= t < T && !isfinite(v);
        ylds[t] = v;
        part += v;
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (tid == 0) flag_skipped(p, w, HMCG_ST_NONFINITE);
        return;
    }
    const double xi = block_sum<NW>(part, sh.bred, wave, lane) / (double)T;
    const int NS = 3 * K + KK + 2 * p.H;
    const int NCK = NS + K;

    if (tid < K) sh.pivot[tid] = xi;
    if (p.resume) {
        for (int t = tid; t < cap; t += NT) xs[t] = t < T ? (uint8_t)min((int)p.xstate[(size_t)w * p.ldY + t], K - 1) : 0;
        if (p.sumacc && tid < K) sh.pivot[tid] = p.sumacc[(size_t)w * NCK + NS + tid];
    } else if (p.x_init) {
        for (int t = tid; t < cap; t += NT) xs[t] = t < T ? (uint8_t)min(max(p.x_init[(size_t)w * p.ldY + t], 0), K - 1) : 0;   // caller data: clamp
    } else {
        // makeParams (src/Hmc.jl:161-195): see the small-K kernel for the nearest-mean rule
        double lmin = 1.0e308, lmax = -1.0e308;
        for (int t = tid; t < T; t += NT) { lmin = fmin(lmin, ylds[t]); lmax = fmax(lmax, ylds[t]); }
        const double ymin = block_minmax<NW>(lmin, sh.bred, wave, lane, false);
        const double ymax = block_minmax<NW>(lmax, sh.bred, wave, lane, true);
        // median: the lower middle order statistic by radix selection over the LDS-resident window (gibbs_device.hpp,
        // block_select); for an even T the upper one from one more count and a minimum
        const int rlo = (T - 1) / 2;
        const double med_lo = key_value(block_select<NW, 0>(L, [&](int i, unsigned long long& k) __attribute__((always_inline)) {
                                                                const int t = tid + i * NT;
                                                                k = order_key(ylds[t < T ? t : 0]);
                                                                return t < T;
                                                            }, rlo, sh.selcnt, wave, lane));
        double med_hi = med_lo;
        if (!(T & 1)) {                  // uniform
            double nle = 0.0, above = 1.0e308;
            for (int t = tid; t < T; t += NT) {
                if (ylds[t] <= med_lo) nle += 1.0;
                else above = fmin(above, ylds[t]);
            }
            const double cle = block_sum<NW>(nle, sh.bred, wave, lane);
            const double nxt = block_minmax<NW>(above, sh.bred, wave, lane, false);
            med_hi = (cle > (double)(rlo + 1)) ? med_lo : nxt;
        }
        {
#pragma clang fp contract(off)
            const double med = (T & 1) ? med_lo : med_lo / 2 + med_hi / 2;
            const double R = ymax - ymin;
            const double lo = med - 0.25 * R, hi = med + 0.25 * R;
            double mu0[K];
#pragma unroll
            for (int k = 0; k < K; ++k) mu0[k] = lo + (hi - lo) * ((double)k / (double)(K - 1));
            mu0[K - 1] = hi;
            for (int t = tid; t < cap; t += NT) {
                int best = 0;
                double bd = fabs(ylds[t] - mu0[0]);
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const double d = fabs(ylds[t] - mu0[k]);
                    if (d < bd) { bd = d; best = k; }
                }
                xs[t] = t < T ? (uint8_t)best : 0;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) if (tid == k) sh.pivot[k] = mu0[k];
        }
    }
    if (tid < 8) xs[cap + tid] = 0;
    __syncthreads();
    int x_end = xs[T - 1];

    Rng rng{p.seed_lo, p.seed_hi, p.window_ids ? p.window_ids[w] : p.window_base + (uint32_t)w, 0u};
    const int shadow_wave = wave - 1;
    constexpr int NSH = NW - 1;

    // ---- shadow jobs -------------------------------------------------------------------------
    auto job_prep = [&](int sw) __attribute__((always_inline)) {                    // see gibbs_device.hpp job_prep
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
            const int gt = t_gx ? task - K : task - (2 * K + 2 * NG);
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
            if (t_gx || t_z) val = sqrt_fast(-2.0 * lg) * cos2pi_fast(u53(r[2], r[3]));
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
    auto job_uniforms = [&](int sw, int b0, int b1) __attribute__((always_inline)) {
        Rng g = rng;
        g.sweep = (uint32_t)sw;
        for (int b = b0 + lane; b < b1; b += 128) {
            const int bb = b + 64;
            uint32_t r[4], q[4];
            g.block(SITE_X, 0, (uint32_t)b, r);
            g.block(SITE_X, 0, (uint32_t)bb, q);
            uxs[2 * b] = u53(r[0], r[1]);
            if (2 * b + 1 < cap) uxs[2 * b + 1] = u53(r[2], r[3]);
            if (bb < b1) {
                uxs[2 * bb] = u53(q[0], q[1]);
                if (2 * bb + 1 < cap) uxs[2 * bb + 1] = u53(q[2], q[3]);
            }
        }
    };
    // outputs: parameter roles [0, NP) take ceil(NP/64) passes over wave 1; forecast roles sit on the last wave
    constexpr int OUT_WAVE = 1, FC_WAVE = NW - 1;
    constexpr int NP = 3 * K + KK;
    constexpr int NPASS = (NP + 63) / 64;
    static_assert(NW > 2, "forecast lanes need their own wave here");
    double sum_par[NPASS], sum_fc = 0.0;
    double smp_par[SIG ? NPASS : 1], smp_fc = 0.0;       // signal path: the same sums over the running noise sample (sample_summary)
#pragma unroll
    for (int q = 0; q < NPASS; ++q) sum_par[q] = 0.0;
#pragma unroll
    for (int q = 0; q < (SIG ? NPASS : 1); ++q) smp_par[q] = 0.0;
    (void)smp_par; (void)smp_fc;
    const int fc_e = (wave == FC_WAVE && lane >= 64 - 2 * HMCG_MAXH && lane - (64 - 2 * HMCG_MAXH) < 2 * p.H) ? lane - (64 - 2 * HMCG_MAXH) : -1;
    double fc_yr = 0.0;
    if (fc_e >= 0) {
        fc_yr = p.yreal ? p.yreal[(size_t)w * p.H + (fc_e >> 1)] : __builtin_nan("");
    }
    if (p.resume && p.sumacc) {
        if (wave == OUT_WAVE) {
#pragma unroll
            for (int q = 0; q < NPASS; ++q) if (lane + 64 * q < NP) sum_par[q] = p.sumacc[(size_t)w * NCK + lane + 64 * q];
        }
        if (fc_e >= 0) sum_fc = p.sumacc[(size_t)w * NCK + NP + fc_e];
    }
    if constexpr (SIG) {
        // a launch that resumes inside a noise sample picks that sample's running sums up from its row
        if (p.resume && p.sample_summary) {
            const int smp = p.sweep_begin / p.per_sample, kb = p.sweep_begin - smp * p.per_sample - p.burnin_s;
            if (kb > 0 && kb < p.nrun_s && smp < p.n_samples) {
                const double* row = p.sample_summary + ((size_t)w * p.n_samples + smp) * NS;
                if (wave == OUT_WAVE) {
#pragma unroll
                    for (int q = 0; q < NPASS; ++q) if (lane + 64 * q < NP) smp_par[q] = row[lane + 64 * q];
                }
                if (fc_e >= 0) smp_fc = row[NP + fc_e];
            }
        }
    }
    auto job_outputs = [&](int sw) __attribute__((always_inline)) {
        const int d = SIG ? kept_index(p, sw) : (sw >= p.burnin_s ? sw - p.burnin_s : -1);
        if (d < 0) return;
        const ThetaBufBig<K>& th = sh.th[sw & 1];
        const int smp = SIG ? sw / p.per_sample : 0;
        const bool smp_done = SIG && p.sample_summary && (sw + 1 == (smp + 1) * p.per_sample);     // the sample's last sweep
        double* const ssrow = (SIG && p.sample_summary) ? p.sample_summary + ((size_t)w * p.n_samples + smp) * NS : nullptr;
        (void)smp_done; (void)ssrow;
        const size_t nrun = (size_t)p.nd_ld;
        const int dcol = d - p.draw_off;                 // column of this draw in this launch's output arrays
        if (wave == OUT_WAVE) {
            double mu_u[K];
            int order[K];
#pragma unroll
            for (int i = 0; i < K; ++i) mu_u[i] = th.mu[i];
            sort_order<K>(mu_u, order);
            // signals past the end date: pi_end reports pib[end_pos,:] (:900) = pif[end_pos,:] o b, b = M_{end_pos+1} ... M_{T-1} 1
            // by the backward recursion b <- A (f_t o b) (backwardupdate_P!, :442-457) over the `tail` last steps, whose
            // emission values still sit in the pdf scratch of sweep sw (the next products overwrite it after this job)
            double bsm[SIG ? K : 1];
            if constexpr (SIG) {
                if (tail > 0) {
#pragma unroll
                    for (int r = 0; r < K; ++r) bsm[r] = 1.0;
                    const double* fw = fscr_w;
                    for (int j = tail - 1; j >= 0; --j) {
                        const int t = T - tail + j, own = t / L, ll = t - own * L;
                        double g[K], nb[K];
#pragma unroll
                        for (int c = 0; c < K; ++c) g[c] = fw[fslot(ll, c, own)] * bsm[c];
#pragma unroll
                        for (int r = 0; r < K; ++r) {
                            double acc = 0.0;
#pragma unroll
                            for (int c = 0; c < K; ++c) acc = fma(th.A[r][c], g[c], acc);
                            nb[r] = acc;
                        }
                        rescale_pow2<K>(nb);
#pragma unroll
                        for (int r = 0; r < K; ++r) bsm[r] = nb[r];
                    }
                    double tot = 0.0;
#pragma unroll
                    for (int c = 0; c < K; ++c) { bsm[c] *= sh.pf_rep[sw & 1][c]; tot += bsm[c]; }
                    const double inv = 1.0 / tot;
#pragma unroll
                    for (int c = 0; c < K; ++c) bsm[c] *= inv;                    // pib[end_pos, c], unsorted labels
                }
            }
            (void)bsm;
#pragma unroll
            for (int q = 0; q < NPASS; ++q) {
                const int orole = lane + 64 * q;
                if (orole < NP) {
                    double val;
                    double* dst = nullptr;
                    if (orole < 3 * K) {
                        const int pos = orole % K, which = orole / K;
                        int src = 0;
#pragma unroll
                        for (int qq = 0; qq < K; ++qq) src = (qq == pos) ? order[qq] : src;
                        val = which == 0 ? th.mu[src] : (which == 1 ? th.sig2[src] : th.pi_end[src]);
                        if constexpr (SIG) {
                            if (tail > 0 && which == 2) {
                                double v2 = bsm[0];
#pragma unroll
                                for (int qq = 1; qq < K; ++qq) v2 = (src == qq) ? bsm[qq] : v2;
                                val = v2;
                            }
                        }
                        double* base = which == 0 ? p.mu : (which == 1 ? p.sig2 : p.pi_end);
                        if (base) dst = base + nrun * ((size_t)pos + (size_t)K * w) + dcol;
                    } else {
                        const int e = orole - 3 * K;                 // column-major e = i + K*j
                        int si = 0, sj = 0;
#pragma unroll
                        for (int qq = 0; qq < K; ++qq) { si = (qq == e % K) ? order[qq] : si; sj = (qq == e / K) ? order[qq] : sj; }
                        val = th.A[si][sj];
                        if (p.A) dst = p.A + nrun * ((size_t)e + (size_t)KK * w) + dcol;
                    }
                    if (dst) *dst = val;
                    const double r5 = round5(val);
                    sum_par[q] += r5;
                    if constexpr (SIG) {
                        if (ssrow) {
                            smp_par[q] += r5;
                            if (smp_done) { ssrow[orole] = p.nrun_s > 0 ? smp_par[q] / (double)p.nrun_s : __builtin_nan(""); smp_par[q] = 0.0; }
                        }
                    }
                }
            }
        }
        if (wave == FC_WAVE && p.H > 0) {
            // (pi' A^h) . mu (src/Hmc.jl:658-667) by binary exponentiation, the whole wave cooperating: lane e
            // owns entry e of the matrix being squared, lanes < K the vector; operands live in LDS scratch
            // (the wave's own LDS operations are ordered, so a compiler fence is all that separates the steps)
            for (int hi = 0; hi < p.H; ++hi) {
                if constexpr (SIG) {
                    if ((p.blend_mask >> hi) & 1) {
                        // forecastsignal (src/Hmc.jl:670-681, :908-909): the horizon equals the number of signal steps past the end
                        // date; blend of the last noisy observation and each state mean, weighted by the LAST step's probabilities
                        if (lane == 0) {
                            const double tau = 1.0 / (p.sigma_signal ? p.sigma_signal[w] : 0.0);
                            const double a = tau / (1.0 + tau);
                            const double ysig = sh.y_last[(sw / p.per_sample) & 1];
                            double fv0 = 0.0;
#pragma unroll
                            for (int i = 0; i < K; ++i) fv0 += th.pi_end[i] * (a * ysig + (1.0 - a) * th.mu[i]);
                            sh.fcval[hi] = fv0;
                        }
                        __builtin_amdgcn_wave_barrier();
                        continue;
                    }
                }
                double* M = sh.fcM[0];
                double* M2 = sh.fcM[1];
                double* vv = sh.fcv[0];
                double* vv2 = sh.fcv[1];
                if (lane < KK) M[lane] = th.A[lane / K][lane % K];
                if (lane < K) vv[lane] = th.mu[lane];
                __builtin_amdgcn_wave_barrier();
                for (unsigned hh = (unsigned)p.horizons[hi]; hh != 0; hh >>= 1) {
                    if (hh & 1u) {
                        if (lane < K) {
                            double acc = 0.0;
#pragma unroll
                            for (int j = 0; j < K; ++j) acc = fma(M[lane * K + j], vv[j], acc);
                            vv2[lane] = acc;
                        }
                        __builtin_amdgcn_wave_barrier();
                        double* t = vv; vv = vv2; vv2 = t;
                    }
                    if (hh > 1u) {
                        if (lane < KK) {
                            const int i = lane / K, j = lane % K;
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < K; ++k) acc = fma(M[i * K + k], M[k * K + j], acc);
                            M2[lane] = acc;
                        }
                        __builtin_amdgcn_wave_barrier();
                        double* t = M; M = M2; M2 = t;
                    }
                }
                if (lane == 0) {
                    double fv0 = 0.0;
#pragma unroll
                    for (int i = 0; i < K; ++i) fv0 = fma(th.pi_end[i], vv[i], fv0);
                    sh.fcval[hi] = fv0;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (fc_e >= 0) {
            const double fv = sh.fcval[fc_e >> 1];
            const double val = (fc_e & 1) ? fv - fc_yr : fv;
            if (p.fcast) p.fcast[nrun * ((size_t)fc_e + (size_t)(2 * p.H) * w) + dcol] = val;
            const double r5 = round5(val);
            sum_fc += r5;
            if constexpr (SIG) {
                if (ssrow) {
                    smp_fc += r5;
                    if (smp_done) { ssrow[NP + fc_e] = p.nrun_s > 0 ? smp_fc / (double)p.nrun_s : __builtin_nan(""); smp_fc = 0.0; }
                }
            }
        }
    };

    // ---- sufficient statistics of the chain state in xs[] -------------------------------------
    // The statistics are accumulated per thread over its L steps, LAST step first -- the order of the backward pass's apply
    // loop, which takes them along at the end of every sweep (below); the stand-alone form (prologue, resumed launches, a new
    // noise sample) walks the same way, so that a resumed chain adds the same numbers in the same order as an uninterrupted one.
    struct StatAcc { double d1[K], d2[K], s1[SIG ? K : 1], s2[SIG ? K : 1]; };
    auto stats_begin = [&](StatAcc& a) __attribute__((always_inline)) {
        for (int e = lane; e < KK; e += 64) sh.cnt[wave][e] = 0;
        if constexpr (SIG) { if (lane < K) sh.cntm[wave][lane] = 0; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < K; ++i) { a.d1[i] = 0.0; a.d2[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < (SIG ? K : 1); ++i) { a.s1[i] = 0.0; a.s2[i] = 0.0; }
    };
    // step t (< T) in state xv, followed by state xn (counted when t + 1 < T), observation yv, pivot pv of state xv
    auto stats_step = [&](StatAcc& a, int t, int xv, int xn, double yv, double pv) __attribute__((always_inline)) {
        const double dl = yv - pv;
        const bool issig = SIG && t >= sb && t < se;
        const int xo = issig ? -1 : xv;                    // observation set (src/Hmc.jl:254-258)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const double dm = (xo == i) ? dl : 0.0;
            a.d1[i] += dm;
            a.d2[i] = fma(dm, dm, a.d2[i]);
        }
        if constexpr (SIG) {
            const int xg = issig ? xv : -1;                // signal set (:268-272)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double dm = (xg == i) ? dl : 0.0;
                a.s1[i] += dm;
                a.s2[i] = fma(dm, dm, a.s2[i]);
            }
            if (issig) atomicAdd(&sh.cntm[wave][xv], 1u);
        }
        if (t + 1 < T) atomicAdd(&sh.cnt[wave][xv * K + xn], 1u);
    };
    auto stats_finish = [&](StatAcc& a) __attribute__((always_inline)) {
        double (&d1)[K] = a.d1; double (&d2)[K] = a.d2;
        double (&s1)[SIG ? K : 1] = a.s1; double (&s2)[SIG ? K : 1] = a.s2;
        (void)s1; (void)s2;
        double o0, o1, v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = i < K ? d1[i] : 0.0;
        wave_sum8_transposed(v8, lane, o0, o1);
        if ((lane & 0x3C) == 12) {
            const int i0 = 4 * (lane & 1) + (lane & 2);
            if (i0 < K) sh.red_d1[wave][i0] = o0;
            if (i0 + 1 < K) sh.red_d1[wave][i0 + 1] = o1;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = i < K ? d2[i] : 0.0;
        wave_sum8_transposed(v8, lane, o0, o1);
        if ((lane & 0x3C) == 12) {
            const int i0 = 4 * (lane & 1) + (lane & 2);
            if (i0 < K) sh.red_d2[wave][i0] = o0;
            if (i0 + 1 < K) sh.red_d2[wave][i0 + 1] = o1;
        }
        if constexpr (SIG) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v8[i] = i < K ? s1[i < K ? i : 0] : 0.0;
            wave_sum8_transposed(v8, lane, o0, o1);
            if ((lane & 0x3C) == 12) {
                const int i0 = 4 * (lane & 1) + (lane & 2);
                if (i0 < K) sh.red_s1[wave][i0] = o0;
                if (i0 + 1 < K) sh.red_s1[wave][i0 + 1] = o1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v8[i] = i < K ? s2[i < K ? i : 0] : 0.0;
            wave_sum8_transposed(v8, lane, o0, o1);
            if ((lane & 0x3C) == 12) {
                const int i0 = 4 * (lane & 1) + (lane & 2);
                if (i0 < K) sh.red_s2[wave][i0] = o0;
                if (i0 + 1 < K) sh.red_s2[wave][i0 + 1] = o1;
            }
        }
        if (tid == 0) sh.x_end = x_end;
    };
    auto publish_stats = [&]() __attribute__((always_inline)) {            // stand-alone: from xs[], ylds[], sh.pivot[]
        StatAcc a;
        stats_begin(a);
        for (int l = L - 1; l >= 0; --l) {
            const int t = t0 + l;
            if (t < T) {
                const int xv = min((int)xs[t], K - 1), xn = min((int)xs[t + 1], K - 1);          // xs has 8 spare entries behind cap
                stats_step(a, t, xv, xn, ylds[t], sh.pivot[xv]);
            }
        }
        stats_finish(a);
    };
    // a new noise sample (src/Hmc.jl:892): Yfake = Yreal + N(0,1) * sigma_signal on the signal range (every thread its own
    // steps: the statistics that follow read the same ones); the chain state carries over (:889-895)
    auto regen_y = [&](int smp, bool report) __attribute__((always_inline)) {
        if constexpr (SIG) {
            const double ssig = p.sigma_signal ? p.sigma_signal[w] : 0.0;
            const bool noisy = ssig != 0.0 || p.n_samples > 1;
            int svb = 0, sve = 0;
            if (p.save_range) { svb = p.save_range[2 * w]; sve = p.save_range[2 * w + 1]; }
            Rng gn = rng;
            gn.sweep = (uint32_t)smp;
            for (int l = 0; l < L; ++l) {
                const int t = t0 + l;
                if (noisy && t >= sb && t < se) {
                    uint32_t r[4];
                    gn.block(SITE_NOISE, 0, (uint32_t)t, r);
                    ylds[t] = p.Y[(size_t)w * p.ldY + t] + box_muller(r) * ssig;
                }
                if (report && p.sigvals && t >= svb && t < sve && t < T && (t - svb) < p.nsave_ld)
                    p.sigvals[((size_t)w * p.n_samples + smp) * p.nsave_ld + (t - svb)] = ylds[t];
                if (t == T - 1) sh.y_last[smp & 1] = ylds[t];       // forecastsignal's `signal` (:909)
            }
        }
    };
    if constexpr (SIG) {
        if (p.sweep_begin % p.per_sample != 0) regen_y(p.sweep_begin / p.per_sample, false);   // resumed inside a sample
    }
    static_assert(K <= 8, "wave_sum8_transposed carries 8 slots");
    publish_stats();
    if (p.sweep_begin < p.sweep_end && shadow_wave == 0) job_prep(p.sweep_begin);

    // pdfs of one observation, scaled by the power of two that brings the largest into [0.5,1)
    auto pdfs = [&](const ThetaBufBig<K>& th, double yv, bool valid, bool issig, double (&fv)[K]) __attribute__((always_inline)) {
        unsigned hm = 0;
        const double kf = (SIG && issig) ? kfac : 1.0;            // signal positions: sd scaled by (1 + kappa) (:382, quirk 4)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const double z = (yv - th.mu[s]) * (th.isd[s] * kf);
            fv[s] = exp_tab<true>(-(z * z), sh.exptab, lane & (EXPTAB_C - 1)) * (th.coef[s] * kf);
            hm = max(hm, (unsigned)__double2hiint(fv[s]));
        }
        {
            const int e = 1022 - (int)(hm >> 20);
#pragma unroll
            for (int s = 0; s < K; ++s) fv[s] = ldexp(fv[s], e);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hm < 0x01A56E1Fu) != 0ull, 0)) {      // rare, wave-uniform branch
            if (hm < 0x01A56E1Fu) {
                if (valid) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                for (int s = 0; s < K; ++s) fv[s] = 1.0;
            }
        }
    };

#ifdef HMCG_STAMPS
    unsigned long long stamp_acc[HMCG_NSTAMP];
    for (int i = 0; i < HMCG_NSTAMP; ++i) stamp_acc[i] = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int sweep = p.sweep_begin; sweep < p.sweep_end; ++sweep) {
        rng.sweep = (uint32_t)sweep;
        const int par = sweep & 1;
        ThetaBufBig<K>& th = sh.th[par];
        // the transposed transition matrix through an opaque LDS pointer: all 64 reads of a step are then immediate
        // offsets from ONE address register (from the struct's own base every read beyond 2 KB costs an s_add + v_mov)
        typedef __attribute__((address_space(3))) const double lds_cdouble;
        lds_cdouble* Atp = (lds_cdouble*)&th.At[0][0];
        asm volatile("" : "+v"(Atp));
        const bool last_sweep = sweep + 1 == p.sweep_end;
        if constexpr (SIG) {
            const int smp = sweep / p.per_sample;
            if (sweep == smp * p.per_sample) { regen_y(smp, true); publish_stats(); }      // the statistics are retaken on the new data
        }
        __syncthreads();                                                     // Ba
        STAMP(0);
        if (wave == 0) {
            // ---- parameter draws; roles [0,K) sig2/mu, [K, K+K^2) A entries; ceil(NG/64) passes ----
            const RngBuf<K>& rb = sh.rb[par];
            for (int base = 0; base < NG; base += 64) {
                const int role = base + lane;
                const bool is_g = role < NG, is_sig = role < K;
                int c = 0;
                if (is_g && !is_sig) {
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) c += (int)sh.cnt[ww][role - K];
                } else if (is_sig) {
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) c += (int)sh.cnt[ww][role * K + j];
                    c += (sh.x_end == role) ? 1 : 0;
                }
                double shape = 1.0, bpar = 1.0, Neff = 0.0, Ssum = 0.0;
                if (is_sig) {
                    double d1 = 0.0, d2 = 0.0;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) { d1 += sh.red_d1[ww][role]; d2 += sh.red_d2[ww][role]; }
                    const double piv = sh.pivot[role];
                    const double beta = (sweep == 0) ? 1.0 : 2.0;
                    if constexpr (!SIG) {
                        Neff = (double)c;
                        const double rn = c > 0 ? rcp_fast(Neff) : 0.0;
                        const double ybar = c > 0 ? piv + d1 * rn : 0.0;
                        const double S2 = c > 0 ? fmax(d2 - d1 * d1 * rn, 0.0) : 0.0;
                        Ssum = piv * Neff + d1;
                        const double dm = ybar - xi;
                        shape = p.alpha + 0.5 * Neff;
                        bpar = beta + 0.5 * S2 + 0.5 * Neff * p.nu / (Neff + p.nu) * (dm * dm);
                    } else {
                        // observation set and signal set (src/Hmc.jl:254-314), as in the SIG variants of gibbs_device.hpp
                        int Mi = 0;
                        double g1 = 0.0, g2 = 0.0;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) { Mi += (int)sh.cntm[ww][role]; g1 += sh.red_s1[ww][role]; g2 += sh.red_s2[ww][role]; }
                        const int Ni = c - Mi;
                        const double dNi = (double)Ni, dMi = (double)Mi;
                        const double S = piv * dNi + d1, Sm = piv * dMi + g1;                       // sums of y
                        const double S2 = Ni > 0 ? fmax(d2 - d1 * d1 / dNi, 0.0) : 0.0;            // :291-294
                        const double Sm2 = Mi > 0 ? fmax(g2 - g1 * g1 / dMi, 0.0) : 0.0;           // :296-300
                        const double totalbar = (Ni + Mi) > 0 ? (S + Sm) / (dNi + dMi) : 0.0;      // :282-288
                        Neff = dNi + dMi * kfac;                                                    // :302-303
                        Ssum = S + Sm;                                                              // :331 (Sm unscaled: quirk 4)
                        const double dm = totalbar - xi;
                        shape = p.alpha + 0.5 * dNi + 0.5 * dMi;                                    // :313
                        bpar = beta + 0.5 * S2 + (0.5 * kfac) * Sm2 + 0.5 * Neff * p.nu * rcp_fast(Neff + p.nu) * (dm * dm);   // :314
                    }
                } else if (is_g) {
                    shape = (double)(c + 1);
                }
                double val = 1.0;
                if (is_g) {
                    const uint32_t site = role < K ? SITE_SIG2 : SITE_A;
                    const uint32_t elem = role < K ? (uint32_t)role : (uint32_t)(role - K);
                    if (shape == 1.0) {
                        uint32_t r[4];
                        rng.block(site, elem, 0, r);
                        val = -log_fast(1.0 - u53(r[0], r[1]));
                    } else {
                        const double a = shape < 1.0 ? shape + 1.0 : shape;
                        const double dd = a - 1.0 / 3.0;
                        const double cc = rcp_fast(3.0 * sqrt_fast(dd));
                        val = mt_try(dd, cc, rb.x[role][0], rb.lu[role][0], rb.uu[role][0]);
                        if (val < 0.0) val = mt_try(dd, cc, rb.x[role][1], rb.lu[role][1], rb.uu[role][1]);
                        if (val < 0.0) {
                            int j = 2;
                            for (; j < GAMMA_MAX_ATTEMPTS && val < 0.0; ++j) {
                                uint32_t r[4];
                                rng.block(site, elem, 2u * (uint32_t)j, r);
                                const double xx = box_muller(r);
                                rng.block(site, elem, 2u * (uint32_t)j + 1u, r);
                                { const double u3 = u53(r[0], r[1]); val = mt_try(dd, cc, xx, log_fast(1.0 - u3), u3); }
                            }
                            if (val < 0.0) { val = dd; st |= HMCG_ST_GAMMA_CAP; }
                        }
                        if (shape < 1.0) {
                            uint32_t r[4];
                            rng.block(site, elem, 0xFFFFFFFFu, r);
                            val *= pow(1.0 - u53(r[0], r[1]), 1.0 / shape);
                        }
                    }
                    sh.gval[role] = val;
                }
                if (is_sig) {
                    const double sig2 = bpar * rcp_fast(val);
                    const double rnn = rcp_fast(Neff + p.nu);
                    const double m = (Ssum + p.nu * xi) * rnn;
                    const double sd = sqrt_fast(sig2);
                    const double sdev = sd * sqrt_fast(rnn);
                    const double isd = rcp_fast(sd);
                    th.mu[role] = m + sdev * rb.z[role];
                    th.sig2[role] = sig2;
                    th.isd[role] = isd * 0.70710678118654752440;                 // 1/(sd sqrt 2): exp(-z^2) = exp(-((y-mu)/sd)^2 / 2)
                    th.coef[role] = INVSQRT2PI * isd;
                    th.rho[role] = rb.rho[role];
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < KK; e += 64) {                // Dirichlet rows of A (:367): normalise in column order
                const int i = e / K;
                double gs = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) gs += sh.gval[K + i * K + j];
                const double a = sh.gval[K + e] * rcp_fast(gs);
                th.A[i][e % K] = a;
                th.At[e % K][i] = a;
            }
        } else {
            if (sweep > p.sweep_begin) job_outputs(sweep - 1);
            if (shadow_wave == NSH - 1 && sweep + 1 < p.sweep_end) job_prep(sweep + 1);
        }
        {
            // This sweep's (T+1)/2 Philox blocks of state-draw uniforms, dealt in proportion to what else a wave carries in
            // this phase (stamps at K=8, T=5000: parameter draw 4.8 k ticks, parameter outputs 3.4 k, forecast + next
            // sweep's RNG preparation 8.5 k, a third of the uniforms 7.7 k): 4/16 to the parameter wave, 4/16 to the output
            // wave, 7/16 to the wave with no fixed job, the rest to the forecast wave; further waves share evenly.
            const int nblk = (T + 1) >> 1;
            int b0, b1;
            if constexpr (NW == 4) {
                const int c0 = (nblk * 4) / 16, c1 = (nblk * 8) / 16, c2 = (nblk * 15) / 16;
                const int wu = __builtin_amdgcn_readfirstlane(wave);
                b0 = wu == 0 ? 0 : (wu == 1 ? c0 : (wu == 2 ? c1 : c2));
                b1 = wu == 0 ? c0 : (wu == 1 ? c1 : (wu == 2 ? c2 : nblk));
            } else {
                const int per = (nblk + NW - 1) / NW;
                b0 = wave * per; b1 = min((wave + 1) * per, nblk);
            }
            job_uniforms(sweep, b0, b1);
        }
        STAMP(1);
        __syncthreads();                                                     // Bb
        STAMP(2);
        // ---- forward filter: local product of this thread's L matrices A diag(f_t) ----
        double Q[KK];
        double N[KK];
        const bool kept_sweep = SIG ? kept_index(p, sweep) >= 0 : sweep >= p.burnin_s;     // (without the signal path a launch is one sample)
        const bool do_smooth = SM && kept_sweep && (p.pi_smooth_mean != nullptr || p.pi_filter_mean != nullptr || p.pi_smooth_draws != nullptr);
        const int dk = SIG ? kept_index(p, sweep) : sweep - p.burnin_s;      // kept-draw index (meaningful when kept_sweep)
        (void)dk;
#ifdef HMCG_BIG_NSC
        constexpr int NSC = K == 8 ? HMCG_BIG_NSC : 0;     // the first NSC columns of A live in scalar registers for the whole phase
#else
        constexpr int NSC = 0;
#endif
        double a_sc[NSC > 0 ? NSC : 1][K];
#pragma unroll
        for (int s = 0; s < NSC; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) a_sc[s][k] = uniform_f64(Atp[s * K + k]);
#ifndef HMCG_BIG_NO_ASM                            // (-DHMCG_BIG_NO_ASM: the C++ loop everywhere, for A/B timing)
        constexpr bool ASM_PRODUCT = K == 8 && !SM;
#else
        constexpr bool ASM_PRODUCT = false;
#endif
        if constexpr (ASM_PRODUCT) {
            // K = 8: the pdf pass by itself (every step's K values into the scratch -- the replay reads them there anyway),
            // then the chunk product as one hand-scheduled assembly loop that reads them back (tools/gen_product_asm.py:
            // same operations in the same order as the C++ loop below, bit-identical results; 580 VALU instructions per
            // step instead of ~930, 32 independent chains in flight instead of ~4)
            // The pdf pass, two steps at a time and STAGE BY STAGE (the same operations per value as pdfs() / exp_tab, so the
            // same bits): every stage is 2 K independent instructions, and the 2 K table reads of a pair of steps are in
            // flight together -- left to itself the scheduler runs the K exponentials of a step nearly one after the other,
            // each waiting for its own table read (7.3 ticks per instruction; staged: measured below).
            {
                double mu_s[K], isd_s[K], coef_s[K];         // the emission parameters as scalar operands
#pragma unroll
                for (int s = 0; s < K; ++s) { mu_s[s] = uniform_f64(th.mu[s]); isd_s[s] = uniform_f64(th.isd[s]); coef_s[s] = uniform_f64(th.coef[s]); }
                constexpr int G = 2;
                // one sched_barrier per operation: the 2 K instances of an operation are issued back to back (independent of
                // each other), the dependent operation follows as the next group -- written as plain loops the scheduler
                // keeps each value's chain together and interleaves only two of them (measured: 6.8 ticks per instruction)
#define HMCG_EACH for (int g = 0; g < G; ++g) _Pragma("unroll") for (int s = 0; s < K; ++s)
#ifdef HMCG_PDF_NOSB
#define HMCG_SB
#else
#define HMCG_SB __builtin_amdgcn_sched_barrier(0)
#endif
                for (int l0 = 0; l0 < L; l0 += G) {
                    double x[G][K], nn[G][K], tj[G][K], fv[G][K], r[G][K], pp[G][K], yv[G];
                    int ni[G][K];
                    unsigned hm[G];
                    double kf[G];
                    bool live[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int l = l0 + g < L ? l0 + g : L - 1;       // (an odd L: the last pair computes step L-1 twice)
                        live[g] = l0 + g < L;
                        yv[g] = ylds[t0 + l];
                        kf[g] = (SIG && t0 + l >= sb && t0 + l < se) ? kfac : 1.0;
                        hm[g] = 0;
                    }
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH x[g][s] = yv[g] - mu_s[s];
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH x[g][s] = x[g][s] * (isd_s[s] * kf[g]);
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH x[g][s] = -(x[g][s] * x[g][s]);
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH x[g][s] = fmax(x[g][s], -746.0);
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH nn[g][s] = fma(x[g][s], EXPTAB_N == 64 ? 92.332482616893657 : 369.3299304675746, EXP_MAGIC);    // exp_tab<true>
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH { ni[g][s] = __double2loint(nn[g][s]); nn[g][s] = nn[g][s] - EXP_MAGIC; }
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
                    HMCG_EACH fv[g][s] = tj[g][s] * pp[g][s];
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH fv[g][s] = ldexp(fv[g][s], ni[g][s] >> (EXPTAB_N == 64 ? 6 : 8));
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH fv[g][s] = fv[g][s] * (coef_s[s] * kf[g]);
                    HMCG_SB;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
#pragma unroll
                        for (int s = 0; s < K; ++s) hm[g] = max(hm[g], (unsigned)__double2hiint(fv[g][s]));
                    }
                    HMCG_SB;
#pragma unroll
                    HMCG_EACH fv[g][s] = ldexp(fv[g][s], 1022 - (int)(hm[g] >> 20));
                    HMCG_SB;
#undef HMCG_EACH
#undef HMCG_SB
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hm[0] < 0x01A56E1Fu || hm[1] < 0x01A56E1Fu) != 0ull, 0)) {   // rare, wave-uniform
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            if (hm[g] < 0x01A56E1Fu) {
                                if (live[g] && t0 + l0 + g < T) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                                for (int s = 0; s < K; ++s) fv[g][s] = 1.0;
                            }
                        }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (live[g]) {
#pragma unroll
                            for (int s = 0; s < K; ++s) fscr[fslot(l0 + g, s, threadIdx.x)] = fv[g][s];
                        }
                    }
                }
            }
            STAMP(3);
            const double* asm_fb = fscr_w;
            const uint32_t asm_voff = (uint32_t)threadIdx.x * 16u;
            const uint32_t asm_lds = (uint32_t)(uintptr_t)(lds_cdouble*)&th.A[0][0];
            int asm_l, asm_t, asm_u;
#if defined(HMCG_BIG_ASM_TESTINC)                 // (timing experiments of tools/gen_product_asm.py --test-*: wrong results)
#include HMCG_BIG_ASM_TESTINC
#else
#include "product_asm_k8.inc"
#endif
            (void)asm_l; (void)asm_t; (void)asm_u;
            rescale_pow2<KK>(Q);
        } else {
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) Q[r * K + s] = (r == s) ? 1.0 : 0.0;
            // Q <- Q * (A diag(f_t)), in place, four rows at a time: row r of the product needs row r of Q only, so once a
            // block of rows has all its columns it replaces the block it came from -- one matrix and half a matrix live
            // (192 registers) instead of two (256, i.e. ~190 VGPR<->AGPR copies per step).  One column of A at a time from
            // LDS (wave-uniform address: a broadcast read), the block's rows against it; every column is read once per block
            // (twice per step for K = 8: 64 broadcast reads instead of 32, far from loading the LDS pipe -- a two-rows form
            // that read A four times per step had been LDS-bound).
#ifdef HMCG_BIG_RB
            constexpr int RB = HMCG_BIG_RB;
#else
            constexpr int RB = 4;
#endif
            for (int l = 0; l < L; ++l) {
                asm volatile("" ::: "memory");       // keep the A columns as per-step LDS reads (hoisting all 64 would spill)
                if constexpr (SM) {
                    if (t0 + l >= T) {               // padded steps stay out of the products (identity): the suffix scan
#pragma unroll                                       //  of the smoothing pass must not see them
                        for (int s = 0; s < K; ++s) fscr[fslot(l, s, threadIdx.x)] = 1.0;
                        continue;
                    }
                }
                double fv[K];
                pdfs(th, ylds[t0 + l], t0 + l < T, SIG && t0 + l >= sb && t0 + l < se, fv);
                // the replay needs the same K values again: they travel through a lane-contiguous HBM scratch (K coalesced
                // 512-byte stores per wave and step) instead of being recomputed (K exponentials per step)
#pragma unroll
                for (int s = 0; s < K; ++s) fscr[fslot(l, s, threadIdx.x)] = fv[s];
#pragma unroll
                for (int r0 = 0; r0 < K; r0 += RB) {
                    double tb[RB][K];
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        double a[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) a[k] = s < NSC ? a_sc[s < NSC ? s : 0][k] : Atp[s * K + k];
#pragma unroll
                        for (int rr = 0; rr < RB; ++rr) {
                            if (r0 + rr < K) {
                                double acc = Q[(r0 + rr) * K] * a[0];
#pragma unroll
                                for (int k = 1; k < K; ++k) acc = fma(Q[(r0 + rr) * K + k], a[k], acc);
                                tb[rr][s] = acc * fv[s];
                            }
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr)
#pragma unroll
                        for (int s = 0; s < K; ++s) if (r0 + rr < K) Q[(r0 + rr) * K + s] = tb[rr][s];
                }
                // (every step's largest pdf lies in [0.5,1): eight steps between two exact power-of-two rescalings are far
                //  inside the fp64 range)
                if ((l & 7) == 7) rescale_pow2<KK>(Q);
            }
            rescale_pow2<KK>(Q);
        }
        double Qloc[SM ? KK : 1];                                 // this thread's own chunk product, for the suffix scan
        if constexpr (SM) {
#pragma unroll
            for (int i = 0; i < KK; ++i) Qloc[i] = Q[i];
        }
        (void)Qloc;
        STAMP(4);
        scan_level_rowwise<K, DPP_ROW_SHR1, 0xF>(Q, N);
        scan_level_rowwise<K, DPP_ROW_SHR2, 0xF>(N, Q);
        rescale_pow2<KK>(Q);
        scan_level_rowwise<K, DPP_ROW_SHR4, 0xF>(Q, N);
        scan_level_rowwise<K, DPP_ROW_SHR8, 0xF>(N, Q);
        rescale_pow2<KK>(Q);
#ifdef HMCG_BIG_FULL_SCAN                         // (A/B: the two upper scan levels as full K x K products in every lane, as until round 4)
        scan_level_rowwise<K, DPP_ROW_BCAST15, 0xA>(Q, N);
        scan_level_rowwise<K, DPP_ROW_BCAST31, 0xC>(N, Q);
        rescale_pow2<KK>(Q);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < KK; ++i) sh.wtot[wave][i] = Q[i];
        }
#else
        // The scan stops at the 16-lane rows (four DPP levels).  What the two upper levels produced -- 2 x K^3 multiply-adds in
        // EVERY lane -- is needed in two much smaller forms only: the wave's total W = R0 R1 R2 R3 (for the later waves), and
        // for the lanes of row j the vector u_j = v R0 ... R_{j-1} that enters the row (v: what enters the wave).  Both are
        // products of the four row totals R_j, which the last lane of each row leaves in LDS: W by three COOPERATIVE K x K
        // products (lane 8 r + c owns entry (r, c): K multiply-adds a product instead of K^3), the u_j after barrier Bc by three
        // cooperative vector-matrix steps (lane c of every eight owns entry c), exactly as the cross-wave prefix is formed.
        const int er = lane >> 3, ec = lane & 7;                 // the entry this lane owns in a cooperative product
        const bool evalid = er < K && ec < K;
        if ((lane & 15) == 15) {
#pragma unroll
            for (int i = 0; i < KK; ++i) sh.rtot[wave][lane >> 4][i] = Q[i];
        }
        __builtin_amdgcn_wave_barrier();                         // (the row totals are read by the other lanes of this wave only)
        {
            auto coop = [&](const double* X, const double* Y) __attribute__((always_inline)) {
                const int rr = evalid ? er : 0, cc2 = evalid ? ec : 0;
                double acc = X[rr * K] * Y[cc2];
#pragma unroll
                for (int k = 1; k < K; ++k) acc = fma(X[rr * K + k], Y[k * K + cc2], acc);
                return acc;
            };
            asm volatile("" ::: "memory");
            double pe = coop(sh.rtot[wave][0], sh.rtot[wave][1]);
            if (evalid) sh.ptmp[wave][0][er * K + ec] = pe;
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            pe = coop(sh.ptmp[wave][0], sh.rtot[wave][2]);
            if (evalid) sh.ptmp[wave][1][er * K + ec] = pe;
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            pe = coop(sh.ptmp[wave][1], sh.rtot[wave][3]);
            if (evalid) sh.wtot[wave][er * K + ec] = pe;
        }
#endif
        STAMP(5);
        __syncthreads();                                                     // Bc
        STAMP(6);
        // prefix vector rho' * (earlier waves) * (exclusive lane prefix)
        double av[K];
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        {
            // rho' W_0 ... W_{wave-1} is the same in every lane, and the last wave's three dependent products stand between every
            // other wave and barrier Bd.  As 64 redundant K x K vector-matrix products each was 64 FMAs behind 32 LDS reads
            // (~1 k ticks a step); cooperatively -- lane c of every 8-lane group carries entry c, reads column c of the total
            // (K reads), gets the K entries of the vector by lane shuffles -- a step is K FMAs.  Same sums in the same order.
            const int c8 = lane & 7, g8 = lane & ~7;
            const int cc = c8 < K ? c8 : 0;
#ifdef HMCG_BIG_SHFL_CHAIN                            // (A/B: the vector's entries fetched by eight lane shuffles a step, as until round 4)
            double vc = th.rho[cc];
            for (int ww = 0; ww < wave_u; ++ww) {
                double col[K];
#pragma unroll
                for (int r = 0; r < K; ++r) col[r] = sh.wtot[ww][r * K + cc];
                double acc = __shfl(vc, g8, 64) * col[0];
#pragma unroll
                for (int r = 1; r < K; ++r) acc = fma(__shfl(vc, g8 | r, 64), col[r], acc);
                vc = acc;
            }
            auto row_step = [&](int j) __attribute__((always_inline)) {
                double col[K];
#pragma unroll
                for (int r = 0; r < K; ++r) col[r] = sh.rtot[wave_u][j][r * K + cc];
                double acc = __shfl(vc, g8, 64) * col[0];
#pragma unroll
                for (int r = 1; r < K; ++r) acc = fma(__shfl(vc, g8 | r, 64), col[r], acc);
                vc = acc;
            };
#else
            // A step v <- v M with lane c of every eight owning entry c: the entries of v reach the lane by DPP ROTATIONS of the
            // 16-lane row (both halves of a row hold the same eight entries, so row_ror:k hands lane c entry (c - k) mod 8) -- 14
            // vector moves and 8 multiply-adds per step, nothing through LDS on the dependent chain (the 8 lane shuffles a step
            // of the first version were 16 LDS round trips, ~1.2 k ticks a step on the last wave, which everyone waits for); the
            // matrix entries M[(c - k) mod 8][c] each lane needs do not depend on v and are read ahead.
            double vc = c8 < K ? th.rho[cc] : 0.0;
            auto coop_step = [&](const double* Mm) __attribute__((always_inline)) {
                double col[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = (c8 - k) & 7;
                    col[k] = (r < K && c8 < K) ? Mm[(r < K ? r : 0) * K + cc] : 0.0;
                }
                double acc = vc * col[0];
                acc = fma(dpp_f64<0x121, 0xF>(0.0, vc), col[1], acc);
                acc = fma(dpp_f64<0x122, 0xF>(0.0, vc), col[2], acc);
                acc = fma(dpp_f64<0x123, 0xF>(0.0, vc), col[3], acc);
                acc = fma(dpp_f64<0x124, 0xF>(0.0, vc), col[4], acc);
                acc = fma(dpp_f64<0x125, 0xF>(0.0, vc), col[5], acc);
                acc = fma(dpp_f64<0x126, 0xF>(0.0, vc), col[6], acc);
                acc = fma(dpp_f64<0x127, 0xF>(0.0, vc), col[7], acc);
                vc = acc;
            };
            for (int ww = 0; ww < wave_u; ++ww) coop_step(sh.wtot[ww]);
            auto row_step = [&](int j) __attribute__((always_inline)) { coop_step(sh.rtot[wave_u][j]); };
#endif
#ifndef HMCG_BIG_FULL_SCAN
            // u_j for the lanes of row j: three more steps of the same kind over this wave's row totals; a lane keeps the
            // vector of its own row
            {
                const int row = lane >> 4;
                double vrow = vc;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    row_step(j);
                    vrow = (row > j) ? vc : vrow;
                }
                vc = vrow;
            }
#endif
#pragma unroll
            for (int s = 0; s < K; ++s) av[s] = __shfl(vc, g8 | s, 64);
        }
        {
            double nv[K];
#pragma unroll
            for (int s = 0; s < K; ++s) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < K; ++r)
#ifdef HMCG_BIG_FULL_SCAN
                    acc = fma(av[r], dpp_f64<DPP_WAVE_SHR1, 0xF>((r == s) ? 1.0 : 0.0, Q[r * K + s]), acc);
#else
                    acc = fma(av[r], dpp_f64<DPP_ROW_SHR1, 0xF>((r == s) ? 1.0 : 0.0, Q[r * K + s]), acc);   // the row's exclusive prefix
#endif
                nv[s] = acc;
            }
            rescale_pow2<K>(nv);
#pragma unroll
            for (int s = 0; s < K; ++s) av[s] = nv[s];
        }
        // ---- replay of the normalised recursion (:413-432), fused with the state maps of update_X (:459-484):
        // at step t the running sums over r of pif[t-1,r] A[r,s] are both pif[t,s]/f and the cumulative
        // weights of the draw X[t-1] | X[t] = s; the eps() guard is pif[t,s] itself.
        // (The pdfs come back from the scratch the product phase filled; the transition matrix stays in LDS, a column at a
        //  time -- 64 more live doubles were measured to land in AGPRs and cost four times the instructions of the reads.)
        {
            const bool want_pif = (last_sweep || do_smooth) && p.pif_final != nullptr;
#ifndef HMCG_BIG_NO_ASM
            constexpr bool ASM_REPLAY = K == 8 && !SM && !SIG && !STREAM;
#else
            constexpr bool ASM_REPLAY = false;
#endif
            bool replay_done = false;
#ifdef HMCG_REPLAY_CHECK
            double av_asm[K] = {}, chk_pie = 0.0;
            uint32_t chk_maps = 0;
            int st_asm = 0;
#endif
            if constexpr (ASM_REPLAY) {
                if (!want_pif) {
                    // K = 8, every sweep that does not also hand out pif per step: the whole loop as one hand-scheduled assembly
                    // statement (tools/gen_replay_asm.py: the same operations in the same order, bit-identical; ~290
                    // instructions per step instead of 384 -- no selects in the categorical count, A in registers)
                    typedef __attribute__((address_space(3))) const void lds_cvoid;
                    const int own_t = (T - 1) / L;
                    const double* asm_fb = fscr_w;
                    const uint32_t asm_voff = (uint32_t)threadIdx.x * 16u;
                    const uint32_t asm_lds = (uint32_t)(uintptr_t)(lds_cdouble*)&th.A[0][0];
                    const uint32_t asm_au = (uint32_t)(uintptr_t)(lds_cvoid*)&uxs[t0 >= 1 ? t0 - 1 : 0];
                    const uint32_t asm_am = (uint32_t)(uintptr_t)(lds_cvoid*)&maps[t0 >= 1 ? t0 - 1 : 0];
                    const uint32_t asm_incu = t0 >= 1 ? 8u : 0u, asm_incm = t0 >= 1 ? 4u : 0u;
                    const int asm_ownw = __builtin_amdgcn_readfirstlane((own_t >> 6) == wave_u ? 1 : 0);
                    const int asm_lown = tid == own_t ? (T - 1) - own_t * L : -1;
                    const uint32_t asm_pie = (uint32_t)(uintptr_t)(lds_cvoid*)&th.pi_end[0];
                    const int asm_flag = HMCG_ST_EMIS_UNDERFLOW;
                    int asm_l, asm_t, asm_u;
#ifdef HMCG_REPLAY_CHECK                              // (debug: the assembly loop first, then the C++ loop from the same state; differences -> status bits)
                    double av_in[K];
                    const int st_in = st;
#pragma unroll
                    for (int s = 0; s < K; ++s) av_in[s] = av[s];
#endif
#include "replay_asm_k8.inc"
                    (void)asm_l; (void)asm_t; (void)asm_u;
                    if (tid == own_t) sh.ulast = uxs[T - 1];
#ifdef HMCG_REPLAY_CHECK
                    __syncthreads();
                    for (int l = 0; l < L; ++l) chk_maps ^= (t0 + l >= 1 && t0 + l < T) ? maps[t0 + l - 1] * (uint32_t)(2 * l + 1) : 0u;
                    chk_pie = 0.0;
                    if (tid == own_t) { for (int s = 0; s < K; ++s) chk_pie += th.pi_end[s] * (s + 1); }
#pragma unroll
                    for (int s = 0; s < K; ++s) { av_asm[s] = av[s]; av[s] = av_in[s]; }
                    st_asm = st; st = st_in;
                    __syncthreads();
#else
                    replay_done = true;
#endif
                }
            }
            if (!replay_done) {
            double fnext[K];                     // pdfs of the step ahead, on their way from the scratch
#pragma unroll
            for (int s = 0; s < K; ++s) fnext[s] = fscr[fslot(0, s, threadIdx.x)];
#ifdef HMCG_BIG_NSC_REPLAY
            constexpr int NRES = NSC > 0 ? 0 : 2, NSR = NSC;      // the scalar columns serve the replay as well
#else
            constexpr int NRES = 2, NSR = 0;     // the first columns of A stay in registers: the step's first chains need not wait for LDS
#endif
            double a_first[NRES > 0 ? NRES : 1][K];
#pragma unroll
            for (int s = 0; s < NRES; ++s)
#pragma unroll
                for (int k = 0; k < K; ++k) a_first[s][k] = Atp[s * K + k];
            for (int l = 0; l < L; ++l) {
                asm volatile("" ::: "memory");       // as above: (the rest of) A stays in LDS
                const int t = t0 + l;
                const double u = uxs[t >= 1 ? t - 1 : 0];
                double fv[K];
#pragma unroll
                for (int s = 0; s < K; ++s) fv[s] = fnext[s];
                if (l + 1 < L) {
#pragma unroll
                    for (int s = 0; s < K; ++s) fnext[s] = fscr[fslot(l + 1, s, threadIdx.x)];
                }
                double nv[K], total = 0.0;
                uint32_t mok = 0;
                {
                    // the K cumulative-sum chains first, then the K categorical draws level by level (count_le_sorted_batch):
                    // one search after another is a chain of compare -> select -> compare through VCC, K of them in a row
                    double cum[K][K], thr[K];
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        double a[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) a[k] = s < NSR ? a_sc[s < NSR ? s : 0][k] : (s < NRES ? a_first[s < NRES ? s : 0][k] : Atp[s * K + k]);
                        double acc = av[0] * a[0];
                        cum[s][0] = acc;
#pragma unroll
                        for (int r = 1; r < K; ++r) { acc = fma(av[r], a[r], acc); cum[s][r] = acc; }
                        thr[s] = u * acc;
                        nv[s] = acc * fv[s];
                        total += nv[s];
                    }
                    int idx[K];
                    count_le_sorted_batch<K>(cum, thr, idx);
#pragma unroll
                    for (int s = 0; s < K; ++s) mok |= (uint32_t)idx[s] << (4 * s);
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(total > 0.0)) != 0ull, 0)) {
                    if (!(total > 0.0)) {
                        if (t < T) st |= HMCG_ST_EMIS_UNDERFLOW;
#pragma unroll
                        for (int s = 0; s < K; ++s) nv[s] = 1.0 / K;
                        total = 1.0;
                    }
                }
                const double inv = rcp_fast(total);
                int idx_uni = 0;
                if constexpr ((K & (K - 1)) == 0) {
                    idx_uni = (int)(u * (double)K);         // K a power of two: the cumulative j/K are exact, #{j/K <= u} = floor(K u), u < 1
                } else {
                    double cp = 0.0;
#pragma unroll
                    for (int r = 0; r < K - 1; ++r) { cp += 1.0 / K; idx_uni += (cp <= u) ? 1 : 0; }
                }
                // guards (:472-480): the entries whose pif[t,s] fails eps() take the uniform draw -- built as a nibble
                // mask so that the common case (no failure in the wave) costs one select per state
                uint32_t fail = 0;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    av[s] = nv[s] * inv;                                        // pif[t,s]
                    fail |= (av[s] > EPS64) ? 0u : (0xFu << (4 * s));
                }
                const uint32_t m = (mok & ~fail) | ((uint32_t)idx_uni * 0x11111111u & fail);
                if (t >= 1) maps[t - 1] = m;                                     // g_{t-1}; entries at/after T-1 are overridden below
                if (t == T - 1) {
#pragma unroll
                    for (int s = 0; s < K; ++s) th.pi_end[s] = av[s];
                    sh.ulast = uxs[T - 1];
                }
                if constexpr (SIG) {
                    if (tail > 0 && t == T - 1 - tail) {
#pragma unroll
                        for (int s = 0; s < K; ++s) sh.pf_rep[par][s] = av[s];
                    }
                }
                if (want_pif) {
                    if (t < T) {
#pragma unroll
                        for (int s = 0; s < K; ++s) p.pif_final[((size_t)w * p.ldY + t) * K + s] = av[s];
                    }
                }
            }
        }
        if constexpr (SM) {
            if (do_smooth) {
                // ---- backwardupdate_P! as the beta recursion b_{t-1} = A (f_t o b_t), b_{T-1} = 1; pib[t,:] ~ pif[t,:] o b_t.
                // b at the end of a thread's chunk = (product of the LATER chunks' matrices) * 1: within the wave an
                // exclusive suffix scan of the chunk products (prefix scan on lane-reversed data, multiplication order
                // flipped), then the later waves' totals -- the forward wave totals, already in LDS.
                double bw[K];
#pragma unroll
                for (int r = 0; r < K; ++r) bw[r] = 1.0;
                for (int ww = NW - 1; ww > wave_u; --ww) {
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
                    for (int r = 0; r < K; ++r) bw[r] = nb[r];
                }
                // lane-reversed copy, inclusive scan with the own matrix on the LEFT (column-wise DPP fetch of the source)
                double R[KK], R2[KK];
#pragma unroll
                for (int i = 0; i < KK; ++i) R[i] = __shfl(Qloc[i], 63 - lane, 64);
                scan_level_colwise<K, DPP_ROW_SHR1, 0xF>(R, R2);
                scan_level_colwise<K, DPP_ROW_SHR2, 0xF>(R2, R);
                rescale_pow2<KK>(R);
                scan_level_colwise<K, DPP_ROW_SHR4, 0xF>(R, R2);
                scan_level_colwise<K, DPP_ROW_SHR8, 0xF>(R2, R);
                rescale_pow2<KK>(R);
                scan_level_colwise<K, DPP_ROW_BCAST15, 0xA>(R, R2);
                scan_level_colwise<K, DPP_ROW_BCAST31, 0xC>(R2, R);
                rescale_pow2<KK>(R);
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
                double mu_s[K];
                int order[K];
#pragma unroll
                for (int i = 0; i < K; ++i) mu_s[i] = th.mu[i];
                sort_order<K>(mu_s, order);
                for (int l = L - 1; l >= 0; --l) {
                    asm volatile("" ::: "memory");
                    const int t = t0 + l;
                    if (t < T) {
                        double g[K], pfv[K], tot = 0.0;
                        const double* pft = p.pif_final + ((size_t)w * p.ldY + t) * K;
#pragma unroll
                        for (int s = 0; s < K; ++s) { pfv[s] = pft[s]; g[s] = pfv[s] * b[s]; tot += g[s]; }
                        const double inv = rcp_fast(tot);
#pragma unroll
                        for (int q = 0; q < K; ++q) {
                            double gq = 0.0, fq = 0.0;
#pragma unroll
                            for (int s = 0; s < K; ++s) { gq = (order[q] == s) ? g[s] : gq; fq = (order[q] == s) ? pfv[s] : fq; }
                            if (p.pi_smooth_mean) p.pi_smooth_mean[((size_t)w * p.ldY + t) * K + q] += gq * inv;   // sorted labels (:513)
                            if (p.pi_smooth_draws)                                                                  // samples.pib[d, t, q] (:558)
                                p.pi_smooth_draws[(size_t)p.nd_ld * ((size_t)q * p.ldY + t + (size_t)K * p.ldY * w) + (dk - p.draw_off)] = gq * inv;
                            if (p.pi_filter_mean) p.pi_filter_mean[((size_t)w * p.ldY + t) * K + q] += fq;         // sorted pif[t,:] (:512)
                        }
                        double fv[K], fb[K], nb[K];
#pragma unroll
                        for (int s = 0; s < K; ++s) fv[s] = fscr[fslot(l, s, threadIdx.x)];    // this step's pdfs, from the product phase
#pragma unroll
                        for (int s = 0; s < K; ++s) fb[s] = fv[s] * b[s];
#pragma unroll
                        for (int r = 0; r < K; ++r) {
                            double acc = 0.0;
#pragma unroll
                            for (int s = 0; s < K; ++s) acc = fma(th.A[r][s], fb[s], acc);
                            nb[r] = acc;
                        }
                        rescale_pow2<K>(nb);
#pragma unroll
                        for (int r = 0; r < K; ++r) b[r] = nb[r];
                    }
                }
            }
            }   // (!replay_done)
#ifdef HMCG_REPLAY_CHECK
            if constexpr (ASM_REPLAY) {
                if (!want_pif) {
                    __syncthreads();
                    uint32_t m2 = 0;
                    for (int l = 0; l < L; ++l) m2 ^= (t0 + l >= 1 && t0 + l < T) ? maps[t0 + l - 1] * (uint32_t)(2 * l + 1) : 0u;
                    bool bad_av = false;
#pragma unroll
                    for (int s = 0; s < K; ++s) bad_av |= __double_as_longlong(av_asm[s]) != __double_as_longlong(av[s]);
                    double pie2 = 0.0;
                    if (tid == (T - 1) / L) { for (int s = 0; s < K; ++s) pie2 += th.pi_end[s] * (s + 1); }
                    if (bad_av) st |= 0x100;
                    if (st_asm != (st & 0xFF)) st |= 0x200;
                    if (m2 != chk_maps) st |= 0x400;
                    if (tid == (T - 1) / L && pie2 != chk_pie) st |= 0x800;
                    __syncthreads();
                }
            }
#endif
        }
        if (tid == NT - 1) maps[cap - 1] = map_identity<K>();
        STAMP(7);
        __syncthreads();                                                     // Bd
        STAMP(8);
        // X[T-1] ~ Categorical(sorted pif[T-1,:]) (:464), redundantly in every thread
        int xlast = 0;
        {
            double mu_u[K];
            int order[K];
#pragma unroll
            for (int i = 0; i < K; ++i) mu_u[i] = th.mu[i];
            sort_order<K>(mu_u, order);
            const double ulast = sh.ulast;
            double cp = 0.0;
#pragma unroll
            for (int q = 0; q < K - 1; ++q) {
                cp += th.pi_end[order[q]];
                xlast += (cp <= ulast) ? 1 : 0;
            }
        }
        // ---- backward sampling: compose this thread's maps, suffix-scan over lanes and waves, apply ----
        // The maps stay 4-bit in LDS (one word per step); for composing they are widened to a byte per entry in two
        // words, where (a o b) is two v_perm_b32 -- the selector bytes 0..7 of b pick from the eight bytes {a.hi, a.lo} --
        // instead of eight shift-mask-shift-or rounds.
        ByteMap G = bytemap_identity();
        uint32_t mnext = maps[t0 + L - 1];           // (the next step's map is fetched one iteration ahead: the loop is a
        for (int l = L - 1; l >= 0; --l) {           //  dependent chain of v_perm's, the LDS latency must not be on it)
            const int t = t0 + l;
            uint32_t m = mnext;
            if (l > 0) mnext = maps[t - 1];
            m = (t == T - 1) ? map_const<K>(xlast) : (t > T - 1 ? map_identity<K>() : m);
            maps[t] = m;
            G = bytemap_compose(bytemap_from_nibbles(m), G);
        }
        STAMP(9);
        ByteMap Hm = G;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            ByteMap O;
            O.lo = __shfl_down(Hm.lo, d, 64);
            O.hi = __shfl_down(Hm.hi, d, 64);
            const ByteMap C = bytemap_compose(Hm, O);
            const bool in = lane + d < 64;
            Hm.lo = in ? C.lo : Hm.lo;
            Hm.hi = in ? C.hi : Hm.hi;
        }
        if (lane == 0) { sh.wmap[wave][0] = Hm.lo; sh.wmap[wave][1] = Hm.hi; }
        STAMP(10);
        __syncthreads();                                                     // Be
        STAMP(11);
        ByteMap Rw = bytemap_identity();
#pragma unroll
        for (int ww = NW - 1; ww >= 1; --ww) {
            ByteMap Wm;
            Wm.lo = sh.wmap[ww][0]; Wm.hi = sh.wmap[ww][1];
            const ByteMap C = bytemap_compose(Wm, Rw);
            Rw.lo = (ww > wave) ? C.lo : Rw.lo;
            Rw.hi = (ww > wave) ? C.hi : Rw.hi;
        }
        ByteMap Hx;
        Hx.lo = __shfl_down(Hm.lo, 1, 64);
        Hx.hi = __shfl_down(Hm.hi, 1, 64);
        if (lane == 63) Hx = bytemap_identity();
        int sin = (int)(bytemap_compose(Hx, Rw).lo & 0xFFu);               // entry 0 (a constant map below T-1)
        // apply the maps last step first -- and take the next sweep's sufficient statistics along: the state entering the
        // chunk is X[t0 + L] itself, so every transition (x_t -> x_{t+1}) is known where x_t is produced, and the pivots of
        // the one-pass sums are this sweep's means (what sh.pivot becomes below).  One pass over the window instead of two.
        StatAcc sa;
        stats_begin(sa);
        uint32_t anext = maps[t0 + L - 1];
        double ynext = ylds[t0 + L - 1];
        for (int l = L - 1; l >= 0; --l) {
            const int t = t0 + l;
            const uint32_t am = anext;
            const double yv = ynext;
            if (l > 0) { anext = maps[t - 1]; ynext = ylds[t - 1]; }
            if (t < T) {
                const int xn = sin;
                sin = map_apply(am, sin);
                xs[t] = (uint8_t)sin;
                stats_step(sa, t, sin, min(xn, K - 1), yv, th.mu[sin]);
            }
        }
        x_end = xlast;
        if (tid < K) sh.pivot[tid] = th.mu[tid];                             // pivots of these statistics (checkpoint; stand-alone form)
        STAMP(12);
        stats_finish(sa);
        if constexpr (SIG) __syncthreads();                                  // Bf: xs, pivot complete (a new noise sample re-reads them)
        STAMP(13);
    }
#ifdef HMCG_STAMPS
    if (lane == 0 && p.dbg) {
        unsigned long long* o = p.dbg + ((size_t)w * NW + wave) * HMCG_NSTAMP_ALL;
        for (int i = 0; i < HMCG_NSTAMP; ++i) o[i] = stamp_acc[i];
        o[HMCG_NSTAMP] = __builtin_amdgcn_s_memtime() - stamp_t0;
        o[HMCG_NSTAMP + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif

    // ---- epilogue ----
    __syncthreads();
    if (p.sweep_end > p.sweep_begin) job_outputs(p.sweep_end - 1);
    if (p.xstate) for (int t = tid; t < T; t += NT) p.xstate[(size_t)w * p.ldY + t] = xs[t];
    if (p.x_final) for (int t = tid; t < T; t += NT) p.x_final[(size_t)w * p.ldY + t] = xs[t];
    if (wave == OUT_WAVE) {
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int orole = lane + 64 * q;
            if (orole < NP) {
                if (p.sumacc) p.sumacc[(size_t)w * NCK + orole] = sum_par[q];
                if (p.summary && p.final_launch) p.summary[(size_t)w * NS + orole] = p.nd > 0 ? sum_par[q] / (double)p.nd : __builtin_nan("");
            }
        }
    }
    if (fc_e >= 0) {
        if (p.sumacc) p.sumacc[(size_t)w * NCK + NP + fc_e] = sum_fc;
        if (p.summary && p.final_launch) p.summary[(size_t)w * NS + NP + fc_e] = p.nd > 0 ? sum_fc / (double)p.nd : __builtin_nan("");
    }
    if constexpr (SIG) {
        // a launch that stops inside a noise sample leaves that sample's running sums in its row (checkpoint)
        if (p.sample_summary) {
            const int smp = p.sweep_end / p.per_sample, kb = p.sweep_end - smp * p.per_sample - p.burnin_s;
            if (kb > 0 && kb < p.nrun_s && smp < p.n_samples) {
                double* row = p.sample_summary + ((size_t)w * p.n_samples + smp) * NS;
                if (wave == OUT_WAVE) {
#pragma unroll
                    for (int q = 0; q < NPASS; ++q) if (lane + 64 * q < NP) row[lane + 64 * q] = smp_par[q];
                }
                if (fc_e >= 0) row[NP + fc_e] = smp_fc;
            }
        }
    }
    if (p.sumacc && tid < K) p.sumacc[(size_t)w * NCK + NS + tid] = sh.pivot[tid];
    if constexpr (SM) {
        if (p.final_launch && p.nd > 0) {                        // running sums -> means (every thread its own steps)
            const double sc = 1.0 / (double)p.nd;
            for (int l = 0; l < L; ++l) {
                const int t = t0 + l;
                if (t < T) {
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        if (p.pi_smooth_mean) p.pi_smooth_mean[((size_t)w * p.ldY + t) * K + q] *= sc;
                        if (p.pi_filter_mean) p.pi_filter_mean[((size_t)w * p.ldY + t) * K + q] *= sc;
                    }
                }
            }
        }
    }
    if (st) atomicOr(&p.status[w], st);
}

}  // namespace hmcg
