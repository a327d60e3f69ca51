/*
 * oracle/hmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64, sequential in time) of the Gibbs-sampled
 * Gaussian-HMM hot path of joe5saia/Hmc.jl.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's library; the product
 * path (hmc.jl_amd/csrc) never links or calls it.
 *
 * Every function cites the reference lines (src/Hmc.jl) it follows.  Operation
 * order inside the deterministic kernels follows the reference statement by
 * statement (s-outer/r-inner accumulation, per-step renormalisation, two-pass
 * SSE, materialised Pf/Pb, quirks 1-9 of SURVEY.md section 8a).
 *
 * PARITY STATUS.  The reference is Julia; no Julia runtime exists in the build
 * container, so the reference itself was never run.  Third-party arithmetic on
 * the path (Julia stdlib Random = dSFMT MersenneTwister + ziggurat randn/randexp,
 * Distributions.jl 0.21.8 samplers, StatsFuns 0.9.0 normpdf; Manifest.toml:190,
 * :475, :566) is NOT reproduced bit-for-bit:
 *   - RNG stream / draw-level parity with Julia:  "parity unpinned".
 *   - what IS pinned: (1) the committed posterior summaries
 *     data/output/official/\*_summary.csv (tests/golden/official_*), statistically;
 *     (2) the reference unit test's truth-recovery tolerances (test/runtests.jl:56-57);
 *     (3) hand-derived known-answer vectors for each deterministic kernel;
 *     (4) for the signal path (estimatesignals!, :868-914) the committed allsignal dispersion outputs
 *     (tests/golden/signals_noise_*).
 *   - signals past the end date (sigLen > 0, :888,:900,:906-910 -- end_pos / blend_mask of
 *     hmco_estimate_window_ex) have no committed reference output: "parity unpinned" for that sub-case
 *     beyond the line-by-line restatement and the consistency checks in tests/.
 *
 * RNG SPEC (shared, by restatement, with the HIP path so that seeded chains agree
 * draw for draw): Philox4x32-10, key = 64-bit seed, counter = (index,
 * site<<16 | element, sweep, window).  Sites follow the reference's draw order per
 * sweep: 0 sigma^2_i (InvGamma), 1 mu_i (Normal), 2 rho_i (Gamma(1)),
 * 3 A_ij (Gamma(count)), 4 X_t (uniform; t = 0-based time index).
 *
 * Sampler transforms (our spec; stand-ins for Distributions.jl's):
 *   uniform  u = ((r0<<21)|(r1>>11)) * 2^-53 in [0,1)
 *   normal   Box-Muller  sqrt(-2 log(1-u1)) cos(2 pi u2), one Philox block
 *   gamma    shape==1: -log(1-u);  shape>1: Marsaglia-Tsang, attempt j uses
 *            blocks index 2j (normal) and 2j+1 (accept uniform);
 *            shape<1: MT(shape+1) * U^(1/shape), U from block index 0xFFFFFFFF
 *   categorical  single uniform + linear CDF scan in state order
 *            (cp = p[1]; while cp <= u && i < n: cp += p[++i]) as
 *            Distributions 0.21.8 DiscreteNonParametric does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HMCO_MAXK 16
#define HMCO_MAXH 8
#define HMCO_GAMMA_MAX_ATTEMPTS 64

/* status bits (same meaning as include/hmcg.h) */
#define ST_BAD_INVGAMMA 1   /* a<=0 or b<=0: old sigma kept (src/Hmc.jl:319-329) */
#define ST_EMIS_UNDERFLOW 2 /* all K emission pdfs < 1e-300 at some t: observation treated as missing */
#define ST_NONFINITE 4      /* non-finite input */
#define ST_GAMMA_CAP 8      /* gamma rejection loop hit its attempt cap */

enum { SITE_SIG2 = 0, SITE_MU = 1, SITE_RHO = 2, SITE_A = 3, SITE_X = 4, SITE_NOISE = 5 };

/* ------------------------------------------------------------------ RNG -- */

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0];
        uint64_t p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

typedef struct {
    uint64_t seed;
    uint32_t window;
    uint32_t sweep;
} rng_t;

static void rng_block(const rng_t *g, uint32_t site, uint32_t elem, uint32_t idx, uint32_t out[4])
{
    out[0] = idx;
    out[1] = (site << 16) | elem;
    out[2] = g->sweep;
    out[3] = g->window;
    philox4x32_10(out, (uint32_t)g->seed, (uint32_t)(g->seed >> 32));
}

static double u53(uint32_t a, uint32_t b)
{
    uint64_t x = ((uint64_t)a << 21) | (uint64_t)(b >> 11);
    return (double)x * 0x1.0p-53;
}

static const double TWO_PI = 6.283185307179586476925286766559;

static double box_muller(const uint32_t r[4])
{
    double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
    return sqrt(-2.0 * log(1.0 - u1)) * cos(TWO_PI * u2);
}

static double rng_normal(const rng_t *g, uint32_t site, uint32_t elem, uint32_t idx)
{
    uint32_t r[4];
    rng_block(g, site, elem, idx, r);
    return box_muller(r);
}

/* Gamma(shape, 1).  Stand-in for rand(Gamma) of Distributions 0.21.8. */
static double rng_gamma(const rng_t *g, uint32_t site, uint32_t elem, double shape, int *status)
{
    uint32_t r[4];
    if (shape == 1.0) {
        rng_block(g, site, elem, 0, r);
        return -log(1.0 - u53(r[0], r[1]));
    }
    double a = shape < 1.0 ? shape + 1.0 : shape;
    double d = a - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    double out = d; /* value used if the attempt cap is ever hit */
    int j;
    for (j = 0; j < HMCO_GAMMA_MAX_ATTEMPTS; ++j) {
        rng_block(g, site, elem, 2u * (uint32_t)j, r);
        double x = box_muller(r);
        rng_block(g, site, elem, 2u * (uint32_t)j + 1u, r);
        double u = u53(r[0], r[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(1.0 - u) < 0.5 * x * x + d - d * v + d * log(v)) { out = d * v; break; }
    }
    if (j == HMCO_GAMMA_MAX_ATTEMPTS) *status |= ST_GAMMA_CAP;
    if (shape < 1.0) {
        rng_block(g, site, elem, 0xFFFFFFFFu, r);
        out *= pow(1.0 - u53(r[0], r[1]), 1.0 / shape);
    }
    return out;
}

static double rng_uniform_x(const rng_t *g, uint32_t t)
{
    uint32_t r[4];
    rng_block(g, SITE_X, 0, t >> 1, r);
    return (t & 1u) ? u53(r[2], r[3]) : u53(r[0], r[1]);
}

/* Categorical draw: Distributions 0.21.8 DiscreteNonParametric rand (call sites
 * src/Hmc.jl:464,481).  Returns a 0-based state. */
static int categorical(const double *p, int n, double u)
{
    double cp = p[0];
    int i = 0;
    while (cp <= u && i < n - 1) { ++i; cp += p[i]; }
    return i;
}

/* ---------------------------------------------------- small numerics ----- */

static const double INVSQRT2PI = 0.3989422804014327;

/* StatsFuns 0.9.0 normpdf(mu, sd, x) = exp(-z^2/2) * invsqrt2pi / sd, z=(x-mu)/sd */
static double normpdf(double mu, double sd, double x)
{
    double z = (x - mu) / sd;
    return exp(-(z * z) / 2.0) * INVSQRT2PI / sd;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Julia round(x; digits=5): rint(x*1e5)/1e5 (basicsave, src/Hmc.jl:719) */
static double round5(double x)
{
    double r = rint(x * 1e5) / 1e5;
    return isfinite(r) ? r : x;
}

/* ------------------------------------------------------- chain state ----- */

typedef struct {
    int K, T;
    const double *Y;
    /* hyper-parameters: 4-arg HyperParams (src/Hmc.jl:132-142) */
    double xi, alpha, nu;
    /* chain */
    double mu[HMCO_MAXK], sig2[HMCO_MAXK], beta[HMCO_MAXK], rho[HMCO_MAXK];
    double A[HMCO_MAXK][HMCO_MAXK];
    double *pif, *pib;   /* [T][K] */
    double *Pf, *Pb;     /* [T][K][K] */
    int *X;              /* [T], 0-based states */
    int *obs_index;      /* [T] stand-in for opt.obsRangeit (faithful-cost mode) */
    int faithful_cost;
    int status;
    /* signal set (src/Hmc.jl:23,29): 0-based half-open range of window positions that are "signals";
     * empty for the live estimatemodel caller.  kappa = hp.kappa (relative noise of a signal). */
    int sig_b, sig_e;
    double kappa;
} chain_t;

#define PIF(c, t, s) ((c)->pif[(size_t)(t) * (c)->K + (s)])
#define PIB(c, t, s) ((c)->pib[(size_t)(t) * (c)->K + (s)])
#define PF(c, t, r, s) ((c)->Pf[((size_t)(t) * (c)->K + (r)) * (c)->K + (s)])
#define PB(c, t, r, s) ((c)->Pb[((size_t)(t) * (c)->K + (r)) * (c)->K + (s)])

/* makeParams (src/Hmc.jl:161-195) + HyperParams(Y,D) (src/Hmc.jl:132-142). */
static void chain_init(chain_t *c, const int *x_init)
{
    const int K = c->K, T = c->T;
    const double *Y = c->Y;
    double ymin = Y[0], ymax = Y[0], sum = 0.0;
    for (int t = 0; t < T; ++t) {
        if (Y[t] < ymin) ymin = Y[t];
        if (Y[t] > ymax) ymax = Y[t];
        sum += Y[t];
    }
    double mean = sum / T;
    double ss = 0.0;
    for (int t = 0; t < T; ++t) ss += (Y[t] - mean) * (Y[t] - mean);
    double sd = T > 1 ? sqrt(ss / (T - 1)) : 0.0;          /* std(Y), :177 */
    double *tmp = (double *)malloc(sizeof(double) * (size_t)T);
    memcpy(tmp, Y, sizeof(double) * (size_t)T);
    qsort(tmp, (size_t)T, sizeof(double), cmp_double);
    double med = (T & 1) ? tmp[T / 2] : tmp[T / 2 - 1] / 2 + tmp[T / 2] / 2; /* Statistics.median */
    free(tmp);
    double R = ymax - ymin;                                 /* :175 */
    double lo = med - 0.25 * R, hi = med + 0.25 * R;        /* :176 */
    for (int k = 0; k < K; ++k) {
        c->mu[k] = K > 1 ? lo + (hi - lo) * ((double)k / (double)(K - 1)) : lo;
        c->sig2[k] = sd;      /* quirk 1: the "variance" slot starts at std(Y) */
        c->rho[k] = 1.0 / K;  /* :178 */
        c->beta[k] = 1.0;     /* :179 */
        for (int j = 0; j < K; ++j) c->A[k][j] = 1.0 / K; /* :171 */
    }
    if (K > 1) c->mu[K - 1] = hi;
    for (int t = 0; t < T; ++t) {                           /* :185-187 */
        if (x_init) { c->X[t] = x_init[t]; continue; }
        /* findmax(pdf.(Normal.(mu, sigma), Y[t]))[2] with every sigma equal (= std(Y), used as an sd
         * here): the largest pdf belongs to the nearest initial mean, first index on ties.  We compare
         * distances rather than pdf values: the two agree except where libm's exp rounds two
         * different arguments to the same double, and that sub-ulp case is structural, not rare --
         * for even K and odd T the median observation sits exactly midway between the two middle
         * means -- so a rule that does not depend on the exp implementation is needed for the
         * GPU path and this oracle to start from the same X. */
        int best = 0;
        double bd = fabs(Y[t] - c->mu[0]);
        for (int k = 1; k < K; ++k) {
            double d = fabs(Y[t] - c->mu[k]);
            if (d < bd) { bd = d; best = k; }
        }
        c->X[t] = best;
    }
    c->xi = mean;      /* :136 */
    c->alpha = 1.0;    /* :137 */
    c->nu = 1.0;       /* :140 */
    for (int t = 0; t < T; ++t) c->obs_index[t] = t + 1;
}

/* `t in opt.obsRangeit` (src/Hmc.jl:409): a linear search of a length-T array on
 * every step.  Only executed in faithful-cost mode; the answer is always "yes"
 * for the live caller (signalRange empty). */
static int in_obs_range(const chain_t *c, int t1)
{
    if (!c->faithful_cost) return 1;
    const volatile int *idx = c->obs_index;
    for (int i = 0; i < c->T; ++i)
        if (idx[i] == t1) return 1;
    return 0;
}

/* update_mu_sigma (src/Hmc.jl:231-336): observation set and signal set (empty for the live caller). */
static void update_mu_sigma(chain_t *c, const rng_t *g)
{
    const int K = c->K, T = c->T;
    const double kap = c->kappa;
    long Ni[HMCO_MAXK], Mi[HMCO_MAXK];
    double S[HMCO_MAXK], Sm[HMCO_MAXK], ybar[HMCO_MAXK], sbar[HMCO_MAXK], totalbar[HMCO_MAXK], S2[HMCO_MAXK], Sm2[HMCO_MAXK];
    for (int i = 0; i < K; ++i) { Ni[i] = 0; Mi[i] = 0; S[i] = 0.0; Sm[i] = 0.0; S2[i] = 0.0; Sm2[i] = 0.0; }
    for (int t = 0; t < T; ++t) {                                                     /* :254-258, :268-272 */
        int i = c->X[t];
        if (t >= c->sig_b && t < c->sig_e) { Mi[i] += 1; Sm[i] += c->Y[t]; }
        else { Ni[i] += 1; S[i] += c->Y[t]; }
    }
    for (int i = 0; i < K; ++i) {
        ybar[i] = Ni[i] > 0 ? S[i] / (double)Ni[i] : 0.0;                             /* :259-265 */
        sbar[i] = Mi[i] > 0 ? Sm[i] / (double)Mi[i] : 0.0;                            /* :273-279 */
        totalbar[i] = (Ni[i] + Mi[i]) > 0 ? (S[i] + Sm[i]) / (double)(Ni[i] + Mi[i]) : 0.0; /* :282-288 */
    }
    for (int t = 0; t < T; ++t) {                                                     /* :291-300 */
        int i = c->X[t];
        if (t >= c->sig_b && t < c->sig_e) { double d = c->Y[t] - sbar[i]; Sm2[i] += d * d; }
        else { double d = c->Y[t] - ybar[i]; S2[i] += d * d; }
    }
    double Neff[HMCO_MAXK];
    for (int i = 0; i < K; ++i) {
        double Meff = (double)Mi[i] / (1.0 + kap);                                    /* :302 */
        Neff[i] = (double)Ni[i] + Meff;                                               /* :303 */
        double a = c->alpha + 0.5 * (double)Ni[i] + 0.5 * (double)Mi[i];              /* :313 */
        double dm = totalbar[i] - c->xi;
        double b = c->beta[i] + 0.5 * S2[i] + (0.5 / (1.0 + kap)) * Sm2[i]
                 + 0.5 * Neff[i] * c->nu / (Neff[i] + c->nu) * (dm * dm);             /* :314 */
        if (a > 0.0 && b > 0.0) {
            /* InverseGamma(a,b) = 1/Gamma(a, scale 1/b)  (:320) */
            double gdraw = rng_gamma(g, SITE_SIG2, (uint32_t)i, a, &c->status);
            c->sig2[i] = 1.0 / (gdraw * (1.0 / b));
        } else {
            c->status |= ST_BAD_INVGAMMA;                                             /* :321-329 keeps old */
        }
    }
    for (int i = 0; i < K; ++i) {
        double m = (S[i] + Sm[i] + c->nu * c->xi) / (Neff[i] + c->nu);                /* :331 (Sm not divided by 1+kappa: quirk 4) */
        double sdev = sqrt(c->sig2[i] / (Neff[i] + c->nu));                           /* :332 */
        c->mu[i] = m + sdev * rng_normal(g, SITE_MU, (uint32_t)i, 0);                 /* :334 */
    }
}

/* update_beta (src/Hmc.jl:338-348): the Gamma draw is commented out upstream. */
static void update_beta(chain_t *c) { for (int i = 0; i < c->K; ++i) c->beta[i] = 2.0; }

/* update_rho (src/Hmc.jl:350-356): Dirichlet(ones(K)), independent of X. */
static void update_rho(chain_t *c, const rng_t *g)
{
    double s = 0.0;
    for (int i = 0; i < c->K; ++i) { c->rho[i] = rng_gamma(g, SITE_RHO, (uint32_t)i, 1.0, &c->status); s += c->rho[i]; }
    double inv = 1.0 / s;
    for (int i = 0; i < c->K; ++i) c->rho[i] *= inv;
}

/* update_A (src/Hmc.jl:358-369): counts + 1, each row ~ Dirichlet. */
static void update_A(chain_t *c, const rng_t *g)
{
    const int K = c->K;
    long Trans[HMCO_MAXK][HMCO_MAXK];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) Trans[i][j] = 1;
    for (int t = 0; t + 1 < c->T; ++t) Trans[c->X[t]][c->X[t + 1]] += 1;
    for (int i = 0; i < K; ++i) {
        double s = 0.0;
        for (int j = 0; j < K; ++j) {
            c->A[i][j] = rng_gamma(g, SITE_A, (uint32_t)(i * K + j), (double)Trans[i][j], &c->status);
            s += c->A[i][j];
        }
        double inv = 1.0 / s;
        for (int j = 0; j < K; ++j) c->A[i][j] *= inv;
    }
}

/* forwardupdate_P (src/Hmc.jl:371-440).  Direct-probability domain, per-step
 * renormalisation, s-outer / r-inner accumulation.  Extension (reference would
 * produce NaN and throw): if every emission pdf at step t is < 1e-300 the
 * observation is treated as missing (f = 1) and ST_EMIS_UNDERFLOW is raised. */
static void forward_update(chain_t *c)
{
    const int K = c->K, T = c->T;
    double sd[HMCO_MAXK], f[HMCO_MAXK];
    double sds[HMCO_MAXK];
    for (int s = 0; s < K; ++s) { sd[s] = sqrt(c->sig2[s]); sds[s] = (1.0 + c->kappa) * sqrt(c->sig2[s]); }   /* :381-382 (quirk 4) */
    for (int t = 0; t < T; ++t) {
        (void)in_obs_range(c, t + 1);                                                 /* :387, :409 */
        const double *sdt = (t >= c->sig_b && t < c->sig_e) ? sds : sd;
        double fmax = 0.0;
        if (c->faithful_cost) {
            /* the reference evaluates the pdf inside the r loop: K^2 evaluations (:415) */
            for (int s = 0; s < K; ++s) for (int r = 0; r < K; ++r) { volatile double v = normpdf(c->mu[s], sdt[s], c->Y[t]); f[s] = v; }
        } else {
            for (int s = 0; s < K; ++s) f[s] = normpdf(c->mu[s], sdt[s], c->Y[t]);
        }
        for (int s = 0; s < K; ++s) if (f[s] > fmax) fmax = f[s];
        if (!(fmax >= 1e-300)) {
            /* every pdf underflowed: the observation carries no usable likelihood; it is treated as
             * missing (f = 1: the step becomes the prediction pi[t-1,:]*A) and the window is flagged */
            c->status |= ST_EMIS_UNDERFLOW;
            for (int s = 0; s < K; ++s) f[s] = 1.0;
        }
        double total = 0.0;
        for (int s = 0; s < K; ++s)
            for (int r = 0; r < K; ++r) {
                double prev = t == 0 ? c->rho[r] : PIF(c, t - 1, r);                  /* :390 / :415 */
                double v = prev * c->A[r][s] * f[s];
                PF(c, t, r, s) = v;
                total += v;
            }
        for (int s = 0; s < K; ++s) {                                                 /* :429-432 */
            double acc = 0.0;
            for (int r = 0; r < K; ++r) { PF(c, t, r, s) /= total; acc += PF(c, t, r, s); }
            PIF(c, t, s) = acc;
        }
    }
}

/* backwardupdate_P (src/Hmc.jl:442-457): deterministic smoother. */
static void backward_update(chain_t *c)
{
    const int K = c->K, T = c->T;
    for (int s = 0; s < K; ++s) {
        PIB(c, T - 1, s) = PIF(c, T - 1, s);
        for (int r = 0; r < K; ++r) PB(c, T - 1, r, s) = PF(c, T - 1, r, s);
    }
    for (int t = T - 2; t >= 0; --t) {
        for (int r = 0; r < K; ++r) PIB(c, t, r) = 0.0;
        for (int s = 0; s < K; ++s) for (int r = 0; r < K; ++r) PIB(c, t, r) += PB(c, t + 1, r, s);
        for (int s = 0; s < K; ++s) for (int r = 0; r < K; ++r)
            PB(c, t, r, s) = PF(c, t, r, s) * PIB(c, t, s) / PIF(c, t, s);
    }
}

/* update_X (src/Hmc.jl:459-484).  pi = label-sorted filter, P = UNSORTED Pf
 * (quirk 5): X[T-1] is drawn in sorted labels, the rest index unsorted Pf. */
static void update_X(chain_t *c, const rng_t *g, const int *order)
{
    const int K = c->K, T = c->T;
    double p[HMCO_MAXK];
    for (int s = 0; s < K; ++s) p[s] = PIF(c, T - 1, order[s]);
    c->X[T - 1] = categorical(p, K, rng_uniform_x(g, (uint32_t)(T - 1)));             /* :464 */
    for (int k = T - 2; k >= 0; --k) {
        int k2 = k + 1, s = c->X[k2];
        double total = 0.0;
        for (int r = 0; r < K; ++r) { p[r] = PF(c, k2, r, s); total += p[r]; }        /* :468-471 */
        if (total > 2.220446049250313e-16) for (int j = 0; j < K; ++j) p[j] /= total; /* :472-475 */
        else for (int j = 0; j < K; ++j) p[j] = 1.0 / K;                              /* :476-480 */
        c->X[k] = categorical(p, K, rng_uniform_x(g, (uint32_t)k));                   /* :481 */
    }
}

/* gibbssweep (src/Hmc.jl:486-515).  `order` receives sortperm(mu). */
static void gibbs_sweep(chain_t *c, const rng_t *g, int *order, int do_smoother)
{
    const int K = c->K, T = c->T;
    update_mu_sigma(c, g);
    update_beta(c);
    update_rho(c, g);
    update_A(c, g);
    forward_update(c);
    if (do_smoother) backward_update(c);
    else for (int s = 0; s < K; ++s) PIB(c, T - 1, s) = PIF(c, T - 1, s);
    /* sortperm(mu): stable insertion sort (:501) */
    for (int i = 0; i < K; ++i) order[i] = i;
    for (int i = 1; i < K; ++i) {
        int o = order[i], j = i - 1;
        while (j >= 0 && c->mu[order[j]] > c->mu[o]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
    }
    double tv[HMCO_MAXK], tA[HMCO_MAXK][HMCO_MAXK];
    for (int i = 0; i < K; ++i) tv[i] = c->mu[order[i]];
    memcpy(c->mu, tv, sizeof(double) * (size_t)K);                                    /* :502 */
    for (int i = 0; i < K; ++i) tv[i] = c->sig2[order[i]];
    memcpy(c->sig2, tv, sizeof(double) * (size_t)K);                                  /* :503 */
    for (int i = 0; i < K; ++i) tv[i] = c->rho[order[i]];
    memcpy(c->rho, tv, sizeof(double) * (size_t)K);                                   /* :505 */
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) tA[i][j] = c->A[order[i]][order[j]];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) c->A[i][j] = tA[i][j];    /* :506-511 */
    /* pif/pib columns are permuted upstream (:512-513) but Pf is not; we keep pif
     * unsorted in memory and apply `order` wherever the sorted view is read. */
    update_X(c, g, order);
}

/* forecast (src/Hmc.jl:658-667): (pi' * A^h) . mu ; A^h by Julia's
 * power_by_squaring schedule, naive k-ordered dot products. */
static void matmul(int K, double out[HMCO_MAXK][HMCO_MAXK], double a[HMCO_MAXK][HMCO_MAXK], double b[HMCO_MAXK][HMCO_MAXK])
{
    double t[HMCO_MAXK][HMCO_MAXK];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += a[i][k] * b[k][j];
        t[i][j] = acc;
    }
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) out[i][j] = t[i][j];
}

static int ctz_u(unsigned p) { int n = 0; while (!(p & 1u)) { p >>= 1; ++n; } return n; }

static void matpow(int K, double out[HMCO_MAXK][HMCO_MAXK], double a[HMCO_MAXK][HMCO_MAXK], int h)
{
    double x[HMCO_MAXK][HMCO_MAXK], y[HMCO_MAXK][HMCO_MAXK];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) { x[i][j] = a[i][j]; out[i][j] = (i == j); }
    if (h <= 0) return;
    unsigned p = (unsigned)h;
    int t = ctz_u(p) + 1;
    p >>= t;
    while (--t > 0) matmul(K, x, x, x);
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) y[i][j] = x[i][j];
    while (p > 0) {
        t = ctz_u(p) + 1;
        p >>= t;
        while (--t >= 0) matmul(K, x, x, x);
        matmul(K, y, y, x);
    }
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) out[i][j] = y[i][j];
}

double hmco_forecast(int K, const double *mu, const double *A_rowmajor, const double *pi_end, int h)
{
    double a[HMCO_MAXK][HMCO_MAXK], ah[HMCO_MAXK][HMCO_MAXK];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) a[i][j] = A_rowmajor[i * K + j];
    matpow(K, ah, a, h);
    double f = 0.0;
    for (int s = 0; s < K; ++s) {
        double s1 = 0.0;
        for (int r = 0; r < K; ++r) s1 += pi_end[r] * ah[r][s];
        f += s1 * mu[s];
    }
    return f;
}

/* ------------------------------------------------------------ entry ------ */

/* gibbssample + estimatemodel (src/Hmc.jl:517-562, 850-865) for ONE window, generalised to the
 * signal Monte-Carlo loop of estimatesignals! (src/Hmc.jl:868-914) when n_samples/sigma_signal say so.
 * Output arrays use the Julia (column-major) layouts of the reference, nd = n_samples*nrun draws:
 *   mu, sig2, pi_end : (nd, K)      -> [k*nd + d]
 *   A                : (nd, K, K)   -> [(j*K + i)*nd + d]
 *   fcast            : (nd, 2H)     -> [(2h+{0,1})*nd + d]
 *   pi_smooth        : (nd, T, K)   -> [(k*T + t)*nd + d]   (optional)
 *   summary          : 3K + K^2 + 2H means of 5-digit-rounded draws, order
 *                      mu | sig2 | pi_end | A(:) col-major | forecasts  (optional)
 *   sigvals          : (n_samples, save_e-save_b): Yfake[signalSave] of each noise sample (optional)
 * Signal model: positions [sig_b, sig_e) (0-based, half-open; must end at T: sigLen = 0, the case the
 * reference's committed outputs cover) are signals with relative noise kappa.  Each of the n_samples
 * chains runs burnin+nrun sweeps on Yfake = Y + N(0,1)*sigma_signal over the signal range (:892), and
 * the chain state is carried from one sample to the next as upstream does (:889-895).  Hyper-
 * parameters alpha, nu are passed in (1,1 for HyperParams(Y,D) :132-142; 2,2 for HyperParams(opt)
 * :148-159); xi is always the mean of the REAL window.
 * flags bit0: faithful-cost mode; bit1: run the backward smoother every sweep.
 * Returns 0, or -1 on bad arguments. */
int hmco_estimate_window_ex(const double *Y, int T, int K, int burnin, int nrun,
                            const int *horizons, int H, const double *yreal,
                            uint64_t seed, uint32_t window_id, int flags, const int *x_init,
                            int sig_b, int sig_e, double kappa, double alpha, double nu,
                            int n_samples, double sigma_signal, int save_b, int save_e,
                            int end_pos, int blend_mask,
                            double *mu, double *sig2, double *A, double *pi_end, double *fcast,
                            double *pi_smooth, double *summary, double *sigvals,
                            int *x_final, double *pif_final, double *pi_filter_mean, int *status)
{
    /* pi_filter_mean [T][K]: mean over the kept draws of the label-sorted FILTERED probabilities pif[t,:] (:512) --
     * what data/output/official_insample/forecats_insample.csv holds in s1..s3 (an older API's "pib").
     * end_pos: 0-based position whose SMOOTHED state probabilities are reported as pi_end -- the reference's
     * samples.pib[:, opt.endIndex, :] (:900) when the window carries sigLen = T-1-end_pos signal steps past the
     * end date (:888); negative or T-1 = the last step (smoothed == filtered there).
     * blend_mask bit k: horizon k equals sigLen and is reported through forecastsignal (:670-681, :908-909)
     * with signal = Yfake[T-1] and noise = sigma_signal; the other horizons go through forecast() with the
     * horizon the caller passes (h - sigLen, :906-907), always from the LAST step's probabilities. */
    if (K < 1 || K > HMCO_MAXK || T < 2 || H < 0 || H > HMCO_MAXH || nrun < 0 || burnin < 0 || n_samples < 1) return -1;
    if (sig_b < sig_e && (sig_b < 0 || sig_e != T)) return -1;       /* the signal set is a tail of the window */
    const int rep_pos = (end_pos >= 0 && end_pos < T - 1) ? end_pos : T - 1;
    chain_t c;
    memset(&c, 0, sizeof c);
    c.K = K; c.T = T;
    c.faithful_cost = flags & 1;
    c.sig_b = sig_b < sig_e ? sig_b : T; c.sig_e = sig_b < sig_e ? sig_e : T; c.kappa = kappa;
    int smoother = (flags & 2) || pi_smooth != NULL || rep_pos != T - 1;
    for (int t = 0; t < T; ++t) if (!isfinite(Y[t])) { if (status) *status = ST_NONFINITE; return 0; }
    double *Yfake = (double *)malloc(sizeof(double) * (size_t)T);
    memcpy(Yfake, Y, sizeof(double) * (size_t)T);                   /* Yfake = deepcopy(Yreal) (:887) */
    c.Y = Yfake;
    c.pif = (double *)malloc(sizeof(double) * (size_t)T * K);
    c.pib = (double *)malloc(sizeof(double) * (size_t)T * K);
    c.Pf = (double *)malloc(sizeof(double) * (size_t)T * K * K);
    c.Pb = (double *)malloc(sizeof(double) * (size_t)T * K * K);
    c.X = (int *)malloc(sizeof(int) * (size_t)T);
    c.obs_index = (int *)malloc(sizeof(int) * (size_t)T);
    chain_init(&c, x_init);                                          /* makeParams on the real data; xi = mean(Yreal) */
    c.alpha = alpha; c.nu = nu;
    const int NS = 3 * K + K * K + 2 * H;
    const int nd = n_samples * nrun;
    if (pi_filter_mean) for (int i = 0; i < T * K; ++i) pi_filter_mean[i] = 0.0;
    double acc[3 * HMCO_MAXK + HMCO_MAXK * HMCO_MAXK + 2 * HMCO_MAXH];
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    int order[HMCO_MAXK];
    rng_t g = { seed, window_id, 0 };
    for (int smp = 0; smp < n_samples; ++smp) {
        if (sigma_signal != 0.0 || n_samples > 1) {                  /* :892: fresh noise on the signal range */
            rng_t gn = { seed, window_id, (uint32_t)smp };
            for (int t = c.sig_b; t < c.sig_e; ++t)
                Yfake[t] = Y[t] + rng_normal(&gn, SITE_NOISE, 0, (uint32_t)t) * sigma_signal;
        }
        if (sigvals) for (int t = save_b; t < save_e; ++t) sigvals[(size_t)smp * (save_e - save_b) + (t - save_b)] = Yfake[t];
        for (int it = 0; it < burnin + nrun; ++it) {
            g.sweep = (uint32_t)(smp * (burnin + nrun) + it);
            gibbs_sweep(&c, &g, order, smoother);
            if (it < burnin) continue;
            const int d = smp * nrun + (it - burnin);
            double pe[HMCO_MAXK], arow[HMCO_MAXK * HMCO_MAXK];
            for (int k = 0; k < K; ++k) {
                pe[k] = PIF(&c, T - 1, order[k]);      /* pib[end,:] == sorted pif[end,:] (:448,:513) */
                const double prep = rep_pos == T - 1 ? pe[k] : PIB(&c, rep_pos, order[k]);     /* :900 */
                if (mu) mu[(size_t)k * nd + d] = c.mu[k];
                if (sig2) sig2[(size_t)k * nd + d] = c.sig2[k];
                if (pi_end) pi_end[(size_t)k * nd + d] = prep;
                acc[k] += round5(c.mu[k]);
                acc[K + k] += round5(c.sig2[k]);
                acc[2 * K + k] += round5(prep);
            }
            for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
                arow[i * K + j] = c.A[i][j];
                if (A) A[((size_t)j * K + i) * nd + d] = c.A[i][j];
                acc[3 * K + j * K + i] += round5(c.A[i][j]);
            }
            for (int h = 0; h < H; ++h) {                                             /* :860-862, :905-910 */
                double f;
                if ((blend_mask >> h) & 1) {                                          /* forecastsignal (:670-681) */
                    const double tau = 1.0 / sigma_signal, a = tau / (1.0 + tau);
                    f = 0.0;
                    for (int k = 0; k < K; ++k) f += pe[k] * (a * Yfake[T - 1] + (1.0 - a) * c.mu[k]);
                } else {
                    f = hmco_forecast(K, c.mu, arow, pe, horizons[h]);
                }
                double e = f - (yreal ? yreal[h] : NAN);
                if (fcast) { fcast[(size_t)(2 * h) * nd + d] = f; fcast[(size_t)(2 * h + 1) * nd + d] = e; }
                acc[3 * K + K * K + 2 * h] += round5(f);
                acc[3 * K + K * K + 2 * h + 1] += round5(e);
            }
            if (pi_smooth)
                for (int k = 0; k < K; ++k) for (int t = 0; t < T; ++t)
                    pi_smooth[((size_t)k * T + t) * nd + d] = PIB(&c, t, order[k]);
            if (pi_filter_mean)
                for (int t = 0; t < T; ++t) for (int k = 0; k < K; ++k)
                    pi_filter_mean[(size_t)t * K + k] += PIF(&c, t, order[k]);
        }
    }
    if (summary) for (int i = 0; i < NS; ++i) summary[i] = nd > 0 ? acc[i] / nd : NAN;
    if (pi_filter_mean && nd > 0) for (int i = 0; i < T * K; ++i) pi_filter_mean[i] /= nd;
    if (x_final) for (int t = 0; t < T; ++t) x_final[t] = c.X[t];
    if (pif_final) memcpy(pif_final, c.pif, sizeof(double) * (size_t)T * K);
    if (status) *status = c.status;
    free(c.pif); free(c.pib); free(c.Pf); free(c.Pb); free(c.X); free(c.obs_index); free(Yfake);
    return 0;
}

int hmco_estimate_window(const double *Y, int T, int K, int burnin, int nrun,
                         const int *horizons, int H, const double *yreal,
                         uint64_t seed, uint32_t window_id, int flags, const int *x_init,
                         double *mu, double *sig2, double *A, double *pi_end, double *fcast,
                         double *pi_smooth, double *summary,
                         int *x_final, double *pif_final, int *status)
{
    return hmco_estimate_window_ex(Y, T, K, burnin, nrun, horizons, H, yreal, seed, window_id, flags, x_init,
                                   T, T, 1.0, 1.0, 1.0, 1, 0.0, 0, 0, -1, 0,
                                   mu, sig2, A, pi_end, fcast, pi_smooth, summary, NULL, x_final, pif_final, NULL, status);
}

/* Batched form over W windows (window-major Y panel, ld = ldY), one window per
 * OpenMP thread.  Same layouts as include/hmcg.h (window index slowest).
 * Used as the checker for the HIP path and as bench.py's cpu_baseline. */
int hmco_estimate_batch(const double *Y, int ldY, const int *T, int W, int K, int burnin, int nrun,
                        const int *horizons, int H, const double *yreal, uint64_t seed,
                        uint32_t window_base, int flags, int nthreads,
                        double *mu, double *sig2, double *A, double *pi_end, double *fcast,
                        double *summary, int *x_final, double *pif_final, int *status)
{
    int rc = 0;
    const size_t NS = (size_t)(3 * K + K * K + 2 * H);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int w = 0; w < W; ++w) {
        size_t kn = (size_t)K * nrun;
        int r = hmco_estimate_window(Y + (size_t)w * ldY, T[w], K, burnin, nrun, horizons, H,
                                     yreal ? yreal + (size_t)w * H : NULL, seed, window_base + (uint32_t)w, flags, NULL,
                                     mu ? mu + w * kn : NULL, sig2 ? sig2 + w * kn : NULL,
                                     A ? A + w * kn * K : NULL, pi_end ? pi_end + w * kn : NULL,
                                     fcast ? fcast + (size_t)w * 2 * H * nrun : NULL, NULL,
                                     summary ? summary + w * NS : NULL,
                                     x_final ? x_final + (size_t)w * ldY : NULL,
                                     pif_final ? pif_final + (size_t)w * ldY * K : NULL,
                                     status ? status + w : NULL);
        if (r) rc = r;
    }
    return rc;
}

/* runaggregate over a signal run (src/Hmc.jl:1053-1075: group the 5-digit-rounded per-draw rows by (date, signalid),
 * mean): from the Julia-layout draw arrays of hmco_estimate_window_ex (nd = n_samples * nrun, sample-major) to
 * out[n_samples][3K + K^2 + 2H], columns mu | sig2 | pi_end | A(:) column-major | forecasts, draws added in order. */
void hmco_sample_summary(const double *mu, const double *sig2, const double *A, const double *pi_end, const double *fcast,
                         int K, int H, int n_samples, int nrun, double *out)
{
    const int NS = 3 * K + K * K + 2 * H;
    const size_t nd = (size_t)n_samples * (size_t)nrun;
    for (int smp = 0; smp < n_samples; ++smp)
        for (int c = 0; c < NS; ++c) {
            const double *col = c < K ? mu + (size_t)c * nd
                              : c < 2 * K ? sig2 + (size_t)(c - K) * nd
                              : c < 3 * K ? pi_end + (size_t)(c - 2 * K) * nd
                              : c < 3 * K + K * K ? A + (size_t)(c - 3 * K) * nd
                              : fcast + (size_t)(c - 3 * K - K * K) * nd;
            double acc = 0.0;
            for (int d = 0; d < nrun; ++d) acc += round5(col[(size_t)smp * nrun + d]);
            out[(size_t)smp * NS + c] = nrun > 0 ? acc / nrun : NAN;
        }
}

/* ---- single-kernel entry points for teacher-forced parity tests ---------- */

/* forward filter only: given theta, returns unsorted pif [T][K]. */
int hmco_forward_filter(const double *Y, int T, int K, const double *mu, const double *sig2,
                        const double *rho, const double *A_rowmajor, double *pif_out, double *Pf_out)
{
    chain_t c;
    memset(&c, 0, sizeof c);
    c.K = K; c.T = T; c.Y = Y;
    c.pif = pif_out;
    c.Pf = Pf_out ? Pf_out : (double *)malloc(sizeof(double) * (size_t)T * K * K);
    for (int i = 0; i < K; ++i) {
        c.mu[i] = mu[i]; c.sig2[i] = sig2[i]; c.rho[i] = rho[i];
        for (int j = 0; j < K; ++j) c.A[i][j] = A_rowmajor[i * K + j];
    }
    forward_update(&c);
    if (!Pf_out) free(c.Pf);
    return c.status;
}

/* backward smoother only (needs Pf from hmco_forward_filter). */
void hmco_backward_smoother(int T, int K, const double *pif, const double *Pf, double *pib_out)
{
    chain_t c;
    memset(&c, 0, sizeof c);
    c.K = K; c.T = T;
    c.pif = (double *)pif; c.Pf = (double *)Pf; c.pib = pib_out;
    c.Pb = (double *)malloc(sizeof(double) * (size_t)T * K * K);
    backward_update(&c);
    free(c.Pb);
}

/* raw RNG access for known-answer tests */
void hmco_philox(uint32_t ctr[4], uint32_t k0, uint32_t k1) { philox4x32_10(ctr, k0, k1); }
double hmco_gamma(uint64_t seed, uint32_t window, uint32_t sweep, uint32_t site, uint32_t elem, double shape)
{
    rng_t g = { seed, window, sweep };
    int st = 0;
    return rng_gamma(&g, site, elem, shape, &st);
}
double hmco_normal(uint64_t seed, uint32_t window, uint32_t sweep, uint32_t site, uint32_t elem)
{
    rng_t g = { seed, window, sweep };
    return rng_normal(&g, site, elem, 0);
}
double hmco_uniform_x(uint64_t seed, uint32_t window, uint32_t sweep, uint32_t t)
{
    rng_t g = { seed, window, sweep };
    return rng_uniform_x(&g, t);
}
int hmco_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
