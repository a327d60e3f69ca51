/*
 * hmcg.h -- C ABI of libhmcgibbs.so: batched Gibbs-sampled Gaussian-HMM estimation
 * on AMD MI355X (gfx950).  This is the drop-in boundary for the data-parallel hot
 * path of joe5saia/Hmc.jl.
 *
 * The reference has no FFI of its own: its boundary is the Julia function surface
 * of `module Hmc`.  What each entry point replaces (reference file:line):
 *
 *   hmcg_estimate_batch[_device]   W x { Hmc.estimatemodel(opt)       src/Hmc.jl:850-865
 *                                        -> makeParams                src/Hmc.jl:161-195
 *                                        -> HyperParams(Y,D)          src/Hmc.jl:132-142
 *                                        -> gibbssample!/gibbssweep!  src/Hmc.jl:486-562
 *                                           (update_mu_sigma :231-336, update_beta :338-348,
 *                                            update_rho :350-356, update_A :358-369,
 *                                            forwardupdate_P :371-440, backwardupdate_P :442-457
 *                                            [only pib[end,:] is produced], sort :501-513,
 *                                            update_X :459-484)
 *                                        -> forecast per draw         src/Hmc.jl:658-667, 860-862 }
 *                                  and the per-window means that runaggregate
 *                                  (src/Hmc.jl:1025-1078) takes over the 5-digit-rounded
 *                                  per-draw CSV rows of basicsave (src/Hmc.jl:707-722).
 *   W windows in one call          replace the SLURM array fan-out, one process per
 *                                  window (slurmscripts/base_estimation.sh:5,17).
 *
 * Conventions: plain C structs, fixed-width ints, no C++ types or exceptions cross
 * the boundary.  Return 0 on success, negative on API misuse (HMCG_E_*), positive
 * = hipError_t of a failed HIP call.  Per-window numerical trouble is reported in
 * status[w] (HMCG_ST_* bits), never as a call failure.  The caller allocates and
 * owns every buffer; the library keeps no caller pointer after return.  The host
 * entries are blocking.  The library keeps one lazily created context per device (streams, events,
 * grow-only device and pinned-host workspaces; hmcg_shutdown releases them); calls on the SAME device are
 * serialised by that context's mutex, calls on different devices run concurrently.  There is NO CPU
 * fallback: without a usable GPU every compute entry fails with HMCG_E_NODEVICE.
 *
 * Array layouts are the reference's Julia (column-major) layouts with the window
 * index appended as the slowest dimension:
 *   mu, sig2, pi_end : Julia (nrun, K, W)      elem (d,k,w)   at d + nrun*(k + K*w)
 *   A                : Julia (nrun, K, K, W)   elem (d,i,j,w) at d + nrun*(i + K*(j + K*w))
 *   fcast            : Julia (nrun, 2H, W)     cols forecast_h, forecast_error_h per horizon
 *   summary          : Julia (3K+K*K+2H, W)    means over draws of round(x; digits=5), rows
 *                      mu(K) | sig2(K) | pi_end(K) | A(:) column-major (K*K) | fcast(2H)
 *   Y                : window-major panel, row w holds Y_w[0..T[w]-1], leading dim ldY
 *   yreal            : (H, W): realised value rawdata[endIndex+h] per horizon (NaN if unknown)
 * "sig2" holds the VARIANCE draws (the reference's `sigma` arrays hold variances).
 * States are labelled 0..K-1 here (1..K in Julia).
 */
#ifndef HMCG_H
#define HMCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMCG_VERSION 107
#define HMCG_MAXH 8
#define HMCG_MAXTAIL 256        /* most signal steps past the end date (sigLen, src/Hmc.jl:888) */
#define HMCG_MAXK 8
#define HMCG_MAXDEV 16           /* devices one process may drive */

/* API-misuse return codes */
#define HMCG_E_BADARG   (-1)
#define HMCG_E_NODEVICE (-2)
#define HMCG_E_UNSUPPORTED (-3) /* (K, max_T) combination has no compiled kernel */
#define HMCG_E_NOMEM    (-4)

/* status[w] bits */
#define HMCG_ST_BAD_INVGAMMA   1 /* a<=0 or b<=0 in the InvGamma update: old variance kept (src/Hmc.jl:319-329) */
#define HMCG_ST_EMIS_UNDERFLOW 2 /* all K emission pdfs underflowed at some step: that observation was treated as missing (f=1); or a zero
                                    normaliser was replaced by the uniform law (reference would produce NaN and throw, src/Hmc.jl:435) */
#define HMCG_ST_NONFINITE      4 /* non-finite observation: window skipped, outputs untouched */
#define HMCG_ST_GAMMA_CAP      8 /* gamma rejection sampler hit its attempt cap */
#define HMCG_ST_BAD_T         16 /* T[w] < 2, T[w] > ldY, T[w] beyond what max_T was sized for, or end_pos[w] outside
                                      [T-1-HMCG_MAXTAIL, T-1]: window skipped */
#define HMCG_ST_BAD_RANGE     32 /* signal path: sig_range[w] / save_range[w] outside [0, T[w]], a non-empty sig_range not ending
                                      at T[w], or a save_range longer than nsave_ld: window skipped */

/* flags */
#define HMCG_FLAG_RESUME 1  /* chain state (extras.xstate) is loaded instead of the makeParams init; sweep numbering continues at sweep_base */

typedef struct hmcg_config {
    int32_t struct_size;     /* = sizeof(hmcg_config), for ABI evolution */
    int32_t W;               /* windows in this call */
    int32_t K;               /* states, 2..HMCG_MAXK   (estopt.D, src/Hmc.jl:32) */
    int32_t ldY;             /* leading dimension of the Y panel, >= max_T */
    int32_t max_T;           /* max over w of T[w] (0: use ldY) */
    int32_t burnin;          /* discarded sweeps (estopt.burnin, src/Hmc.jl:33) */
    int32_t nrun;            /* kept sweeps      (estopt.Nrun,   src/Hmc.jl:34) */
    int32_t H;               /* number of forecast horizons, 0..HMCG_MAXH */
    int32_t horizons[HMCG_MAXH]; /* estopt.horizons, src/Hmc.jl:31 */
    uint64_t seed;           /* estopt.seed (src/Hmc.jl:41,59); Philox key */
    uint32_t window_base;    /* RNG stream id of window 0 of this call; window w uses window_base+w,
                                so a sharded run reproduces the unsharded one */
    int32_t device;          /* HIP device ordinal */
    int32_t flags;           /* HMCG_FLAG_* */
    int32_t threads_per_window; /* 0 = auto (256); 128/256/512 */
    int32_t sweep_base;      /* global index of the first sweep of this call (0 unless resuming) */
    int32_t sweep_count;     /* sweeps to run in this call; 0 = all remaining (burnin+nrun-sweep_base).  A call that
                                stops short writes extras.xstate/sumacc so a later RESUME call can continue */
    double alpha;            /* InvGamma prior sample size, 0 -> 1.0 (HyperParams(Y,D), src/Hmc.jl:137; 2.0 for HyperParams(opt) :154) */
    double nu;               /* Normal prior sample size,   0 -> 1.0 (src/Hmc.jl:140; 2.0 for HyperParams(opt) :157) */
    /* signal Monte-Carlo path (estimatesignals!, src/Hmc.jl:868-914); leave zero for estimatemodel */
    double kappa;            /* hp.kappa: relative noise of a signal observation (opt.noise :158; 1.0 in the base run :132) */
    int32_t n_samples;       /* opt.noiseSamples: consecutive chains of burnin+nrun sweeps, each on fresh noise, the chain
                                state carried over (:889-895); 0 or 1 = a single chain.  Outputs then hold n_samples*nrun
                                draws (sample-major), burnin/nrun being opt.signalburnin/opt.signalNrun */
    int32_t blend_mask;      /* signal path: bit k set = horizon k equals sigLen and is reported through forecastsignal
                                (src/Hmc.jl:670-681, :908-909): sum_i pi_last[i] (a Yfake[T-1] + (1-a) mu[i]), a = tau/(1+tau),
                                tau = 1/sigma_signal[w]; horizons[k] is then ignored.  0 otherwise */
    int32_t min_T;           /* device entry, optional hint: min over w of T[w] (0: unknown).  With it a ragged batch -- the
                                reference's production run is 460 expanding windows of 120..579 months, code/run_hmm.jl:79-109 --
                                is dispatched by length: every window runs on the steps-per-thread variant its OWN length
                                selects (one launch per length class, side by side), and its result is that of a call with
                                this window alone.  Without it one launch is sized for max_T.  The host entries read T
                                themselves and ignore the field */
    int32_t reserved3;
} hmcg_config;

/* Optional debug / teacher-forcing / checkpoint buffers (all may be NULL).  Pointer
 * residency follows the entry point (host pointers for hmcg_estimate_batch, device
 * pointers for hmcg_estimate_batch_device). */
typedef struct hmcg_extras {
    int32_t struct_size;
    int32_t reserved;
    const int32_t* x_init;   /* [W][ldY] initial states (0-based) instead of makeParams' argmax-pdf init */
    int32_t* x_final;        /* [W][ldY] states after the last sweep */
    double* pif_final;       /* [W][ldY][K] UNSORTED filtered probabilities pif[t,:] of the last sweep */
    uint8_t* xstate;         /* [W][ldY] chain state for HMCG_FLAG_RESUME: read at start when the flag is
                                set, written at the end whenever non-NULL (checkpoint) */
    double* sumacc;          /* [W][3K+K*K+2H+K] checkpoint block: the running sums behind `summary`, then the K pivots of
                                the one-pass sufficient statistics (so that a resumed chain is bit-identical) */
    const uint32_t* window_ids; /* [W] explicit RNG stream ids (NULL: window_base + w); lets a sharded /
                                   load-balanced run reproduce the unsharded one window for window */
    /* signal path (all NULL for estimatemodel).  Positions are 0-based and window-relative. */
    const int32_t* sig_range;   /* [W][2] signal positions [begin, end): opt.signalRange; end must equal T[w] */
    const int32_t* save_range;  /* [W][2] positions reported in sigvals: opt.signalSave */
    const double* sigma_signal; /* [W] opt.sigma_signal: sd of the N(0,1) noise added to the signal positions (:892) */
    double* sigvals;            /* [W][n_samples][nsave_ld] Yfake[signalSave] of every noise sample (:904) */
    int32_t nsave_ld;
    int32_t reserved2;
    const int32_t* end_pos;     /* [W] signal path, optional: position (opt.endIndex - 1) whose SMOOTHED probabilities are
                                   reported in pi_end when the window carries sigLen = T-1-end_pos signal steps past the end
                                   date (samples.pib[:, opt.endIndex, :], src/Hmc.jl:888,900); 0 <= sigLen <= HMCG_MAXTAIL.
                                   NULL: the last step.  Forecasts always start from the last step (:907,909): pass
                                   horizons[k] = h - sigLen, or set blend_mask for h == sigLen */
    double* pi_smooth_mean;     /* [W][ldY][K] optional: mean over the kept draws of the SMOOTHED probabilities
                                   P(X_t | Y_1:T, theta) in sorted labels = the draw-average of the reference's
                                   samples.pib[:, t, :] (backwardupdate_P!, src/Hmc.jl:442-457, sorted :513).  Every K and every
                                   supported T (the LDS-resident kernel needs extras.pif_final on the device entry); about
                                   half as many draws per second as without.  NULL: only pib[end,:] is produced */
    double* pi_filter_mean;     /* [W][ldY][K] optional: the same draw average for the FILTERED probabilities pif[t,:] in
                                   sorted labels (what the reference's older API returned as "pib" and averaged per date in
                                   data/output/official_insample/forecats_insample.csv, columns s1..s3).  Runs on the same
                                   kernel variants as pi_smooth_mean */
    double* corr;               /* [W][NC][NC] optional, NC = HMCG_CORR_COLUMNS(K): Pearson correlations between the per-draw
                                   outputs of each window over its kept draws, columns mu_1..K | sigma_1..K | pi_1..K |
                                   vec(A) column-major | forecast of horizons[0] -- the matrix calccorr (src/Hmc.jl:1094-1163)
                                   computes per end date from the 5-digit-rounded per-draw CSV files, here taken from the
                                   rounded draws while they are in HBM (second moments about the first draw, accumulated in
                                   draw order, chunk by chunk).  Needs all five per-draw outputs (mu, sig2, A, pi_end, fcast
                                   with H >= 1; the host entries accept NULL for them and then keep the draws on the device),
                                   n_samples <= 1 and the whole run in one call (no sweep_base / sweep_count / RESUME).  A
                                   constant column gives NaN, as Statistics.cor does */
    double* pi_smooth_draws;    /* [W][K][ldY][nd] = Julia (Nrun, N, D, W), optional: EVERY kept draw's smoothed probabilities in sorted
                                   labels -- the reference's samples.pib[Nrun, N, D] itself (gibbssample!, src/Hmc.jl:552,558), of
                                   which its live outputs only read [:, end, :] (= pi_end).  8 K T bytes per draw and window (24 KB at
                                   K = 3, T = 1000): the host entries stream it chunk by chunk like the other per-draw outputs.  Runs the
                                   smoothing variants, as pi_smooth_mean does */
    double* sample_summary;     /* [W][n_samples][3K+K*K+2H] optional, signal path: for every noise sample the mean over its
                                   nrun kept draws of the 5-digit-rounded outputs, columns as in `summary` -- one row of the
                                   `*_summary.csv` files upstream's runaggregate (src/Hmc.jl:1025-1057, grouped by date and
                                   signalid) makes out of a signal run's per-draw files, taken on the device instead (the
                                   n_samples * nrun draws need not leave it).  While a sample is incomplete (a call that
                                   stops inside it: sweep_count) its row holds the raw running sums; a RESUME call reads
                                   them back from the same buffer */
} hmcg_extras;
#define HMCG_CORR_COLUMNS(K) (3 * (K) + (K) * (K) + 1)

typedef struct hmcg_timing {
    double kernel_ms;        /* HIP-event time of the sweep kernel(s) on the launch stream (summed over the chunks of a call) */
    int32_t launches;        /* kernel launches of this call: the host entries run a long chain in chunks whose per-draw
                                outputs travel to the caller while the next chunk samples */
    int32_t threads_per_window;
    int32_t steps_per_thread;
    int32_t lds_bytes;
    int32_t helper_waves;    /* extra 64-thread waves per window that carry the draw-phase side jobs (0 or 4) */
    int32_t device;          /* HIP device ordinal this record describes */
    double call_ms;          /* host entries: wall time of the whole call on this device -- staging, H2D, kernels, D2H and the
                                scatter into the caller's arrays (0 for the device entry) */
    int32_t windows;         /* windows this device ran */
    int32_t occupancy;       /* register-resident kernels: 1 = the whole register file per window, 2 = capped so that two windows
                                share a CU (the OCC template argument of the kernel that ran); 0 for the LDS-resident kernel */
    int32_t buckets;         /* length classes the call was dispatched in (1: one launch for every window); threads_per_window,
                                steps_per_thread, helper_waves, occupancy and lds_bytes then describe the LONGEST class */
    int32_t streaming;       /* 1: the LDS-resident kernel ran in its HBM-streaming form (window longer than a CU's LDS holds) */
} hmcg_timing;

int hmcg_version(void);
int hmcg_device_count(void);          /* number of usable HIP devices (0 if none) */
const char* hmcg_last_error(void);    /* thread-local, never NULL */
void hmcg_shutdown(void);             /* releases the library's streams/workspaces */

/* Host-buffer entry: stages Y/T/yreal through pinned memory to the device, runs the chain in a few chunks and
 * streams each chunk's per-draw outputs back (SDMA copy into pinned staging, then into the caller's arrays)
 * while the next chunk samples.  Any output pointer may be NULL (that output is not produced).  Outputs of
 * skipped windows (status HMCG_ST_NONFINITE / HMCG_ST_BAD_T / HMCG_ST_BAD_RANGE) are zero here; the device entry
 * leaves them untouched. */
int hmcg_estimate_batch(const hmcg_config* cfg, const double* Y, const int32_t* T, const double* yreal,
                        double* mu, double* sig2, double* A, double* pi_end, double* fcast,
                        double* summary, int32_t* status, const hmcg_extras* extras, hmcg_timing* timing);

/* Device-buffer entry: every data pointer is HBM-resident on cfg->device.  Work is
 * enqueued on `stream` (a hipStream_t; NULL = the library's own stream).  The call
 * returns after enqueueing unless `timing` is non-NULL, in which case it waits for
 * completion and fills it.  status must be zero-initialised by the caller or is
 * zeroed by the library when HMCG_FLAG_RESUME is not set. */
int hmcg_estimate_batch_device(const hmcg_config* cfg, const double* dY, const int32_t* dT, const double* dyreal,
                               double* dmu, double* dsig2, double* dA, double* dpi_end, double* dfcast,
                               double* dsummary, int32_t* dstatus, const hmcg_extras* dextras,
                               void* stream, hmcg_timing* timing);

/* Multi-device host entry: the same contract as hmcg_estimate_batch, with the W windows of the call partitioned over
 * n_devices GPUs of this node -- what the reference does with one SLURM array task per end date
 * (slurmscripts/base_estimation.sh:5,17: `--array=120-579`, one Julia process per window).  Windows are independent
 * chains, so the data path has no collective: the library sorts the windows by length, deals them to the lightest
 * device (count-balanced), runs one host thread + stream + workspace per device, keys every window's random numbers by
 * its GLOBAL id (extras.window_ids[w], or cfg->window_base + w) -- so the result is bit-identical to the
 * single-device call whatever the partition -- and each device's thread places its windows' blocks straight into the
 * caller's arrays (the gather).  cfg->device is ignored; device_ids lists distinct HIP ordinals (NULL: 0..n_devices-1).
 * timing, if not NULL, receives n_devices records.  Returns 0, an HMCG_E_* code, or the hipError_t of the first
 * device that failed (hmcg_last_error() names it). */
int hmcg_estimate_batch_multi(const hmcg_config* cfg, int32_t n_devices, const int32_t* device_ids,
                              const double* Y, const int32_t* T, const double* yreal,
                              double* mu, double* sig2, double* A, double* pi_end, double* fcast,
                              double* summary, int32_t* status, const hmcg_extras* extras, hmcg_timing* timing);

/* ---- per-draw CSV output (host code, no GPU): basicsave / saveresults, src/Hmc.jl:707-748 ------------------------------
 * The five per-window files `filtered_means_<date>.csv`, `filtered_variances_<date>.csv`, `filtered_state_probs_<date>.csv`,
 * `filtered_trans_probs_<date>.csv`, `forecasts_<date>.csv` (:741-746), written straight from the draw arrays of the
 * estimate entries above (window w's block of every array), byte for byte as upstream's CSV.jl 0.5.16 writes them: header
 * `date[,signalid],state_1..K | trans_i_j (i fastest, :727) | forecast_h,forecast_error_h [,signal_1..]`, one row per kept
 * draw, values round(x; digits=5) (:719), LF line ends, CSV.jl float text (shortest round-trip digits; integral values
 * without a fraction; |x| < 1e-4 as <integer mantissa>e-<n>).  250 000 rows x 5 files per window in production
 * (code/run_hmm.jl:103-104).  Any of mu/sig2/pi_end/A/fcast may be NULL (that file is not written).
 * dates: W strings "yyyy-mm-dd" (the end date of each window, Hmc.enddate).  sigvals (signal path, [W][n_samples][nsave_ld])
 * adds the `signalid` column (1-based sample number of the row) and the `signal_j` columns (:715-717).
 * n_threads: windows are written in parallel (0 = one thread per hardware thread).  Returns 0 or HMCG_E_BADARG (bad
 * arguments / a file could not be written). */
#define HMCG_CSV_LEGACY_TRANS_HEADER 1 /* name the transition columns trans_<j>_<i> as the committed fixtures do
                                          (code/deprecated/Hmc.jl_08072019bak:711); data order is unchanged */
int hmcg_save_results_csv(const char* dir, int32_t W, const char* const* dates, int32_t K, int32_t H, const int32_t* horizons,
                          int64_t nd, const double* mu, const double* sig2, const double* pi_end, const double* A,
                          const double* fcast, const double* sigvals, int32_t n_samples, int32_t nsave, int32_t nsave_ld,
                          int32_t flags, int32_t n_threads);
/* One table: `date,<colnames>` with n rows from a column-major block (element (d, c) at data[d + ld*c]), rounded to
 * `precision` digits (basicsave, :707-722). */
int hmcg_write_table_csv(const char* path, const char* date, int32_t ncol, const char* const* colnames, const double* data,
                         int64_t n, int64_t ld, int32_t precision);
/* CSV.jl 0.5.16 text of one Float64 into buf (>= 48 bytes, NUL-terminated); returns its length. */
int hmcg_format_float(double x, char* buf);

#ifdef __cplusplus
}
#endif
#endif /* HMCG_H */
