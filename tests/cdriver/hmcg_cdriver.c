/* Plain-C caller of the libhmcgibbs C ABI (include/hmcg.h) -- no Python, no torch in the process: the library binds
 * the system ROCm runtime, exactly as under a Julia `ccall` (the reference's callers are
 * code/run_hmm.jl:95-120).  Reads a little-endian request file written by tests/test_gpu_cdriver.py, runs
 *   mode 0  hmcg_estimate_batch                     (estimatemodel, src/Hmc.jl:850-865)
 *   mode 1  hmcg_estimate_batch on the signal path  (estimatesignals!, src/Hmc.jl:868-914)
 *   mode 2  hmcg_estimate_batch_multi               (windows partitioned over the listed devices)
 *   mode 3  timing loop over hmcg_estimate_batch with caller-owned, reused buffers (what a Julia caller does):
 *           prints the mean wall time per call; bench.py reports it as the end-to-end figure of the C ABI
 *   mode 4  hmcg_estimate_batch with extras.corr    (calccorr's matrices, src/Hmc.jl:1094-1163, taken on the device)
 * and writes the raw outputs for the pytest wrapper to compare with the oracle.
 * usage: hmcg_cdriver <request.bin> <response.bin> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hmcg.h"

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }
static int wr(FILE* f, const void* p, size_t n) { return fwrite(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s request.bin response.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hd[20];
    double par[3];
    if (rd(f, hd, sizeof hd) || rd(f, par, sizeof par)) { fprintf(stderr, "short header\n"); return 2; }
    const int32_t magic = hd[0], mode = hd[1], W = hd[2], K = hd[3], ldY = hd[4], burnin = hd[5], nrun = hd[6], H = hd[7];
    const int32_t n_samples = hd[16], nsave_ld = hd[17], n_devices = hd[18];
    if (magic != 0x484d4347) { fprintf(stderr, "bad magic\n"); return 2; }
    const size_t nd = (size_t)(n_samples > 1 ? n_samples : 1) * (size_t)nrun, NS = 3 * (size_t)K + (size_t)K * K + 2 * (size_t)H;
    double* Y = malloc(sizeof(double) * W * ldY);
    int32_t* T = malloc(sizeof(int32_t) * W);
    double* yreal = malloc(sizeof(double) * W * (H ? H : 1));
    if (rd(f, Y, sizeof(double) * W * ldY) || rd(f, T, sizeof(int32_t) * W) || rd(f, yreal, sizeof(double) * W * H)) return 2;
    int32_t *sig = NULL, *save = NULL;
    double *ssig = NULL, *sigvals = NULL;
    if (mode == 1) {
        sig = malloc(8 * W); save = malloc(8 * W); ssig = malloc(8 * W);
        sigvals = calloc((size_t)W * (n_samples > 1 ? n_samples : 1) * nsave_ld, 8);
        if (rd(f, sig, 8 * W) || rd(f, save, 8 * W) || rd(f, ssig, 8 * W)) return 2;
    }
    fclose(f);

    hmcg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.W = W; cfg.K = K; cfg.ldY = ldY; cfg.max_T = 0; cfg.burnin = burnin; cfg.nrun = nrun; cfg.H = H;
    for (int i = 0; i < 8; ++i) cfg.horizons[i] = hd[8 + i];
    cfg.seed = 1234; cfg.window_base = 0; cfg.device = 0;
    cfg.kappa = par[0]; cfg.alpha = par[1]; cfg.nu = par[2];
    cfg.n_samples = n_samples;
    hmcg_extras ex;
    memset(&ex, 0, sizeof ex);
    ex.struct_size = (int32_t)sizeof ex;
    if (mode == 1) { ex.sig_range = sig; ex.save_range = save; ex.sigma_signal = ssig; ex.sigvals = sigvals; ex.nsave_ld = nsave_ld; }
    const size_t NC = (size_t)HMCG_CORR_COLUMNS(K);
    double* corr = NULL;
    if (mode == 4) { corr = calloc((size_t)W * NC * NC, 8); ex.corr = corr; }

    double* mu = calloc(W * K * nd, 8); double* sig2 = calloc(W * K * nd, 8); double* A = calloc(W * K * K * nd, 8);
    double* pe = calloc(W * K * nd, 8); double* fc = calloc(W * 2 * (H ? H : 1) * nd, 8); double* sm = calloc(W * NS, 8);
    int32_t* st = calloc(W, 4);
    hmcg_timing tm[HMCG_MAXDEV];
    memset(tm, 0, sizeof tm);
    if (hmcg_version() != HMCG_VERSION) { fprintf(stderr, "header/library version mismatch\n"); return 3; }
    if (hmcg_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", hmcg_last_error()); return 4; }
    int rc;
    if (mode == 3) {
        const int reps = n_devices > 0 ? n_devices : 10;        /* (the header slot doubles as the repetition count) */
        for (int i = 0; i < 3; ++i) {
            rc = hmcg_estimate_batch(&cfg, Y, T, yreal, mu, sig2, A, pe, fc, sm, st, NULL, NULL);
            if (rc) { fprintf(stderr, "libhmcgibbs rc=%d: %s\n", rc, hmcg_last_error()); return 5; }
        }
        /* per-call wall times; the median is reported (a host thread that gets descheduled once costs a mean 10 %) */
        double* tc = calloc((size_t)reps, 8);
        double* mu_ref = calloc(W * K * nd, 8);
        double sum = 0.0;
        for (int i = 0; i < reps; ++i) {
            const double t0 = now_ms();
            rc = hmcg_estimate_batch(&cfg, Y, T, yreal, mu, sig2, A, pe, fc, sm, st, NULL, NULL);
            tc[i] = now_ms() - t0;
            sum += tc[i];
            /* a failing (early-returning) call must not be reported as a fast one; nor one whose draws changed */
            if (rc) { fprintf(stderr, "libhmcgibbs rc=%d in timed call %d: %s\n", rc, i, hmcg_last_error()); return 5; }
            if (i == 0) memcpy(mu_ref, mu, W * K * nd * 8);
            else if (i == reps - 1 && memcmp(mu_ref, mu, W * K * nd * 8)) { fprintf(stderr, "timed call %d: draws differ from the first\n", i); return 6; }
        }
        for (int i = 1; i < reps; ++i)                          /* insertion sort */
            for (int j = i; j > 0 && tc[j - 1] > tc[j]; --j) { const double x = tc[j]; tc[j] = tc[j - 1]; tc[j - 1] = x; }
        const double per = reps & 1 ? tc[reps / 2] : 0.5 * (tc[reps / 2 - 1] + tc[reps / 2]);
        const double mean = sum / reps, tmin = tc[0], tmax = tc[reps - 1];
        rc = hmcg_estimate_batch(&cfg, Y, T, yreal, mu, sig2, A, pe, fc, sm, st, NULL, tm);
        printf("cdriver bench: %.4f ms per call over %d calls (W=%d K=%d T<=%d draws=%d, all per-draw outputs to host); "
               "timed call: %d launches, kernels %.3f ms, call %.3f ms; median of calls, mean %.4f min %.4f max %.4f\n", per, reps, W, K, ldY,
               nrun, tm[0].launches, tm[0].kernel_ms, tm[0].call_ms, mean, tmin, tmax);
        hmcg_shutdown();
        return rc ? 5 : 0;
    }
    if (mode == 2) {
        int32_t devs[HMCG_MAXDEV];
        for (int i = 0; i < n_devices; ++i) devs[i] = i;
        rc = hmcg_estimate_batch_multi(&cfg, n_devices, devs, Y, T, yreal, mu, sig2, A, pe, fc, sm, st, &ex, tm);
    } else {
        rc = hmcg_estimate_batch(&cfg, Y, T, yreal, mu, sig2, A, pe, fc, sm, st, &ex, tm);
    }
    if (rc) { fprintf(stderr, "libhmcgibbs rc=%d: %s\n", rc, hmcg_last_error()); return 5; }
    /* a second call reuses the library's workspaces (no allocation): results must not change */
    double* mu2 = calloc(W * K * nd, 8);
    rc = mode == 2 ? hmcg_estimate_batch_multi(&cfg, n_devices, NULL, Y, T, yreal, mu2, NULL, NULL, NULL, NULL, NULL, st, &ex, NULL)
                   : hmcg_estimate_batch(&cfg, Y, T, yreal, mu2, NULL, NULL, NULL, NULL, NULL, st, &ex, NULL);
    if (rc || memcmp(mu, mu2, W * K * nd * 8)) { fprintf(stderr, "second call differs (rc=%d)\n", rc); return 6; }
    hmcg_shutdown();

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    int bad = wr(f, mu, W * K * nd * 8) || wr(f, sig2, W * K * nd * 8) || wr(f, A, W * K * K * nd * 8) || wr(f, pe, W * K * nd * 8) ||
              wr(f, fc, W * 2 * H * nd * 8) || wr(f, sm, W * NS * 8) || wr(f, st, W * 4);
    if (mode == 1) bad = bad || wr(f, sigvals, (size_t)W * (n_samples > 1 ? n_samples : 1) * nsave_ld * 8);
    if (mode == 4) bad = bad || wr(f, corr, (size_t)W * NC * NC * 8);
    fclose(f);
    printf("cdriver ok: mode %d, %d windows, %d launches, kernel %.3f ms, call %.3f ms\n", mode, W, tm[0].launches, tm[0].kernel_ms, tm[0].call_ms);
    return bad ? 7 : 0;
}
