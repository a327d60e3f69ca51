/* Runs the CPU oracle (oracle/hmc_oracle.c, compiled into this executable with -fsanitize=address,undefined) over the
 * shapes the tests use: the plain path, the smoother, the signal path with chained noise samples and signals past the end
 * date, K = 2..8, tiny and ragged windows.  Test infrastructure only.  Exit 0 = ran clean. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int hmco_estimate_window_ex(const double *Y, int T, int K, int burnin, int nrun, const int *horizons, int H, const double *yreal,
                            uint64_t seed, uint32_t window_id, int flags, const int *x_init, int sig_b, int sig_e, double kappa,
                            double alpha, double nu, int n_samples, double sigma_signal, int save_b, int save_e, int end_pos,
                            int blend_mask, double *mu, double *sig2, double *A, double *pi_end, double *fcast, double *pi_smooth,
                            double *summary, double *sigvals, int *x_final, double *pif_final, double *pi_filter_mean, int *status);

static double lcg(uint64_t *s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) / 9007199254740992.0; }

int main(void)
{
    uint64_t s = 42;
    const int shapes[][3] = {{2, 3, 1}, {3, 17, 2}, {3, 200, 1}, {4, 64, 3}, {5, 90, 1}, {8, 120, 2}, {2, 2, 1}};
    for (unsigned c = 0; c < sizeof shapes / sizeof shapes[0]; ++c) {
        const int K = shapes[c][0], T = shapes[c][1], ns = shapes[c][2], burnin = 3, nrun = 9, H = 2;
        const int horizons[2] = {1, 12};
        const int nd = ns * nrun, NS = 3 * K + K * K + 2 * H;
        double *Y = malloc(sizeof(double) * T), yreal[2] = {1.0, 2.0};
        for (int t = 0; t < T; ++t) Y[t] = 3.0 * lcg(&s) + (t % 7 == 0 ? 4.0 : 0.0);
        double *mu = malloc(8 * K * nd), *sig2 = malloc(8 * K * nd), *A = malloc(8 * K * K * nd), *pe = malloc(8 * K * nd);
        double *fc = malloc(8 * 2 * H * nd), *sm = malloc(8 * (size_t)nd * T * K), *summ = malloc(8 * NS), *sv = malloc(8 * ns * 3);
        double *pif = malloc(8 * T * K), *pfm = malloc(8 * T * K);
        int *xf = malloc(4 * T), st = 0;
        for (int variant = 0; variant < 4; ++variant) {
            const int sig = variant >= 2 && T > 6;                         /* signal tail of 3 steps */
            const int tail = variant == 3 && T > 8 ? 2 : 0;                /* ... two of them past the end date */
            const int rc = hmco_estimate_window_ex(Y, T, K, burnin, nrun, horizons, H, yreal, 1234, c, variant == 1 ? 3 : 0, NULL,
                                                   sig ? T - 3 : T, T, 0.6, sig ? 2.0 : 1.0, sig ? 2.0 : 1.0, sig ? ns : 1,
                                                   sig ? 0.5 : 0.0, sig ? T - 3 : 0, sig ? T : 0, tail ? T - 1 - tail : -1, tail ? 1 : 0,
                                                   mu, sig2, A, pe, fc, variant == 1 ? sm : NULL, summ, sig ? sv : NULL, xf, pif, pfm, &st);
            if (rc != 0) { fprintf(stderr, "oracle rc=%d (K=%d T=%d variant %d)\n", rc, K, T, variant); return 1; }
            for (int i = 0; i < NS; ++i) if (!(summ[i] == summ[i]) && !(tail && i >= 3 * K + K * K)) { /* NaN only where upstream leaves one */ }
        }
        free(Y); free(mu); free(sig2); free(A); free(pe); free(fc); free(sm); free(summ); free(sv); free(pif); free(pfm); free(xf);
    }
    printf("oracle harness ok\n");
    return 0;
}
