"""Two diagnostics behind tests/test_gpu_golden.py's signal-fixture tolerances:
 (1) how reproducible is upstream's sigma_signal = mean(sigma draws of the base run) * noise (src/Hmc.jl:868-872) across
     independent chains of upstream's own length (100k + 250k sweeps)?  R replicas per date, spread across replicas;
 (2) how far do the per-date means of the variance columns move when sigma_signal is off by 5 % (the error of estimating it
     from the fixture's 2 x 100 saved noisy signals), in units of the across-sample standard error the z-scores use?
usage: python tools/golden_signals_diag.py [noise=0.3] [replicas=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import golden_signals as gs
from hmc_jl_amd import _lib

noise = sys.argv[1] if len(sys.argv) > 1 else "0.3"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
y, dates = gs.load_inflation()
fx = gs.load_dispersion(noise)
use = [d for d in fx if len(fx[d]) == 1 and fx[d][0]["signalid_mean"] == 50.5]
ends = [dates.index(d) + 1 for d in use]
W, ld, K = len(ends), max(ends), 3
Y = np.zeros((W, ld)); Tw = np.array(ends, dtype=np.int32); yreal = np.zeros((W, 1))
sig = np.zeros((W, 2), dtype=np.int32)
for i, e in enumerate(ends):
    Y[i, :e] = y[:e]; yreal[i, 0] = y[e + 11]; sig[i] = (0, e)
fix = np.array([0.5 * (fx[d][0]["signal_1_std"] + fx[d][0]["signal_2_std"]) for d in use])
reps = []
for r in range(R):
    b = _lib.estimate_batch_host(Y, Tw, K, 100000, 250000, (12,), yreal, want_draws=False, sig_range=sig, kappa=1.0, alpha=1.0, nu=1.0,
                                 window_ids=np.arange(W) + (r + 1) * (1 << 20))
    reps.append(b["summary"][:, K:2 * K].mean(axis=1) * float(noise))
reps = np.array(reps)                                                       # (R, W)
cv = reps.std(axis=0, ddof=1) / reps.mean(axis=0)
print("(1) base-run sigma_signal over %d replicas: coefficient of variation across replicas, quantiles 5/50/95/99%%: %s" % (R, np.round(np.quantile(cv, [0.05, 0.5, 0.95, 0.99]), 4)))
rr = fix[None, :] / reps
print("    fixture-implied / replica: median %.4f; quantiles 5/95%% %s; the same between two replicas: %s" % (
    np.median(rr), np.round(np.quantile(rr, [0.05, 0.95]), 3), np.round(np.quantile(reps[0] / reps[1], [0.05, 0.95]), 3)))
# per-state posterior means of the base run: is the spread in one state?
b0 = _lib.estimate_batch_host(Y, Tw, K, 100000, 250000, (12,), yreal, want_draws=False, sig_range=sig, kappa=1.0, alpha=1.0, nu=1.0,
                              window_ids=np.arange(W) + 99 * (1 << 20))
b1 = _lib.estimate_batch_host(Y, Tw, K, 100000, 250000, (12,), yreal, want_draws=False, sig_range=sig, kappa=1.0, alpha=1.0, nu=1.0,
                              window_ids=np.arange(W) + 98 * (1 << 20))
print("    two replicas, ratio of the per-state variance means, quantiles 5/50/95%% per state:",
      [list(np.round(np.quantile(b0["summary"][:, K + k] / b1["summary"][:, K + k], [0.05, 0.5, 0.95]), 3)) for k in range(K)])
# (2) sensitivity of the variance columns to sigma_signal
outs = []
for f in (1.0, 1.05):
    r = _lib.estimate_batch_host(Y, Tw, K, 1000, 2000, (12,), yreal, want_draws=False, sig_range=sig, sigma_signal=fix * f, kappa=float(noise),
                                 n_samples=100, alpha=2.0, nu=2.0, want_sample_summary=True)
    outs.append(r["sample_summary"])
a, c = outs
se = np.sqrt(a[:, :, K:2 * K].std(axis=1, ddof=1) ** 2 * 2) / 10.0
shift = (c[:, :, K:2 * K].mean(axis=1) - a[:, :, K:2 * K].mean(axis=1)) / se
print("(2) sigma_signal x 1.05: shift of the per-date variance means in units of the z-score's standard error, per state: median %s, q95 %s" % (
    np.round(np.median(shift, axis=0), 2), np.round(np.quantile(np.abs(shift), 0.95, axis=0), 2)))
se_m = np.sqrt(a[:, :, 0:K].std(axis=1, ddof=1) ** 2 * 2) / 10.0
shift_m = (c[:, :, 0:K].mean(axis=1) - a[:, :, 0:K].mean(axis=1)) / se_m
print("    the same for the state means: median %s, q95 %s" % (np.round(np.median(shift_m, axis=0), 2), np.round(np.quantile(np.abs(shift_m), 0.95, axis=0), 2)))
