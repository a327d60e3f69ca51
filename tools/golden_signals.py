"""The signal Monte-Carlo path against every row of the reference's committed
data/output/signals_official_noise_{0.1,0.3,0.6}_allsignal/forecasts_dispersion.csv (tests/golden/signals_noise_*):
all end dates in one GPU call per noise level, 100 noise samples each.  Exploratory twin of tests/test_gpu_golden.py."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_inflation():
    rows = list(csv.DictReader(open(os.path.join(GOLDEN, "inflation.csv"))))
    return np.array([np.float32(r["offic_inf"]) for r in rows]).astype(np.float64), [r["date"] for r in rows]


def load_dispersion(noise):
    """date -> list of fixture rows (a date can occur more than once: upstream re-ran some dates)"""
    out = {}
    for r in csv.DictReader(open(os.path.join(GOLDEN, "signals_noise_%s_allsignal_forecasts_dispersion.csv" % noise))):
        out.setdefault(r["date"], []).append({k: float(v) for k, v in r.items() if k != "date"})
    return out


def signal_run(noise, ns=100, burnin=1000, nrun=2000):
    y, dates = load_inflation()
    fx = load_dispersion(noise)
    use = [d for d in fx if len(fx[d]) == 1 and fx[d][0]["signalid_mean"] == 50.5]       # plain 100-sample rows
    ends = [dates.index(d) + 1 for d in use]
    W, ld, K = len(ends), max(ends), 3
    Y = np.zeros((W, ld)); Tw = np.array(ends, dtype=np.int32)
    yreal = np.zeros((W, 1)); ssig = np.zeros(W)
    sig = np.zeros((W, 2), dtype=np.int32); save = np.zeros((W, 2), dtype=np.int32)
    for i, (d, e) in enumerate(zip(use, ends)):
        Y[i, :e] = y[:e]
        yreal[i, 0] = y[e + 11]
        ssig[i] = 0.5 * (fx[d][0]["signal_1_std"] + fx[d][0]["signal_2_std"])
        sig[i] = (0, e); save[i] = (e - 2, e)
    t0 = time.perf_counter()
    r = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), yreal, want_draws=("fcast",), sig_range=sig, save_range=save,
                                 sigma_signal=ssig, kappa=float(noise), n_samples=ns, alpha=2.0, nu=2.0)
    wall = time.perf_counter() - t0
    f = r["fcast"][:, 0, :].reshape(W, ns, nrun).mean(axis=2)           # per-sample mean forecast = one row of forecasts_summary.csv
    e_ = r["fcast"][:, 1, :].reshape(W, ns, nrun).mean(axis=2)
    sv = r["sigvals"]                                                    # (W, ns, 2)
    return dict(dates=use, f_mean=f.mean(axis=1), f_std=f.std(axis=1, ddof=1), e_mean=e_.mean(axis=1), sv_mean=sv.mean(axis=1),
                sv_std=sv.std(axis=1, ddof=1), ssig=ssig, status=r["status"], wall=wall, kernel_ms=r["kernel_ms"], fx=fx, ns=ns,
                skipped=[d for d in fx if d not in use])


def compare(run):
    fx, ns = run["fx"], run["ns"]
    ref = lambda k: np.array([fx[d][0][k] for d in run["dates"]])
    se = np.hypot(run["f_std"], ref("forecast_12_std")) / np.sqrt(ns)
    z_mean = (run["f_mean"] - ref("forecast_12_mean")) / se
    z_err = (run["e_mean"] - ref("forecast_error_12_mean")) / se
    ratio = run["f_std"] / ref("forecast_12_std")
    zs1 = (run["sv_mean"][:, 0] - ref("signal_1_mean")) / (np.hypot(run["sv_std"][:, 0], ref("signal_1_std")) / np.sqrt(ns))
    zs2 = (run["sv_mean"][:, 1] - ref("signal_2_mean")) / (np.hypot(run["sv_std"][:, 1], ref("signal_2_std")) / np.sqrt(ns))
    return dict(z_mean=z_mean, z_err=z_err, ratio=ratio, zs1=zs1, zs2=zs2, diff=run["f_mean"] - ref("forecast_12_mean"))


if __name__ == "__main__":
    for noise in (sys.argv[1:] or ["0.1", "0.3", "0.6"]):
        run = signal_run(noise)
        c = compare(run)
        q = lambda v: np.round(np.quantile(v, [0.01, 0.25, 0.5, 0.75, 0.99]), 3)
        print("noise %s: %d dates (%d fixture rows skipped: %s), %.1f s wall, kernel %.1f s, flagged %d" % (
            noise, len(run["dates"]), len(run["skipped"]), run["skipped"][:4], run["wall"], run["kernel_ms"] / 1e3, int((run["status"] != 0).sum())))
        print("   z(mean forecast) quantiles", q(c["z_mean"]), "max|z| %.2f at %s" % (np.abs(c["z_mean"]).max(), run["dates"][int(np.abs(c["z_mean"]).argmax())]),
              " mean z %.3f  rms z %.3f" % (c["z_mean"].mean(), np.sqrt((c["z_mean"] ** 2).mean())))
        print("   max |diff| %.4f; std ratio quantiles" % np.abs(c["diff"]).max(), q(c["ratio"]), "geo-mean %.3f" % np.exp(np.log(c["ratio"]).mean()))
        print("   z(signal means) rms %.3f %.3f" % (np.sqrt((c["zs1"] ** 2).mean()), np.sqrt((c["zs2"] ** 2).mean())))
