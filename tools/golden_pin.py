"""All 460 official windows at upstream's own 100k + 250k sweeps (code/run_hmm.jl:103-104) in one GPU run, against every
row of the five committed data/output/official/*_summary.csv files (tests/golden/official_*): z-scores from batch-means
Monte-Carlo standard errors.  Exploratory twin of tests/test_gpu_golden.py (prints the distribution)."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FILES = ("filtered_means", "filtered_variances", "filtered_state_probs", "filtered_trans_probs", "forecasts")


def official_run(burnin=100000, nrun=250000, nbatch=25, ends=range(120, 580)):
    rows = list(csv.DictReader(open(os.path.join(GOLDEN, "inflation.csv"))))
    y = np.array([np.float32(r["offic_inf"]) for r in rows]).astype(np.float64)
    dates = [r["date"] for r in rows]
    ends = list(ends)
    W, ld, K = len(ends), max(ends), 3
    Y = np.zeros((W, ld)); Tw = np.array(ends, dtype=np.int32)
    yreal = np.full((W, 1), np.nan)
    for i, e in enumerate(ends):
        Y[i, :e] = y[:e]
        if e + 12 <= len(y):
            yreal[i, 0] = y[e + 11]
    NS = 3 * K + K * K + 2
    B = nrun // nbatch
    sums = np.zeros((nbatch + 1, W, NS))
    st, base = None, 0
    t0 = time.perf_counter()
    kms = 0.0
    for b in range(nbatch):
        count = burnin + B if b == 0 else (B if b < nbatch - 1 else 0)
        r = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), yreal, want_draws=False, want_state=True,
                                     sweep_base=base, sweep_count=count, resume_state=st)
        kms += r["kernel_ms"]
        st = r
        sums[b + 1] = r["sumacc"][:, :NS]
        base += count if count else 0
    wall = time.perf_counter() - t0
    bm = np.diff(sums, axis=0) / B                       # (nbatch, W, NS) batch means of the 5-digit-rounded draws
    mean = r["summary"]
    mcse = bm.std(axis=0, ddof=1) / np.sqrt(nbatch)
    return dict(mean=mean, mcse=mcse, status=r["status"], dates=[dates[e - 1] for e in ends], wall=wall, kernel_ms=kms,
                batch_mean_check=np.max(np.abs(bm.mean(axis=0) - mean)))


def load_fixture():
    out = {}
    for n in FILES:
        rows = list(csv.reader(open(os.path.join(GOLDEN, "official_%s_summary.csv" % n))))
        out[n] = (rows[0], {r[0]: np.array([float(v) for v in r[1:]]) for r in rows[1:]})
    return out


def compare(run, fx):
    K = 3
    cols = {"filtered_means": slice(0, K), "filtered_variances": slice(K, 2 * K), "filtered_state_probs": slice(2 * K, 3 * K),
            "filtered_trans_probs": slice(3 * K, 3 * K + K * K), "forecasts": slice(3 * K + K * K, 3 * K + K * K + 2)}
    res = {}
    for n, sl in cols.items():
        ref = np.stack([fx[n][1][d] for d in run["dates"]])
        got, se = run["mean"][:, sl], run["mcse"][:, sl]
        z = (got - ref) / (np.sqrt(2.0) * se + 1e-7)
        res[n] = dict(z=z, diff=got - ref, se=se, ref=ref)
    return res


if __name__ == "__main__":
    burnin = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nrun = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
    run = official_run(burnin, nrun)
    print("run: %d windows, %.1f s wall, %.1f s kernels, flagged %d, batch-mean identity %.2e" % (
        len(run["dates"]), run["wall"], run["kernel_ms"] / 1e3, int((run["status"] != 0).sum()), run["batch_mean_check"]))
    if len(sys.argv) > 3:
        np.savez(sys.argv[3], mean=run["mean"], mcse=run["mcse"], dates=np.array(run["dates"]))
    res = compare(run, load_fixture())
    for n, r in res.items():
        z = r["z"]
        w, c = np.unravel_index(np.argmax(np.abs(z)), z.shape)
        print("%-22s rows %d cols %d | max|diff| %.5f | |z|: median %.2f  p99 %.2f  max %.2f (row %s col %d: got-ref %.5f, mcse %.5f, ref %.5f) | frac |z|>4: %.4f"
              % (n, z.shape[0], z.shape[1], np.max(np.abs(r["diff"])), np.median(np.abs(z)), np.quantile(np.abs(z), 0.99), np.abs(z[w, c]),
                 run["dates"][w], c, r["diff"][w, c], r["se"][w, c], r["ref"][w, c], np.mean(np.abs(z) > 4)))
