"""extras.corr against the reference's committed data/output/official/correlations.xlsx (calccorr, src/Hmc.jl:1094-1163;
numbers extracted by tools/make_corr_fixture.py): the 455 end dates 1980-01 .. 2017-11 at upstream's own 100k + 250k sweeps,
R independent replicas (RNG stream ids offset per replica), matrices accumulated on the device -- no draw leaves the GPU.
The replicas give the Monte-Carlo spread of a correlation estimated from one 250k-draw chain (what the fixture is).
Exploratory twin of tests/test_gpu_golden.py::test_correlations_vs_committed_workbook."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_fixture():
    rows = list(csv.reader(open(os.path.join(GOLDEN, "official_correlations_forecast_row.csv"))))
    dates = [r[0] for r in rows[1:]]
    frow = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    mats = {}
    for r in list(csv.reader(open(os.path.join(GOLDEN, "official_correlations_matrices.csv"))))[1:]:
        mats.setdefault(r[0], []).append([float(v) for v in r[2:]])
    return dates, frow, {d: np.array(m) for d, m in mats.items()}


def corr_run(burnin=100000, nrun=250000, replicas=4):
    rows = list(csv.DictReader(open(os.path.join(GOLDEN, "inflation.csv"))))
    y = np.array([np.float32(r["offic_inf"]) for r in rows]).astype(np.float64)
    alld = [r["date"] for r in rows]
    dates, frow, mats = load_fixture()
    ends = [alld.index(d) + 1 for d in dates]
    W, ld, K = len(ends), max(ends), 3
    Y = np.zeros((W, ld)); Tw = np.array(ends, dtype=np.int32)
    yreal = np.zeros((W, 1))
    for i, e in enumerate(ends):
        Y[i, :e] = y[:e]
        yreal[i, 0] = y[e + 11] if e + 12 <= len(y) else np.nan
    t0 = time.perf_counter()
    cs, kms = [], 0.0
    for rep in range(replicas):
        r = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), yreal, want_draws=False, want_corr=True,
                                     window_ids=np.array(ends) + 100000 * rep)
        assert (r["status"] == 0).all()
        cs.append(r["corr"]); kms += r["kernel_ms"]
    c = np.stack(cs)                                    # (R, W, 19, 19)
    return dict(dates=dates, corr=c, frow=frow, mats=mats, wall=time.perf_counter() - t0, kernel_ms=kms)


def compare(run):
    c = run["corr"]
    R = c.shape[0]
    with np.errstate(invalid="ignore"):
        m = np.nanmean(c, axis=0)
        sd = np.nanstd(c, axis=0, ddof=1) if R > 1 else np.zeros_like(m)
    ours_row, sd_row = m[:, -1, :], sd[:, -1, :]
    ref = run["frow"]
    both = np.isfinite(ref) & np.isfinite(ours_row)
    tol = 5.0 * np.sqrt(1.0 + 1.0 / R) * sd_row + 0.02
    out = dict(nan_agree=float(np.mean(np.isfinite(ref) == np.isfinite(ours_row))), both=both, diff=np.where(both, ours_row - ref, 0.0), tol=tol,
               z=np.where(both, (ours_row - ref) / (np.sqrt(1.0 + 1.0 / R) * sd_row + 1e-3), 0.0))
    md = []
    for i, d in enumerate(run["dates"]):
        if d in run["mats"]:
            ok = np.isfinite(run["mats"][d]) & np.isfinite(m[i])
            md.append(np.max(np.abs(np.where(ok, m[i] - run["mats"][d], 0.0))))
    out["matrix_max_diff"] = np.array(md)
    return out


if __name__ == "__main__":
    run = corr_run(*(int(a) for a in sys.argv[1:4]))
    c = compare(run)
    print("%d dates x %d replicas, %.1f s wall, kernels %.1f s" % (len(run["dates"]), run["corr"].shape[0], run["wall"], run["kernel_ms"] / 1e3))
    print("forecast row: NaN pattern agreement %.4f; |diff| median %.4f q99 %.4f max %.4f; outside tolerance %d of %d; |z| median %.2f q99 %.2f" % (
        c["nan_agree"], np.median(np.abs(c["diff"][c["both"]])), np.quantile(np.abs(c["diff"][c["both"]]), 0.99), np.abs(c["diff"]).max(),
        int((np.abs(c["diff"]) > c["tol"]).sum()), int(c["both"].sum()), np.median(np.abs(c["z"][c["both"]])), np.quantile(np.abs(c["z"][c["both"]]), 0.99)))
    print("full matrices (%d dates): max |diff| per date: median %.4f max %.4f" % (len(c["matrix_max_diff"]), np.median(c["matrix_max_diff"]), c["matrix_max_diff"].max()))
