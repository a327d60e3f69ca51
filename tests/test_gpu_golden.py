"""The reference's committed outputs, every row.

(1) data/output/official/*_summary.csv (tests/golden/official_*): all 460 expanding windows (end indices 120..579) in ONE
    GPU run at upstream's own 100k burn-in + 250k kept sweeps (code/run_hmm.jl:103-104; ~3.5 s of GPU time), every row of
    all five files.  Tolerance per value: max(5 * sqrt(2) * MCSE, floor) with MCSE from 25 batch means of this run (the
    factor sqrt(2): the fixture is itself one chain of the same length) and the floors of the 3-row test
    (means 0.04, variances 0.08 + 2 %, probabilities and transition probabilities 0.003, forecasts 0.03).
    459 of 460 rows are inside on all five files.  The one row outside, end date 2011-07-01, is an anomaly OF THE FIXTURE:
    its state-3 mean (8.526) jumps +0.38 against its own neighbours (8.211 the month before, 8.077 the month after; the
    windows differ by one observation), its state-3 variance drops from 14.8 to 13.2 and back; 64 independent GPU chains
    of that window at the same sweep counts all give 8.188 +- 0.006 (tools/golden_outlier.py), in line with the
    neighbours.  The test pins that diagnosis instead of widening the tolerance.
(2) data/output/signals_official_noise_{0.1,0.3,0.6}_allsignal/forecasts_dispersion.csv (tests/golden/signals_noise_*):
    all 455 end dates x 3 noise levels, 100 noise samples each, one GPU call per level; per date the mean and the
    across-sample standard deviation of the per-sample mean forecast, and the means of the saved noisy signals."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
FLOORS = {"filtered_means": 0.04, "filtered_variances": 0.08, "filtered_state_probs": 0.003, "filtered_trans_probs": 0.003,
          "forecasts": 0.03}
FIXTURE_ANOMALY = "2011-07-01"


def test_all_460_official_windows_all_five_files(hmclib):
    import golden_pin as gp
    run = gp.official_run(100000, 250000, nbatch=25)
    assert (run["status"] == 0).all() and len(run["dates"]) == 460
    assert run["batch_mean_check"] < 1e-9               # the resumed 25 calls are one chain: batch means average to the summary
    res = gp.compare(run, gp.load_fixture())
    out_rows = set()
    worst = {}
    for name, r in res.items():
        tol = np.maximum(5.0 * np.sqrt(2.0) * r["se"], FLOORS[name] + (0.02 * np.abs(r["ref"]) if name == "filtered_variances" else 0.0))
        bad = np.abs(r["diff"]) > tol
        out_rows |= {run["dates"][w] for w in np.nonzero(bad.any(axis=1))[0]}
        ok = ~bad.any(axis=1)
        worst[name] = float(np.max(np.abs(r["z"][ok])))
        assert np.median(np.abs(r["z"])) < 1.0, name                    # z-scores are centred: no systematic offset
        assert np.quantile(np.abs(r["z"]), 0.99) < 4.0, name
    print("worst |z| inside tolerance per file:", worst)
    assert out_rows <= {FIXTURE_ANOMALY}, sorted(out_rows)
    # the anomalous fixture row: discontinuous against its own neighbours, where this run is smooth
    d = run["dates"]
    i = d.index(FIXTURE_ANOMALY)
    fx = res["filtered_means"]["ref"][:, 2]
    ours = run["mean"][:, 2]
    assert abs(fx[i] - 0.5 * (fx[i - 1] + fx[i + 1])) > 0.3
    assert abs(ours[i] - 0.5 * (ours[i - 1] + ours[i + 1])) < 0.06
    assert abs(ours[i - 1] - fx[i - 1]) < 0.04 and abs(ours[i + 1] - fx[i + 1]) < 0.04


@pytest.mark.parametrize("noise,nrun", [("0.1", 6000), ("0.3", 2000), ("0.6", 2000)])
def test_all_signal_dates_vs_committed_dispersion(hmclib, noise, nrun):
    import golden_signals as gs
    run = gs.signal_run(noise, ns=100, burnin=1000, nrun=nrun)
    assert (run["status"] == 0).all() and len(run["dates"]) == 455
    c = gs.compare(run)
    rms = lambda z: float(np.sqrt(np.mean(z ** 2)))
    print("noise %s: rms z %.3f, max |z| %.2f, std-ratio geo-mean %.3f" % (noise, rms(c["z_mean"]), np.abs(c["z_mean"]).max(),
                                                                            np.exp(np.log(c["ratio"]).mean())))
    # mean forecast and mean forecast error across the 100 noise samples: standardised differences behave like N(0,1)
    assert np.abs(c["z_mean"]).max() < 4.75 and np.abs(c["z_err"]).max() < 4.75
    assert 0.85 < rms(c["z_mean"]) < 1.25 and abs(float(c["z_mean"].mean())) < 0.25
    # across-sample standard deviation: ratio of two 100-sample estimates (log-ratio sd ~ 0.10).  Robust statistics: a few
    # fixture rows carry an across-sample std three times that of their neighbouring dates (one stuck chain among upstream's
    # 100 samples inflates it -- e.g. 2009-10-01 at noise 0.1: 0.540 against a local median of 0.148), so the bounds are on
    # the quantiles, not on every row.  At noise 0.1 the within-sample Monte-Carlo error of these shorter chains (6000
    # draws against upstream's 250k) still adds a few per cent to ours.
    lo, hi = (0.93, 1.15) if noise == "0.1" else (0.94, 1.06)
    q05, q50, q95 = np.quantile(c["ratio"], [0.05, 0.5, 0.95])
    assert lo < q50 < hi, q50
    assert q05 > 0.72 and q95 < 1.45, (q05, q95)
    assert np.mean((c["ratio"] < 0.6) | (c["ratio"] > 1.75)) < 0.03
    # the saved noisy signals (Yreal + N(0,1) sigma_signal at the last two dates)
    assert 0.8 < rms(c["zs1"]) < 1.25 and 0.8 < rms(c["zs2"]) < 1.25
