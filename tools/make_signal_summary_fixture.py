"""Condenses the reference's committed data/output/signals_official_noise_<n>_allsignal/forecasts_summary.csv (45 500 rows:
one per (date, signalid) = one noise sample's mean forecast, mean forecast error and its two saved noisy signals; 4 MB per
noise level) into one row of statistics per date: tests/golden/signals_noise_<n>_allsignal_forecasts_summary_stats.csv.
Per date (plain 100-sample dates only): n, mean, std, quartiles of forecast_12_mean across the noise samples, and the
least-squares slope (with its standard error) of forecast_12_mean on signal_2_mean (the last noisy signal the sample saw)
and on signal_1_mean.  DATA derived from reference outputs -- runs only where /root/reference exists (the build container);
the GPU box uses the committed CSVs.    python tools/make_signal_summary_fixture.py [/root/reference]"""
import csv
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for noise in ("0.1", "0.3", "0.6"):
    rows = {}
    with open(os.path.join(ref, "data", "output", "signals_official_noise_%s_allsignal" % noise, "forecasts_summary.csv")) as fh:
        for r in csv.DictReader(fh):
            rows.setdefault(r["date"], []).append((int(float(r["signalid"])), float(r["forecast_12_mean"]), float(r["forecast_error_12_mean"]),
                                                   float(r["signal_1_mean"]), float(r["signal_2_mean"])))
    out = os.path.join(GOLDEN, "signals_noise_%s_allsignal_forecasts_summary_stats.csv" % noise)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow(["date", "n", "f_mean", "f_std", "f_q25", "f_q50", "f_q75", "slope_s2", "slope_s2_se", "slope_s1", "slope_s1_se", "corr_s1_s2"])
        kept = 0
        for d in sorted(rows):
            a = np.array(sorted(rows[d]))
            if len(a) != 100 or not np.array_equal(a[:, 0], np.arange(1, 101)):
                continue                                   # dates upstream ran twice (or partially): as in golden_signals.signal_run
            f, s1, s2 = a[:, 1], a[:, 3], a[:, 4]
            rec = [d, len(a), f.mean(), f.std(ddof=1)] + list(np.quantile(f, [0.25, 0.5, 0.75]))
            for s in (s2, s1):
                fc, sc = f - f.mean(), s - s.mean()
                b = (fc * sc).sum() / (sc * sc).sum()
                se = np.sqrt(((fc - b * sc) ** 2).sum() / (len(a) - 2) / (sc * sc).sum())
                rec += [b, se]
            rec.append(np.corrcoef(s1, s2)[0, 1])
            w.writerow([rec[0], rec[1]] + [repr(float(x)) for x in rec[2:]])
            kept += 1
    print(noise, kept, "dates ->", out)
