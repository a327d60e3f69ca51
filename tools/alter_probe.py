"""Why data/output/alter/*_summary.csv (3 rows, the reference's only other committed posterior summaries) is NOT a pin.
The rows are end indices 250, 251, 255 of the alternative inflation series; slurmscripts/make_test_data.sh:8-10 produces such
rows with the TEST settings of code/run_hmm.jl:76-92 (burnin = 10, Nrun = 10).  This probe runs the oracle's base model on those
windows at every short chain length that could be meant -- (burnin, Nrun) from (0, 1) to (10, 10) -- and at stationarity, 300
seeds each, and prints where the committed values fall in the distribution of the n-draw means.
Result (this container, oracle = literal restatement of src/Hmc.jl:231-562): the committed state variances (2.27, 1.61, 5.91 at
1990-10-01) exceed EVERY one of the 300 chains' means at every chain length (ours: 0.6-0.9, 0.5-0.7, 3.6-3.9, at stationarity
0.90, 0.52, 3.59); the committed forecast (6.68) lies below all of them (8.34).  The identity forecast - error = y[e+12] does
hold for the committed rows, so the data are the same series.  No setting of the CURRENT source's base model produces these
numbers -- they come from another model configuration or code generation (the directory holds two of the five summary files) --
so they are recorded here and pinned nowhere.    python tools/alter_probe.py [/root/reference]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
rows = list(csv.DictReader(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inflation.csv"))))
y = np.array([np.float32(r["alter_inf"]) for r in rows]).astype(np.float64)
dates = [r["date"] for r in rows]
var = {r["date"]: [float(r["state_%d_mean" % k]) for k in (1, 2, 3)] for r in csv.DictReader(open(os.path.join(ref, "data/output/alter/filtered_variances_summary.csv")))}
fc = {r["date"]: float(r["forecast_12_mean"]) for r in csv.DictReader(open(os.path.join(ref, "data/output/alter/forecasts_summary.csv")))}
for d in sorted(var):
    e = dates.index(d) + 1
    fx = np.array(var[d] + [fc[d]])
    print(d, "end index", e, "committed (variances 1-3, forecast):", fx)
    for b, n in ((0, 1), (0, 3), (3, 5), (0, 10), (10, 10), (1000, 1000)):
        out = []
        for sd in range(300 if n < 100 else 20):
            o = oracle.estimate_window(y[:e], 3, b, n, (12,), [y[e + 11]], seed=100 + sd, window_id=sd)
            out.append(np.concatenate([o["summary"][3:6], o["summary"][-2:-1]]))
        out = np.array(out)
        print("   burnin %4d Nrun %4d: ours mean %s | fraction of our chains below the committed value %s" % (
            b, n, np.round(out.mean(0), 3), np.round((out < fx[None, :]).mean(0), 3)))
