"""Investigation of the one official-summary row (end date 2011-07-01) that the 460-window pin flags: is the posterior
of that window explored differently by different chains?  Runs the window and its two neighbours under 64 RNG streams
each at upstream's sweep counts and prints the spread of the per-chain posterior means."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import _lib
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rows = list(csv.DictReader(open(os.path.join(G, "inflation.csv"))))
y = np.array([np.float32(r["offic_inf"]) for r in rows]).astype(np.float64)
dates = [r["date"] for r in rows]
for d in ("2011-06-01", "2011-07-01", "2011-08-01"):
    e = dates.index(d) + 1
    n = 64
    Y = np.tile(y[:e], (n, 1))
    r = _lib.estimate_batch_host(Y, [e] * n, 3, 100000, 250000, (12,), np.full((n, 1), y[e + 11]), want_draws=False,
                                 window_ids=np.arange(1000, 1000 + n))
    s = r["summary"]
    print(d, "T", e, "last obs", y[e - 3:e])
    for name, c in (("mu3", 2), ("sig2_3", 5), ("p1", 6), ("p2", 7), ("fcast", 18)):
        v = s[:, c]
        print("   %-7s chains: mean %.4f sd %.4f min %.4f max %.4f  | quantiles %s" % (name, v.mean(), v.std(), v.min(), v.max(),
              np.round(np.quantile(v, [0.05, 0.25, 0.5, 0.75, 0.95]), 4)))
