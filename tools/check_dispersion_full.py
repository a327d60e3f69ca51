"""calcdispersion on the reference's FULL committed `forecasts_summary.csv` files (45 500 rows each; too large to carry as
fixtures) against its committed `forecasts_dispersion.csv`: every line must match character for character.  Runs only
where /root/reference is mounted (the build container); prints the number of differing lines per noise level."""
import os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import hmc
REF = "/root/reference/data/output"
for noise in ("0.1", "0.3", "0.6"):
    d = os.path.join(REF, "signals_official_noise_%s_allsignal" % noise)
    if not os.path.isdir(d):
        print("reference not mounted:", d); continue
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.join(d, "forecasts_summary.csv"), tmp)
        hmc.calcdispersion(tmp)
        a = open(os.path.join(tmp, "forecasts_dispersion.csv")).read().splitlines()
        b = open(os.path.join(d, "forecasts_dispersion.csv")).read().splitlines()
        print("noise %s: %d lines, %d differing" % (noise, len(b), sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))))
