"""Kernel time of the reference's production shape (460 expanding windows, T = 120..579, K = 3; code/run_hmm.jl:79-109) under
the length-bucketed dispatch with different flavours per bucket (longest first), and as one launch.
Usage: python tools/bucket_sweep.py [draws]"""
import os
import sys
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import device as hdev, synth

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lens = list(range(120, 580))
Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
panel = hdev.DevicePanel(Y, Tw, 3, draws, (12,), fut[:, 11:12], keep_draws=True)


def timed(label, bucketed=True):
    panel.run(burnin=0, bucketed=bucketed)
    ms = [panel.run(burnin=0, bucketed=bucketed) for _ in range(4)]
    tm = panel.last_timing
    print("%-22s %7.3f ms (min %7.3f)  %6.2f M draws/s  buckets %d" % (label, np.mean(ms), min(ms), len(lens) * draws / np.mean(ms) / 1e3, tm.buckets), flush=True)


timed("one launch", bucketed=False)
timed("table")
for combo in ("p2,p2,p2", "p1,p1,p1", "p1,p1,p2", "p1,p2,p2", "p2,p1,p1", "p2,p2,p1", "h,p1,p1", "p1,p2,p1"):
    os.environ["HMCG_BUCKET_FLAVOURS"] = combo
    timed(combo)
