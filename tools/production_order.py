"""Does the order of the windows in the call matter?  The reference's production shape (460 windows, T = 120..579, K = 3, 1000
sweeps) with its windows ascending (as the reference builds them), descending, shuffled, and long/short interleaved, under the
length-bucketed dispatch (optionally with forced flavours per class) and as one launch.  usage: python tools/production_order.py"""
import os, sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import device as hdev, synth
draws = 1000
base = list(range(120, 580))
rng = np.random.default_rng(1)
orders = {"ascending": base, "shuffled": list(rng.permutation(base))}
for name, lens in orders.items():
    lens = [int(x) for x in lens]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    panel = hdev.DevicePanel(Y, Tw, 3, draws, (12,), fut[:, 11:12], keep_draws=True)
    for label, env, bucketed in (("bucketed (table)", {}, True), ("bucketed p2,p2,p2", {"HMCG_BUCKET_FLAVOURS": "p2,p2,p2"}, True),
                                 ("bucketed p1,p1,p1", {"HMCG_BUCKET_FLAVOURS": "p1,p1,p1"}, True), ("one launch", {}, False)):
        os.environ.pop("HMCG_BUCKET_FLAVOURS", None)
        os.environ.update(env)
        panel.run(burnin=0, bucketed=bucketed)
        ms = [panel.run(burnin=0, bucketed=bucketed) for _ in range(4)]
        print("%-12s %-20s %7.3f ms (min %7.3f)  %6.2f M draws/s" % (name, label, np.mean(ms), min(ms), len(lens) * draws / np.mean(ms) / 1e3), flush=True)
