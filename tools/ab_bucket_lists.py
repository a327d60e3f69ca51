import os, sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
from hmc_jl_amd import device as hdev, synth
lens = list(range(120, 580))
Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
panel = hdev.DevicePanel(Y, Tw, 3, 1000, (12,), fut[:, 11:12], keep_draws=True)
for rep in range(4):
    for name, env in (("lists", None), ("no lists", "1")):
        if env: os.environ["HMCG_NO_BUCKET_LISTS"] = env
        else: os.environ.pop("HMCG_NO_BUCKET_LISTS", None)
        panel.run(burnin=0)
        ms = [panel.run(burnin=0) for _ in range(5)]
        print("%-9s %7.3f ms (min %7.3f)" % (name, np.mean(ms), min(ms)), flush=True)
