import os, sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
from hmc_jl_amd import device as hdev, synth
from hmc_jl_amd.device import DevicePanel
lens = list(range(120, 580))
Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
panel = hdev.DevicePanel(Y, Tw, 3, 1000, (12,), fut[:, 11:12], keep_draws=True)
panel.run(burnin=0)
ms = [panel.run(burnin=0) for _ in range(6)]
Y2, T2, f2 = synth.generate_panel(256, 1000, 3)
p2 = DevicePanel(Y2, T2, 3, 1000, (12,), f2[:, 11:12])
for _ in range(4): p2.run(burnin=0)
m2 = [p2.run(burnin=0) for _ in range(8)]
print("%-24s production %7.3f ms (min %7.3f)   headline %7.4f ms (min %7.4f)" % (os.environ.get("HMCG_LIB", "new"), np.mean(ms), min(ms), np.mean(m2), min(m2)), flush=True)
