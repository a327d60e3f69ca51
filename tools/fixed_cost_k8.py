"""Per-launch fixed cost of the LDS-resident kernel at configs[3] (K=8, T=5000, 512 windows): kernel time of short runs."""
import os, sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
rng = np.random.default_rng(0)
base, _, fut0 = synth.generate_panel(8, 5000, 8)
idx = rng.integers(0, 8, 512)
Y = base[idx] + rng.normal(0, 1e-3, (512, 5000)); fut = fut0[idx]
Tw = np.full(512, 5000, dtype=np.int32)
for n in (1, 2, 5, 10, 20, 50):
    p = DevicePanel(Y, Tw, 8, n, (12,), fut[:, 11:12])
    ms = [p.run(burnin=0) for _ in range(6)]
    print("nrun=%4d  kernel %.4f ms (min %.4f)  per sweep %.3f us" % (n, float(np.median(ms[2:])), min(ms[2:]), 1e3 * float(np.median(ms[2:])) / n), flush=True)
