"""Per-launch fixed cost of the sweep kernel at the headline shape (K=3, T=1000, 256 windows): kernel time of runs of 1..1000
sweeps through the device entry, and of 1- and 31-sweep runs through the host entry in one launch with and without
caller-provided initial states (x_init skips makeParams' median / nearest-mean start).  profiles/r04/fixed_cost.txt.
usage: python tools/fixed_cost.py"""
import os, sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
Y, Tw, fut = synth.generate_panel(256, 1000, 3)
for n in (1, 2, 5, 10, 31, 63, 125, 250, 500, 1000):
    p = DevicePanel(Y, Tw, 3, n, (12,), fut[:, 11:12])
    ms = [p.run(burnin=0) for _ in range(12)]
    print("nrun=%4d  kernel %.4f ms (min %.4f)  per sweep %.3f us" % (n, float(np.median(ms[2:])), min(ms[2:]), 1e3 * float(np.median(ms[2:])) / n))
# the same through the host entry in one launch, with and without caller-provided initial states (x_init skips makeParams'
# median / nearest-mean assignment): what the fresh start costs inside the kernel
os.environ["HMCG_DIAG"] = "1"; os.environ["HMCG_NO_CHUNKS"] = "1"
from hmc_jl_amd import _lib
xi = np.zeros((256, 1000), dtype=np.int32); xi[:, 500:] = 1; xi[:, 800:] = 2
for n in (1, 31):
    for name, kw in (("makeParams init", {}), ("x_init", {"x_init": xi})):
        ks = []
        for _ in range(8):
            o = _lib.estimate_batch_host(Y, Tw, 3, 0, n, (12,), fut[:, 11:12], **kw)
            ks.append(o["kernel_ms"])
        print("nrun=%4d host entry, one launch, %-16s kernel %.4f ms (min %.4f)" % (n, name, float(np.median(ks[2:])), min(ks[2:])))
