"""In-kernel shader clock of the sweep kernels (MI355X_MICROARCH.md "DVFS give-back" item 6):
the diagnostic (stamped) build is launched back to back for >= `warm_s` seconds, then one more launch prints its
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the sweep loop (median over every wave of every window) and
the phase table.  usage: python tools/clock_stamps.py [K=3] [T=1000] [W=256] [draws=1000] [warm_s=2.5]"""
import os, sys, time
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
os.environ.setdefault("HMCG_LIB", "libhmcgibbs_stamps.so")
os.environ["HMCG_STAMPS_AFTER"] = "1000000"      # replaced below: the library reads it at its first launch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
warm_s = float(sys.argv[5]) if len(sys.argv) > 5 else 2.5
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
rng = np.random.default_rng(0)
if W * T > 300000:
    base, _, fut0 = synth.generate_panel(8, T, K)
    idx = rng.integers(0, 8, W)
    Y = base[idx] + rng.normal(0, 1e-3, (W, T)); fut = fut0[idx]
    Tw = np.full(W, T, dtype=np.int32)
else:
    Y, Tw, fut = synth.generate_panel(W, T, K)
p = DevicePanel(Y, Tw, K, n, (12,), fut[:, 11:12])
# how many launches fill warm_s?  time two with the unstamped wall clock first (they are silent: HMCG_STAMPS_AFTER is huge)
# -- the library latches HMCG_STAMPS_AFTER at its first stamped launch, so the count must be known before that: estimate it
# from the shape instead (sweeps x windows / a conservative rate), then top up by time below.
est_ms = {3: 6.5, 8: 230.0}.get(K, 20.0) * (W / (256.0 if K <= 4 else 512.0)) * (n / 1000.0) * (T / (1000.0 if K <= 4 else 5000.0))
nwarm = max(3, int(warm_s * 1e3 / max(est_ms, 0.05)) + 1)
os.environ["HMCG_STAMPS_AFTER"] = str(nwarm)
t0 = time.time()
ms = [p.run(burnin=0) for _ in range(nwarm)]
t1 = time.time()
print("warm-up: %d launches in %.2f s (kernel %.3f ms each, stamped build -- not a benchmark)" % (nwarm, t1 - t0, float(np.mean(ms[1:]))), flush=True)
last = p.run(burnin=0)          # prints [clock] and the phase table on stderr
print("measured launch: kernel %.3f ms" % last)
