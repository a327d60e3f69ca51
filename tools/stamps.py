"""Run cfg2 once on the diagnostic (stamped) build and print the per-phase cycle table (device entry: the table is
printed by the library).  usage: [HMCG_FLAVOUR=p1|p2|h] python tools/stamps.py [threads_per_window] [W] [T] [K]"""
import os, sys
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
os.environ.setdefault("HMCG_LIB", "libhmcgibbs_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmc_jl_amd
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
tpw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
Y, Tw, fut = synth.generate_panel(W, T, K)
p = DevicePanel(Y, Tw, K, 300 if K <= 4 else 50, (12,), fut[:, 11:12])
ms = p.run(burnin=0, threads_per_window=tpw)
print("kernel_ms (stamped build, not a benchmark): %.3f" % ms)
