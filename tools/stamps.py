"""Run cfg2 once on the diagnostic (stamped) build and print the per-phase cycle table."""
import os, sys
os.environ["HMCG_LIB"] = "libhmcgibbs_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
tpw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
Y, Tw, fut = synth.generate_panel(256, 1000, 3)
g = _lib.estimate_batch_host(Y, Tw, 3, 0, 300, (12,), fut[:, 11:12], threads_per_window=tpw)
print("kernel_ms (stamped build, not a benchmark): %.3f" % g["kernel_ms"])
