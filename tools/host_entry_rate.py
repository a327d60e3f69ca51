"""PCIe-inclusive rate of the host-buffer entry (hmcg_estimate_batch) at the headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
Y, Tw, fut = synth.generate_panel(256, 1000, 3)
for draws, label in ((True, "all per-draw outputs (41 MB D2H into pageable numpy arrays)"), (False, "summary only (41 KB D2H)")):
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        g = _lib.estimate_batch_host(Y, Tw, 3, 0, 1000, (12,), fut[:, 11:12], want_draws=draws)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[1:]))
    print("host entry, %s: %.2f ms per call (kernel %.2f ms) -> %.1f M draws/s" % (label, t * 1e3, g["kernel_ms"], 256e3 / t / 1e6))
