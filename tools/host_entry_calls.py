"""A few host-entry calls at the headline shape, for `rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/host_entry_calls.py`
(when each chunk's copy starts and ends relative to the kernels).  usage: python tools/host_entry_calls.py [calls=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W, T, K, DRAWS = 256, 1000, 3, 1000
Y, Tw, fut = synth.generate_panel(W, T, K)
out = None
for _ in range(n):
    out = _lib.estimate_batch_host(Y, Tw, K, 0, DRAWS, (12,), fut[:, 11:12], out=out)
print("ok", out["status"].max())
