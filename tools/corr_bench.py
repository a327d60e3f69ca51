import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import _lib, synth
W, T, K, D = 256, 1000, 3, 1000
Y, Tw, fut = synth.generate_panel(W, T, K, horizon_pad=12)
yr = fut[:, 11:12]
def run(**kw):
    out = None
    for _ in range(3):
        out = _lib.estimate_batch_host(Y, Tw, K, 0, D, (12,), yr, out=out, **kw)
    t = []
    for _ in range(9):
        t0 = time.perf_counter(); out = _lib.estimate_batch_host(Y, Tw, K, 0, D, (12,), yr, out=out, **kw); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t)), out["kernel_ms"]
r = {"draws_to_host": run(), "draws_to_host_plus_corr": run(want_corr=True), "summary_only": run(want_draws=False), "summary_plus_corr_no_draw_copy": run(want_draws=False, want_corr=True)}
print(json.dumps(r))
