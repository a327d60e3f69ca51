"""The reference's production shape: 460 expanding windows T = 120..579 (K = 3, h = 12), one call vs one call per
steps-per-thread group (T <= 256, <= 512, <= 1024).  Prints kernel times (HIP events, host entry, summary only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lens = list(range(120, 580))
Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
yr = fut[:, 11:12]
def run(idx):
    idx = np.asarray(idx)
    ld = int(Tw[idx].max())
    best = 1e9
    for _ in range(3):
        r = _lib.estimate_batch_host(np.ascontiguousarray(Y[idx, :ld]), Tw[idx], 3, 0, draws, (12,), yr[idx], want_draws=False,
                                     window_ids=idx)
        best = min(best, r["kernel_ms"])
    return best, r["steps_per_thread"], r["helper_waves"]
allw = np.arange(len(lens))
t, L, nh = run(allw)
print("one call: %d windows, L=%d helper_waves=%d: %.2f ms -> %.2f M draws/s" % (len(lens), L, nh, t, len(lens) * draws / t / 1e3))
tot = 0.0
for lo, hi in ((0, 256), (256, 512), (512, 1024)):
    idx = allw[(Tw > lo) & (Tw <= hi)]
    if len(idx):
        t, L, nh = run(idx)
        tot += t
        print("  group T in (%d, %d]: %d windows, L=%d helper_waves=%d: %.2f ms" % (lo, hi, len(idx), L, nh, t))
print("grouped, back to back: %.2f ms -> %.2f M draws/s" % (tot, len(lens) * draws / tot / 1e3))
