"""Time an arbitrary (K, T, W, draws) shape on one GPU: python tools/bench_cfg.py K T W draws [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
K, T, W, n = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rng = np.random.default_rng(0)
if W * T > 300000:      # fast synthetic panel: tile a few generated windows with per-window shifts
    base, _, fut0 = synth.generate_panel(8, T, K)
    idx = rng.integers(0, 8, W)
    Y = base[idx] + rng.normal(0, 1e-3, (W, T)); fut = fut0[idx]
    Tw = np.full(W, T, dtype=np.int32)
else:
    Y, Tw, fut = synth.generate_panel(W, T, K)
p = DevicePanel(Y, Tw, K, n, (12,), fut[:, 11:12])
ms = [p.run(burnin=0) for _ in range(reps + 1)][1:]
B = T * (8 + 16 * K + 2) + 8 * (3 * K + K * K + 2)
k = float(np.mean(ms))
tm = p.last_timing
print("K=%d T=%d W=%d draws=%d: kernel %.3f ms -> %.3f M draws/s; algorithmic %.0f B/draw -> %.1f GB/s (%.1f%% of 8 TB/s); NT=%d L=%d lds=%d flagged=%d"
      % (K, T, W, n, k, W * n / k / 1e3, B, B * W * n / k / 1e6, B * W * n / k / 1e6 / 80.0, tm.threads_per_window, tm.steps_per_thread,
         tm.lds_bytes, int((p.status != 0).sum().item())))
