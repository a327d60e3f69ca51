"""Time an arbitrary (K, T, W, draws) shape on one GPU: python tools/bench_cfg.py K T W draws [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import synth
from hmc_jl_amd.device import DevicePanel
K, T, W, n = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
smooth = len(sys.argv) > 6 and sys.argv[6] == "smooth"
rng = np.random.default_rng(0)
if W * T > 300000:      # fast synthetic panel: tile a few generated windows with per-window shifts
    base, _, fut0 = synth.generate_panel(8, T, K)
    idx = rng.integers(0, 8, W)
    Y = base[idx] + rng.normal(0, 1e-3, (W, T)); fut = fut0[idx]
    Tw = np.full(W, T, dtype=np.int32)
else:
    Y, Tw, fut = synth.generate_panel(W, T, K)
p = DevicePanel(Y, Tw, K, n, (12,), fut[:, 11:12])
if smooth:
    import ctypes as C
    import torch
    from hmc_jl_amd import _lib
    sm = torch.zeros((W, T, K), dtype=torch.float64, device="cuda")
    def run():
        cfg = _lib.make_config(W, K, T, T, 0, n, (12,))
        ex = _lib.Extras(); ex.struct_size = C.sizeof(_lib.Extras); ex.pi_smooth_mean = sm.data_ptr()
        return _lib.estimate_batch_device(cfg, p.Y.data_ptr(), p.T.data_ptr(), p.yreal.data_ptr(), 0, 0, 0, 0, 0,
                                          p.summary.data_ptr(), p.status.data_ptr(), ex, None, True).kernel_ms
    ms = [run() for _ in range(reps + 1)][1:]
    p.last_timing = type("T", (), dict(threads_per_window=256, steps_per_thread=(T + 255) // 256, lds_bytes=0))()
else:
    ms = [p.run(burnin=0) for _ in range(reps + 1)][1:]
B = T * (8 + 16 * K + 2) + 8 * (3 * K + K * K + 2)
k = float(np.mean(ms))
tm = p.last_timing
print("K=%d T=%d W=%d draws=%d: kernel %.3f ms -> %.3f M draws/s; algorithmic %.0f B/draw -> %.1f GB/s (%.1f%% of 8 TB/s); NT=%d L=%d lds=%d flagged=%d"
      % (K, T, W, n, k, W * n / k / 1e3, B, B * W * n / k / 1e6, B * W * n / k / 1e6 / 80.0, tm.threads_per_window, tm.steps_per_thread,
         tm.lds_bytes, int((p.status != 0).sum().item())))
