import os, sys
os.environ["HMCG_DIAG"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch, ctypes as C
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
rng = np.random.default_rng(0)
def run(K, T, W, n, env):
    for k in ("HMCG_FLAVOUR", "HMCG_FORCE_BIG"): os.environ.pop(k, None)
    os.environ.update(env)
    base, _, fut0 = synth.generate_panel(8, T, K)
    idx = rng.integers(0, 8, W)
    Y = base[idx] + rng.normal(0, 1e-3, (W, T)); Tw = np.full(W, T, dtype=np.int32); yr = np.ascontiguousarray(fut0[idx, 11:12])
    dY = torch.from_numpy(Y).cuda(); dT = torch.from_numpy(Tw).cuda(); dyr = torch.from_numpy(yr).cuda()
    summ = torch.zeros((W, 3 * K + K * K + 2), dtype=torch.float64, device="cuda"); st = torch.zeros(W, dtype=torch.int32, device="cuda")
    cfg = _lib.make_config(W, K, T, T, 0, n, (12,))
    ms = []
    for _ in range(3):
        tm = _lib.estimate_batch_device(cfg, dY.data_ptr(), dT.data_ptr(), dyr.data_ptr(), 0, 0, 0, 0, 0, summ.data_ptr(), st.data_ptr(), None, None, True)
        ms.append(tm.kernel_ms)
    print("K=%d T=%d W=%d %-28s %8.2f ms  %6.2f M draws/s  L=%d NT=%d occ=%d" % (K, T, W, env, min(ms[1:]), W * n / min(ms[1:]) / 1e3, tm.steps_per_thread, tm.threads_per_window, tm.occupancy), flush=True)
for W, n in ((2048, 100), (256, 200)):
    for env in ({}, {"HMCG_FLAVOUR": "p1"}, {"HMCG_FLAVOUR": "p2"}, {"HMCG_FORCE_BIG": "1"}):
        run(3, 4072, W, n, env)
    run(3, 3000, W, n, {})
    run(3, 3000, W, n, {"HMCG_FORCE_BIG": "1"})
    run(3, 2024, W, n, {})
    run(3, 1500, W, n, {})
