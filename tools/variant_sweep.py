"""Kernel time of every (K, steps-per-thread, path) variant in its three flavours -- p1 (plain, whole register
file), p2 (plain, two blocks per CU), h (four helper waves) -- at one window per CU (W = 256) and at eight
(W = 2048).  Prints one table row per variant; the preference table in csrc/variants_*.hip is read off this table.
Usage: python tools/variant_sweep.py [draws]"""
import ctypes as C
import os
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hmc_jl_amd
from hmc_jl_amd import _lib, synth

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)


def panel(K, T, W):
    base, _, fut0 = synth.generate_panel(8, T, K)
    idx = rng.integers(0, 8, W)
    Y = base[idx] + rng.normal(0, 1e-3, (W, T))
    return Y, np.full(W, T, dtype=np.int32), fut0[idx, 11:12]


def time_one(K, T, W, path, flavour, n):
    os.environ["HMCG_FLAVOUR"] = flavour
    Y, Tw, yr = panel(K, T, W)
    dY = torch.from_numpy(Y).cuda(); dT = torch.from_numpy(Tw).cuda(); dyr = torch.from_numpy(np.ascontiguousarray(yr)).cuda()
    summ = torch.zeros((W, 3 * K + K * K + 2), dtype=torch.float64, device="cuda")
    st = torch.zeros(W, dtype=torch.int32, device="cuda")
    ex = _lib.Extras(); ex.struct_size = C.sizeof(_lib.Extras)
    keep = []
    kw = {}
    if path == "sig":
        sr = torch.from_numpy(np.tile([0, T], (W, 1)).astype(np.int32)).cuda()
        ss = torch.full((W,), 0.5, dtype=torch.float64, device="cuda")
        ex.sig_range = sr.data_ptr(); ex.sigma_signal = ss.data_ptr(); keep += [sr, ss]
        kw = dict(kappa=0.3, n_samples=2)
    if path == "smooth":
        sm = torch.zeros((W, T, K), dtype=torch.float64, device="cuda")
        ex.pi_smooth_mean = sm.data_ptr(); keep.append(sm)
    nn = n // 2 if path == "sig" else n
    cfg = _lib.make_config(W, K, T, T, 0, nn, (12,), **kw)
    ms = []
    for _ in range(3):
        tm = _lib.estimate_batch_device(cfg, dY.data_ptr(), dT.data_ptr(), dyr.data_ptr(), 0, 0, 0, 0, 0, summ.data_ptr(),
                                        st.data_ptr(), ex, None, True)
        ms.append(tm.kernel_ms)
    torch.cuda.synchronize()
    return min(ms[1:]), tm.helper_waves


rows = []
for path, Ks in (("base", (2, 3, 4)), ("sig", (2, 3, 4)), ("smooth", (2, 3, 4))):
    for K in Ks:
        Ls = {("base", 3): (1, 2, 3, 4, 6, 8, 12, 16), ("base", 2): (1, 2, 3, 4, 8), ("base", 4): (1, 2, 3, 4, 8), ("sig", 3): (1, 2, 3, 4, 8),
              ("smooth", 3): (1, 2, 4, 8)}.get((path, K), (1, 2, 4))
        for L in Ls:
            T = 256 * L - 24
            line = "%-6s K=%d L=%-2d T=%-4d" % (path, K, L, T)
            for W in (256, 2048):
                n = draws if W == 256 else max(draws // 2, 20)
                res = {}
                for fl in ("p1", "p2", "h"):
                    ms, nh = time_one(K, T, W, path, fl, n)
                    res[fl] = W * n / ms / 1e3
                best = max(res, key=res.get)
                line += " | W=%-4d " % W + " ".join("%s %6.2f" % (f, res[f]) for f in ("p1", "p2", "h")) + " best %s" % best
            print(line, flush=True)
