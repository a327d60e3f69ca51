"""Quick GPU-vs-oracle check used during bring-up (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd as H
from hmc_jl_amd import synth, _lib
from oracle import oracle as O

def run(K, T, W, burnin, nrun, tpw=0):
    Y, Tw, fut = synth.generate_panel(W, T, K)
    yreal = fut[:, 11:12]
    t0 = time.time()
    g = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), yreal, want_state=True, threads_per_window=tpw)
    tg = time.time() - t0
    o = O.estimate_batch(Y, Tw, K, burnin, nrun, (12,), yreal, want_state=True)
    for k in ("mu", "sig2", "A", "pi_end", "fcast", "summary"):
        d = np.abs(g[k] - o[k])
        rel = d / (1e-300 + np.abs(o[k]))
        print("  %-8s max abs %.3e  max rel %.3e" % (k, d.max(), rel.max()))
    print("  x_final mismatches:", int((g["x_final"] != o["x_final"]).sum()), " pif max abs %.3e" % np.abs(g["pif_final"] - o["pif_final"]).max())
    print("  status gpu", g["status"][:8], "oracle", o["status"][:8])
    n = W * (burnin + nrun)
    print("  K=%d T=%d W=%d sweeps=%d: kernel %.3f ms -> %.3f M draws/s (L=%d NT=%d lds=%d); host call %.2fs" % (
        K, T, W, burnin + nrun, g["kernel_ms"], n / g["kernel_ms"] / 1e3, g["steps_per_thread"], g["threads_per_window"], g["lds_bytes"], tg))

if __name__ == "__main__":
    print("devices", _lib.load().hmcg_device_count())
    run(3, 200, 4, 0, 50)
    run(3, 1000, 8, 5, 100)
    run(2, 476, 4, 10, 100)
    run(4, 700, 4, 0, 60)
    run(3, 1000, 256, 0, 1000)
    run(3, 1000, 256, 0, 1000, 512)
    run(3, 1000, 256, 0, 1000, 128)
