"""Throughput of the signal Monte-Carlo variant at the headline shape (K=3, T=1000, 256 windows):
all positions signals, noiseSamples chains of (burnin + nrun) sweeps back to back in one launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
K, T, W = 3, 1000, 256
ns, burn, n = 10, 0, 100
Y, Tw, fut = synth.generate_panel(W, T, K)
sig = np.tile([0, T], (W, 1)); save = np.tile([T - 2, T], (W, 1))
kw = dict(sig_range=sig, save_range=save, sigma_signal=np.full(W, 0.8), kappa=0.3, n_samples=ns, alpha=2.0, nu=2.0)
for rep in range(3):
    g = _lib.estimate_batch_host(Y, Tw, K, burn, n, (12,), fut[:, 11:12], want_draws=False, **kw)
draws = W * ns * (burn + n)
print("signal path K=3 T=1000 W=256, %d samples x %d sweeps: kernel %.3f ms -> %.2f M draws/s (flagged %d)" % (
    ns, burn + n, g["kernel_ms"], draws / g["kernel_ms"] / 1e3, int((g["status"] != 0).sum())))
