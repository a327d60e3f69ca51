"""Diagnostic: the flavour-identity cases of tests/test_gpu_parity.py one call at a time, announcing each before it runs."""
import os, sys
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd
from hmc_jl_amd import _lib, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lens = [5, 200, 256, 500, 1000, 2047]
Y, Tw, fut = synth.generate_panel(len(lens), max(lens), K, ragged=lens)
for sub in ([0, 1, 2], [3], [4], [5]):
    idx = np.array(sub); ld = int(Tw[idx].max())
    args = (np.ascontiguousarray(Y[idx, :ld]), Tw[idx], K, 3, 12, (3, 12, 40),
            np.column_stack([fut[idx, 2], fut[idx, 11], np.zeros(len(idx))]))
    extra = [dict()]
    if K <= 3 and ld <= 1024:
        sig = np.stack([np.maximum(Tw[idx] - 12, 0), Tw[idx]], axis=1).astype(np.int32)
        extra += [dict(sig_range=sig, kappa=1.0, n_samples=2, sigma_signal=np.full(len(idx), 0.3)), dict(want_smooth=True)]
    for kw in extra:
        for fl in ("p1", "p2", "h"):
            os.environ["HMCG_FLAVOUR"] = fl
            print("case", sub, sorted(kw), fl, flush=True)
            r = _lib.estimate_batch_host(*args, want_state=True, window_ids=idx, **kw)
            print("  ok status", r["status"], "L", r["steps_per_thread"], "lds", r["lds_bytes"], flush=True)
