"""Host-side timeline of hmcg_estimate_batch at the headline shape (HMCG_TRACE=1: the library prints where the wall time of
a call goes), through the plain-C driver so that no Python sits in the process.  usage: python tools/trace_host_entry.py"""
import os, struct, subprocess, sys, tempfile
os.environ.setdefault("HMCG_DIAG", "1")      # arms the library's diagnostic switches (read once at first use)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, T, K, DRAWS = 256, 1000, 3, 1000
Y, Tw, fut = synth.generate_panel(W, T, K)
yreal = fut[:, 11:12]
hd = [0x484d4347, 3, W, K, T, 0, DRAWS, 1, 12, 0, 0, 0, 0, 0, 0, 0, 0, 0, int(os.environ.get("TRACE_CALLS", "8")), 0]
with tempfile.TemporaryDirectory() as tmp:
    req = os.path.join(tmp, "req.bin")
    with open(req, "wb") as f:
        f.write(struct.pack("<20i", *hd)); f.write(struct.pack("<3d", 0.0, 0.0, 0.0))
        f.write(np.ascontiguousarray(Y, dtype="<f8").tobytes()); f.write(np.ascontiguousarray(Tw, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(yreal, dtype="<f8").tobytes())
    env = dict(os.environ); env.setdefault("HMCG_TRACE", "1"); env.pop("LD_PRELOAD", None)      # HMCG_TRACE= (empty) still traces; unset it in the library by not exporting
    r = subprocess.run([os.path.join(ROOT, "tests", "cdriver", "hmcg_cdriver"), req, os.path.join(tmp, "resp.bin")], capture_output=True, text=True, env=env)
print(r.stdout)
print("\n".join(r.stderr.splitlines()[-5:]))
