"""Cross-check of the assembly filter replay (tools/gen_replay_asm.py) against the C++ loop it replaces, on the GPU:
`make -C hmc.jl_amd/csrc abunit NAME=chk EXTRA=-DHMCG_REPLAY_CHECK=1 LINT=0` builds a library whose K = 8 kernel runs the
assembly loop first and then the C++ loop from the same state, and reports any difference in the window's status word:
0x100 filtered vector, 0x200 flags, 0x400 state maps, 0x800 pif[T-1,:].  (Found with it in round 4: a v_rcp_f64 result read
by the next instruction -- the gfx940 transcendental-use hazard the compiler pads by itself and an asm statement must pad
by hand.)  usage (on the GPU box): python tools/replay_check.py"""
import os, sys
os.environ["HMCG_DIAG"] = "1"; os.environ["HMCG_LIB"] = "libhmcgibbs_chk.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmc_jl_amd import _lib, synth
bad = 0
for T, W in ((200, 2), (700, 3), (5000, 2), (257, 1), (64, 1), (3, 1), (1281, 2)):
    Y, Tw, fut = synth.generate_panel(W, T, 8)
    g = _lib.estimate_batch_host(Y, Tw, 8, 1, 4, (12,), fut[:, 11:12], nan_fill=False)
    print("T=%d W=%d status" % (T, W), [hex(int(x)) for x in g["status"]], "L=", g["steps_per_thread"], flush=True)
    bad += int((g["status"] & 0xF00).any())
sys.exit(1 if bad else 0)
