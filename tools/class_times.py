"""Kernel time of each length class of the production shape by itself, per flavour (what a block of that class costs alone /
paired), and of two-class mixes.  Usage: python tools/class_times.py [draws]"""
import os
import sys
os.environ.setdefault("HMCG_DIAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmc_jl_amd  # noqa: F401
from hmc_jl_amd import device as hdev, synth

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def run(label, lens, flav, nt=0):
    if flav:
        os.environ["HMCG_BUCKET_FLAVOURS"] = flav
        os.environ["HMCG_FLAVOUR"] = flav.split(",")[0]
    else:
        os.environ.pop("HMCG_BUCKET_FLAVOURS", None); os.environ.pop("HMCG_FLAVOUR", None)
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    panel = hdev.DevicePanel(Y, Tw, 3, draws, (12,), fut[:, 11:12], keep_draws=True)
    panel.run(burnin=0, threads_per_window=nt)
    ms = [panel.run(burnin=0, threads_per_window=nt) for _ in range(3)]
    tm = panel.last_timing
    print("%-34s W=%4d %-9s %7.3f ms  L=%d NT=%d nh=%d occ=%d buckets=%d" % (label, len(lens), flav, min(ms), tm.steps_per_thread, tm.threads_per_window, tm.helper_waves, tm.occupancy, tm.buckets), flush=True)


top, mid, low = list(range(513, 580)), list(range(257, 513)), list(range(120, 257))
for fl in ("h", "p1", "p2"):
    run("top class alone (513..579)", top, fl)
    run("mid class alone (257..512)", mid, fl)
    run("low class alone (120..256)", low, fl)
run("low class NT=128", low, "p2", nt=128)
run("mid class NT=128", mid, "p2", nt=128)
for fl in ("p2,p2", "h,p2", "h,h", "p1,p1"):
    run("top+mid", top + mid, fl)
    run("top+low", top + low, fl)
    run("mid+low", mid + low, fl)
run("top x2 (134 windows)", top + top, "h")
run("top x4 (268 windows)", top * 4, "p2")
run("top x7 (469 windows)", top * 7, "p2")
