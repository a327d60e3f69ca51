"""The C ABI driven from plain C (tests/cdriver/hmcg_cdriver.c): no Python and no torch in the computing process, the
library binds the system ROCm runtime -- the closest executable stand-in for the Julia `ccall` path INTEGRATION.md
documents (reference callers: code/run_hmm.jl:95-120).  The wrapper writes a request file, runs the driver as a child
process and compares what it wrote with the oracle: states are not returned on this path, so parity is on the
per-draw floats (1e-9) -- a wrong state path would show there within a sweep."""
import os
import struct
import subprocess

import numpy as np
import pytest

from hmc_jl_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV_DIR = os.path.join(ROOT, "tests", "cdriver")
DRV = os.path.join(DRV_DIR, "hmcg_cdriver")
TOL = 1e-9


def build_driver():
    src = os.path.join(DRV_DIR, "hmcg_cdriver.c")
    lib = os.path.join(ROOT, "hmc.jl_amd", "csrc")
    if not os.path.exists(DRV) or os.path.getmtime(DRV) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "hmcg.h"))):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", DRV, src, "-L" + lib, "-lhmcgibbs",
                               "-Wl,-rpath,$ORIGIN/../../hmc.jl_amd/csrc", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return DRV


def run_driver(tmp_path, mode, Y, Tw, K, burnin, nrun, horizons, yreal, n_samples=0, kappa=0.0, alpha=0.0, nu=0.0,
               sig=None, save=None, ssig=None, nsave_ld=0, n_devices=1, env_extra=None, expect_rc=0):
    W, ld = Y.shape
    H = len(horizons)
    hz = list(horizons) + [0] * (8 - H)
    hd = [0x484d4347, mode, W, K, ld, burnin, nrun, H] + hz + [n_samples, nsave_ld, n_devices, 0]
    req, resp = str(tmp_path / "req.bin"), str(tmp_path / "resp.bin")
    with open(req, "wb") as f:
        f.write(struct.pack("<20i", *hd))
        f.write(struct.pack("<3d", kappa, alpha, nu))
        f.write(np.ascontiguousarray(Y, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(Tw, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(yreal, dtype="<f8").tobytes())
        if mode == 1:
            f.write(np.ascontiguousarray(sig, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(save, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(ssig, dtype="<f8").tobytes())
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    env.update(env_extra or {})
    r = subprocess.run([build_driver(), req, resp], capture_output=True, text=True, env=env, timeout=300)
    if expect_rc:
        assert r.returncode == expect_rc, (r.returncode, r.stdout, r.stderr)
        return None, r.stdout + r.stderr
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "cdriver ok" in r.stdout
    nd = max(n_samples, 1) * nrun
    NS = 3 * K + K * K + 2 * H
    raw = np.fromfile(resp, dtype=np.uint8)
    out, off = {}, 0
    for name, shape, dt in (("mu", (W, K, nd), "<f8"), ("sig2", (W, K, nd), "<f8"), ("A", (W, K, K, nd), "<f8"),
                            ("pi_end", (W, K, nd), "<f8"), ("fcast", (W, 2 * H, nd), "<f8"), ("summary", (W, NS), "<f8"),
                            ("status", (W,), "<i4")) + ((("sigvals", (W, max(n_samples, 1), nsave_ld), "<f8"),) if mode == 1 else ()) \
            + ((("corr", (W, 3 * K + K * K + 1, 3 * K + K * K + 1), "<f8"),) if mode == 4 else ()):
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[name] = raw[off:off + n].view(dt).reshape(shape)
        off += n
    assert off == raw.size
    return out, r.stdout


def close(g, o):
    return float(np.max(np.abs(g - o) / (1.0 + np.abs(o)))) if g.size else 0.0


def check(g, o, w):
    assert g["status"][w] == o["status"] == 0
    assert close(g["mu"][w].T, o["mu"]) < TOL and close(g["sig2"][w].T, o["sig2"]) < TOL
    assert close(np.transpose(g["A"][w], (2, 1, 0)), o["A"]) < TOL and close(g["pi_end"][w].T, o["pi_end"]) < TOL
    assert close(g["fcast"][w].T, o["fcast"]) < TOL and close(g["summary"][w], o["summary"]) < TOL


def test_c_driver_cfg1_estimatemodel(hmclib, oracle, tmp_path):
    """BASELINE configs[0] (3-state, T=200, 1 window, 100 draws) through hmcg_estimate_batch from C."""
    Y, Tw, fut = synth.generate_panel(1, 200, 3)
    g, log = run_driver(tmp_path, 0, Y, Tw, 3, 0, 100, (12,), fut[:, 11:12])
    check(g, oracle.estimate_window(Y[0], 3, 0, 100, (12,), fut[0, 11:12], window_id=0), 0)
    assert " 3 launches" in log or " 4 launches" in log or " 2 launches" in log      # the chain ran in chunks


def test_c_driver_signal_call(hmclib, oracle, tmp_path):
    """One estimatesignals!-shaped call (3 windows, 3 chained noise samples) from C."""
    K, T = 3, 300
    Y, Tw, fut = synth.generate_panel(3, T, K)
    sig = np.array([[T - 40, T], [T - 1, T], [T // 2, T]], dtype=np.int32)
    save = np.array([[T - 3, T], [T - 1, T], [T - 2, T]], dtype=np.int32)
    ssig = np.array([0.5, 1.0, 0.2])
    g, _ = run_driver(tmp_path, 1, Y, Tw, K, 3, 8, (12,), fut[:, 11:12], n_samples=3, kappa=0.6, alpha=2.0, nu=2.0,
                      sig=sig, save=save, ssig=ssig, nsave_ld=3)
    for w in range(3):
        o = oracle.estimate_signals(Y[w], K, 3, 8, 3, sig=tuple(sig[w]), kappa=0.6, alpha=2.0, nu=2.0, sigma_signal=float(ssig[w]),
                                    save=tuple(save[w]), yreal=fut[w, 11:12], window_id=w)
        check(g, o, w)
        ns = save[w][1] - save[w][0]
        assert close(g["sigvals"][w][:, :ns], o["sigvals"]) < TOL


def test_c_driver_multi_device_entry_one_device(hmclib, oracle, tmp_path):
    """hmcg_estimate_batch_multi with n_devices = 1 (all this box has): ragged windows, global RNG ids."""
    lens = [500, 120, 333, 64, 257]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    g, _ = run_driver(tmp_path, 2, Y, Tw, 3, 4, 20, (1, 12), fut[:, [0, 11]], n_devices=1)
    for w in range(len(lens)):
        check(g, oracle.estimate_window(Y[w, :Tw[w]], 3, 4, 20, (1, 12), fut[w, [0, 11]], window_id=w), w)


def test_multi_entry_four_virtual_devices_equals_single_device(hmclib, oracle, tmp_path):
    """The G > 1 branch of hmcg_estimate_batch_multi -- worker threads, one context (streams, arenas, scatter helpers) per
    device id, LPT partition, error aggregation -- on a one-GPU box: HMCG_VIRTUAL_DEVICES=4 maps four device ids onto the
    one physical device.  Ragged windows, enough draws that every worker's chunks go through its scatter helpers; the
    result must equal the single-device call bit for bit (global RNG ids) and the oracle within 1e-9.
    Reference role: the SLURM fan-out, slurmscripts/base_estimation.sh:5,17."""
    lens = [500, 120, 333, 64, 257, 480, 199, 401, 77, 350, 512]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    one, _ = run_driver(tmp_path, 2, Y, Tw, 3, 4, 3000, (1, 12), fut[:, [0, 11]], n_devices=1)
    four, log = run_driver(tmp_path, 2, Y, Tw, 3, 4, 3000, (1, 12), fut[:, [0, 11]], n_devices=4,
                           env_extra={"HMCG_VIRTUAL_DEVICES": "4", "HMCG_CHUNK_DRAWS": "1500"})
    for k in ("mu", "sig2", "A", "pi_end", "fcast", "summary", "status"):
        assert np.array_equal(one[k], four[k]), k
    for w in (0, 3, 10):
        check(four, oracle.estimate_window(Y[w, :Tw[w]], 3, 4, 3000, (1, 12), fut[w, [0, 11]], window_id=w), w)
    # more devices than the box offers without the switch: refused, not silently folded
    _, msg = run_driver(tmp_path, 2, Y, Tw, 3, 4, 20, (1, 12), fut[:, [0, 11]], n_devices=4, expect_rc=5)
    assert "out of range" in msg


def test_multi_entry_reports_the_failing_device(hmclib, tmp_path):
    """One worker of four fails (HMCG_FAIL_DEVICE: injected before it touches the GPU): the call returns that device's
    error, named, after the other workers have finished."""
    lens = [300, 200, 250, 150, 280, 120, 90, 310]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    _, msg = run_driver(tmp_path, 2, Y, Tw, 3, 2, 50, (12,), fut[:, 11:12], n_devices=4, expect_rc=5,
                        env_extra={"HMCG_VIRTUAL_DEVICES": "4", "HMCG_FAIL_DEVICE": "2"})
    assert "device 2: injected failure" in msg, msg


def test_c_driver_correlation_matrices(tmp_path):
    """extras.corr from plain C (mode 4): the matrices calccorr builds (src/Hmc.jl:1094-1163), taken on the device, against
    numpy.corrcoef of the 5-digit-rounded draws the same call returned."""
    K, W, T, nrun = 3, 3, 300, 900
    Y, Tw, fut = synth.generate_panel(W, T, K)
    out, _ = run_driver(tmp_path, 4, Y, Tw, K, 30, nrun, (12,), fut[:, 11:12])
    assert (out["status"] == 0).all()
    for w in range(W):
        cols = np.concatenate([out["mu"][w], out["sig2"][w], out["pi_end"][w], out["A"][w].reshape(K * K, nrun), out["fcast"][w][:1]])
        with np.errstate(invalid="ignore", divide="ignore"):
            exp = np.corrcoef(np.round(cols, 5))
        ok = np.isfinite(exp)
        assert np.array_equal(np.isfinite(out["corr"][w]), ok) and np.abs(out["corr"][w][ok] - exp[ok]).max() < 1e-10
