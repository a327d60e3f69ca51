"""ctypes binding of libhmcgibbs.so (C ABI: include/hmcg.h).

The library is the product: hand-written HIP kernels for gfx950.  There is no
CPU fallback anywhere in this package -- if the shared object is missing or no
GPU is usable, the compute entry points raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.path.join(CSRC, os.environ.get("HMCG_LIB", "libhmcgibbs.so"))   # HMCG_LIB: diagnostic builds only

HMCG_MAXH = 8
HMCG_MAXK = 8
FLAG_RESUME = 1

ST_BAD_INVGAMMA, ST_EMIS_UNDERFLOW, ST_NONFINITE, ST_GAMMA_CAP, ST_BAD_T, ST_BAD_RANGE = 1, 2, 4, 8, 16, 32
ST_SKIPPED = ST_NONFINITE | ST_BAD_T | ST_BAD_RANGE      # the window was not computed at all

HMCG_MAXTAIL = 256
EXPORTS = ("hmcg_version", "hmcg_device_count", "hmcg_last_error", "hmcg_shutdown",
           "hmcg_estimate_batch", "hmcg_estimate_batch_device", "hmcg_estimate_batch_multi",
           "hmcg_save_results_csv", "hmcg_write_table_csv", "hmcg_format_float")
HMCG_MAXDEV = 16


class HmcgError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("W", C.c_int32), ("K", C.c_int32), ("ldY", C.c_int32),
                ("max_T", C.c_int32), ("burnin", C.c_int32), ("nrun", C.c_int32), ("H", C.c_int32),
                ("horizons", C.c_int32 * HMCG_MAXH), ("seed", C.c_uint64), ("window_base", C.c_uint32),
                ("device", C.c_int32), ("flags", C.c_int32), ("threads_per_window", C.c_int32),
                ("sweep_base", C.c_int32), ("sweep_count", C.c_int32), ("alpha", C.c_double), ("nu", C.c_double),
                ("kappa", C.c_double), ("n_samples", C.c_int32), ("blend_mask", C.c_int32),
                ("min_T", C.c_int32), ("reserved3", C.c_int32)]


class Extras(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32), ("x_init", C.c_void_p),
                ("x_final", C.c_void_p), ("pif_final", C.c_void_p), ("xstate", C.c_void_p), ("sumacc", C.c_void_p), ("window_ids", C.c_void_p),
                ("sig_range", C.c_void_p), ("save_range", C.c_void_p), ("sigma_signal", C.c_void_p),
                ("sigvals", C.c_void_p), ("nsave_ld", C.c_int32), ("reserved2", C.c_int32),
                ("end_pos", C.c_void_p), ("pi_smooth_mean", C.c_void_p), ("pi_filter_mean", C.c_void_p),
                ("corr", C.c_void_p), ("pi_smooth_draws", C.c_void_p), ("sample_summary", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("launches", C.c_int32), ("threads_per_window", C.c_int32),
                ("steps_per_thread", C.c_int32), ("lds_bytes", C.c_int32), ("helper_waves", C.c_int32),
                ("device", C.c_int32), ("call_ms", C.c_double), ("windows", C.c_int32), ("occupancy", C.c_int32),
                ("buckets", C.c_int32), ("streaming", C.c_int32)]


_LIB = None


def build(force=False):
    """Compile libhmcgibbs.so for gfx950 with hipcc (cross-compiles without a GPU).  make decides what is stale;
    the kernel instantiations are several translation units, compiled in parallel."""
    jobs = str(max(1, min(8, len(os.sched_getaffinity(0)))))
    cmd = ["make", "-C", CSRC, "-j", jobs] + (["-B"] if force else []) + ["libhmcgibbs.so"]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return SO_PATH


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7) next to libtorch_hip.so and resolves it by FILE name, while
    libhmcgibbs.so needs the SONAME.  If ours were loaded first (binding the system
    /opt/rocm runtime), a later `import torch` would bring a second runtime into the process
    and neither could use the other's device pointers or streams.  So when torch is
    installed, its copy is made resident first; libhmcgibbs then binds to it by SONAME, and
    a later `import torch` finds the same file already loaded.  Without torch (C or Julia
    callers) the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared object (no GPU needed for this; compute calls need one)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise HmcgError("libhmcgibbs.so is not built (%s); run __graft_entry__.build() -- "
                            "this package has no CPU fallback" % SO_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(SO_PATH)
        L.hmcg_version.restype = C.c_int
        L.hmcg_device_count.restype = C.c_int
        L.hmcg_last_error.restype = C.c_char_p
        L.hmcg_shutdown.restype = None
        L.hmcg_estimate_batch.restype = C.c_int
        L.hmcg_estimate_batch_device.restype = C.c_int
        L.hmcg_estimate_batch_multi.restype = C.c_int
        L.hmcg_save_results_csv.restype = C.c_int
        L.hmcg_write_table_csv.restype = C.c_int
        L.hmcg_format_float.restype = C.c_int
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        msg = load().hmcg_last_error().decode("utf-8", "replace")
        raise HmcgError("libhmcgibbs rc=%d: %s" % (rc, msg))


def make_config(W, K, ldY, max_T, burnin, nrun, horizons, seed=1234, window_base=0, device=0, flags=0,
                threads_per_window=0, sweep_base=0, alpha=0.0, nu=0.0, sweep_count=0, kappa=0.0, n_samples=0,
                blend_mask=0, min_T=0):
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.W, cfg.K, cfg.ldY, cfg.max_T = int(W), int(K), int(ldY), int(max_T)
    cfg.burnin, cfg.nrun, cfg.H = int(burnin), int(nrun), len(horizons)
    if len(horizons) > HMCG_MAXH:
        raise ValueError("at most %d horizons" % HMCG_MAXH)
    for i, h in enumerate(horizons):
        cfg.horizons[i] = int(h)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.window_base = int(window_base)
    cfg.device, cfg.flags = int(device), int(flags)
    cfg.threads_per_window, cfg.sweep_base = int(threads_per_window), int(sweep_base)
    cfg.sweep_count = int(sweep_count)
    cfg.kappa, cfg.n_samples = float(kappa), int(n_samples)
    cfg.blend_mask = int(blend_mask)
    cfg.min_T = int(min_T)
    cfg.alpha, cfg.nu = float(alpha), float(nu)
    return cfg


def _np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def estimate_batch_host(Y, T, K, burnin, nrun, horizons=(12,), yreal=None, seed=1234, window_base=0, device=0,
                        threads_per_window=0, x_init=None, want_state=False, want_draws=True, alpha=0.0, nu=0.0,
                        resume_state=None, sweep_base=0, window_ids=None, sweep_count=0,
                        sig_range=None, save_range=None, sigma_signal=None, kappa=0.0, n_samples=0, want_smooth=False,
                        end_pos=None, blend_mask=0, want_filter_mean=False, devices=None, out=None, want_corr=False,
                        want_sample_summary=False, resume_sample_summary=None, nan_fill=True, want_smooth_draws=False):
    """hmcg_estimate_batch over host (numpy) buffers.  Returns dict of arrays in the
    C-ABI layouts (window slowest): mu/sig2/pi_end (W,K,nrun), A (W,K,K,nrun) with
    A[w, j, i, d] = draw d of A[i,j], fcast (W,2H,nrun), summary (W,NS), status (W,).
    devices: a list of HIP ordinals -> hmcg_estimate_batch_multi (windows partitioned over those GPUs).
    out: a dict returned by an earlier call of the same shape -- its arrays are reused (a caller that owns its
    buffers, as a Julia or C caller does, pays no allocation or first-touch page faults per call).
    want_corr: out["corr"] (W, NC, NC), NC = 3K + K^2 + 1 -- the correlation matrix calccorr (src/Hmc.jl:1094-1163) builds
    per end date from the per-draw CSV files, accumulated on the device from the rounded draws (extras.corr); works with
    want_draws=False (the draws then never leave the device).
    want_sample_summary (signal path): out["sample_summary"] (W, n_samples, NS) -- per noise sample the mean over its kept
    draws of the 5-digit-rounded outputs (extras.sample_summary: the rows runaggregate makes per (date, signalid));
    resume_sample_summary carries the buffer of a call that stopped inside a sample into its RESUME call."""
    L = load()
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    W, ldY = Y.shape
    T = np.ascontiguousarray(T, dtype=np.int32)
    H = len(horizons)
    NS = 3 * K + K * K + 2 * H
    yr = None if yreal is None else np.ascontiguousarray(yreal, dtype=np.float64).reshape(W, H)
    prev = out if out is not None else {}
    out = {}
    nd = max(int(n_samples), 1) * nrun          # kept draws per window (sample-major on the signal path)
    shapes = dict(mu=(W, K, nd), sig2=(W, K, nd), A=(W, K, K, nd), pi_end=(W, K, nd), fcast=(W, 2 * H, nd))
    keep = tuple(shapes) if want_draws is True else (tuple(want_draws) if want_draws else ())   # True, False or names

    def buf(name, shape, dtype=np.float64):
        a = prev.get(name)
        return a if isinstance(a, np.ndarray) and a.shape == tuple(shape) and a.dtype == dtype and a.flags.c_contiguous else np.zeros(shape, dtype=dtype)
    for name in keep:
        out[name] = buf(name, shapes[name])
    out["summary"] = buf("summary", (W, NS))
    out["status"] = buf("status", (W,), np.int32)
    if resume_state is None:
        out["status"][:] = 0
    ex = Extras()
    ex.struct_size = C.sizeof(Extras)
    flags = 0
    if x_init is not None:
        xi = np.ascontiguousarray(x_init, dtype=np.int32).reshape(W, ldY)
        ex.x_init = xi.ctypes.data
    if window_ids is not None:
        wid = np.ascontiguousarray(window_ids, dtype=np.uint32).reshape(W)
        ex.window_ids = wid.ctypes.data
    if sig_range is not None:                      # signal Monte-Carlo path (estimatesignals!)
        sr = np.ascontiguousarray(sig_range, dtype=np.int32).reshape(W, 2)
        ex.sig_range = sr.ctypes.data
        if save_range is not None:
            svr = np.ascontiguousarray(save_range, dtype=np.int32).reshape(W, 2)
            ex.save_range = svr.ctypes.data
            nsave = int(max(1, (svr[:, 1] - svr[:, 0]).max()))
            out["sigvals"] = np.zeros((W, max(int(n_samples), 1), nsave))
            ex.sigvals = out["sigvals"].ctypes.data
            ex.nsave_ld = nsave
        if sigma_signal is not None:
            ssg = np.ascontiguousarray(sigma_signal, dtype=np.float64).reshape(W)
            ex.sigma_signal = ssg.ctypes.data
        if end_pos is not None:                    # signals past the end date (sigLen > 0)
            epos = np.ascontiguousarray(end_pos, dtype=np.int32).reshape(W)
            ex.end_pos = epos.ctypes.data
    if want_sample_summary:
        out["sample_summary"] = (np.zeros((W, max(int(n_samples), 1), NS)) if resume_sample_summary is None else
                                 np.ascontiguousarray(resume_sample_summary, dtype=np.float64).reshape(W, max(int(n_samples), 1), NS).copy())
        ex.sample_summary = out["sample_summary"].ctypes.data
    if want_smooth_draws:                      # samples.pib[Nrun, N, D] of every window: (W, K, ldY, nd), draw index fastest
        out["pi_smooth_draws"] = np.zeros((W, K, ldY, nd))
        ex.pi_smooth_draws = out["pi_smooth_draws"].ctypes.data
    if want_smooth:
        out["pi_smooth_mean"] = np.zeros((W, ldY, K))
        ex.pi_smooth_mean = out["pi_smooth_mean"].ctypes.data
    if want_filter_mean:
        out["pi_filter_mean"] = np.zeros((W, ldY, K))
        ex.pi_filter_mean = out["pi_filter_mean"].ctypes.data
    if want_corr:
        NC = 3 * K + K * K + 1
        out["corr"] = np.zeros((W, NC, NC))
        ex.corr = out["corr"].ctypes.data
    if want_state:
        out["x_final"] = np.zeros((W, ldY), dtype=np.int32)
        out["pif_final"] = np.zeros((W, ldY, K))
        out["xstate"] = np.zeros((W, ldY), dtype=np.uint8)
        out["sumacc"] = np.zeros((W, NS + K))
        ex.x_final = out["x_final"].ctypes.data
        ex.pif_final = out["pif_final"].ctypes.data
        ex.xstate = out["xstate"].ctypes.data
        ex.sumacc = out["sumacc"].ctypes.data
    if resume_state is not None:
        flags |= FLAG_RESUME
        out["xstate"] = np.ascontiguousarray(resume_state["xstate"], dtype=np.uint8).copy()
        out["sumacc"] = np.ascontiguousarray(resume_state["sumacc"], dtype=np.float64).copy()
        out["status"] = np.ascontiguousarray(resume_state["status"], dtype=np.int32).copy()
        ex.xstate = out["xstate"].ctypes.data
        ex.sumacc = out["sumacc"].ctypes.data
    cfg = make_config(W, K, ldY, min(int(T.max()), ldY), burnin, nrun, horizons, seed, window_base, device, flags,
                      threads_per_window, sweep_base, alpha, nu, sweep_count, kappa, n_samples, blend_mask)
    if devices is None:
        tm = Timing()
        rc = L.hmcg_estimate_batch(C.byref(cfg), _np_ptr(Y), _np_ptr(T), _np_ptr(yr),
                                   _np_ptr(out.get("mu")), _np_ptr(out.get("sig2")), _np_ptr(out.get("A")),
                                   _np_ptr(out.get("pi_end")), _np_ptr(out.get("fcast")), _np_ptr(out["summary"]),
                                   _np_ptr(out["status"]), C.byref(ex), C.byref(tm))
        tms = [tm]
    else:
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        tms = (Timing * len(devices))()
        rc = L.hmcg_estimate_batch_multi(C.byref(cfg), C.c_int32(len(devices)), devs, _np_ptr(Y), _np_ptr(T), _np_ptr(yr),
                                         _np_ptr(out.get("mu")), _np_ptr(out.get("sig2")), _np_ptr(out.get("A")),
                                         _np_ptr(out.get("pi_end")), _np_ptr(out.get("fcast")), _np_ptr(out["summary"]),
                                         _np_ptr(out["status"]), C.byref(ex), tms)
        tm = tms[0]
    _check(rc)
    # a skipped window (non-finite data, bad T, bad ranges) was not computed: its outputs read NaN, never a
    # plausible-looking zero (the reference would have thrown, src/Hmc.jl:435)
    skipped = (out["status"] & ST_SKIPPED) != 0
    if skipped.any() and nan_fill:
        for name in keep + ("summary", "sigvals", "pi_smooth_mean", "pi_filter_mean", "pif_final", "corr", "sample_summary", "pi_smooth_draws"):
            if name in out:
                out[name][skipped] = np.nan
    out["kernel_ms"] = tm.kernel_ms
    out["threads_per_window"] = tm.threads_per_window
    out["steps_per_thread"] = tm.steps_per_thread
    out["lds_bytes"] = tm.lds_bytes
    out["helper_waves"] = tm.helper_waves
    out["launches"] = tm.launches
    out["buckets"] = tm.buckets
    out["streaming"] = bool(tm.streaming)
    out["call_ms"] = max(t.call_ms for t in tms)
    out["per_device"] = [dict(device=t.device, windows=t.windows, kernel_ms=t.kernel_ms, call_ms=t.call_ms, launches=t.launches)
                         for t in tms]
    return out


def estimate_batch_device(cfg, dY, dT, dyreal, dmu, dsig2, dA, dpi_end, dfcast, dsummary, dstatus,
                          extras=None, stream=None, timed=True):
    """hmcg_estimate_batch_device over raw device pointers (ints, e.g. torch
    tensor.data_ptr()).  Returns the Timing struct when timed, else None."""
    L = load()
    tm = Timing() if timed else None

    def vp(x):
        return None if not x else C.c_void_p(int(x))

    rc = L.hmcg_estimate_batch_device(C.byref(cfg), vp(dY), vp(dT), vp(dyreal), vp(dmu), vp(dsig2), vp(dA),
                                      vp(dpi_end), vp(dfcast), vp(dsummary), vp(dstatus),
                                      C.byref(extras) if extras is not None else None, vp(stream),
                                      C.byref(tm) if timed else None)
    _check(rc)
    return tm


def format_float(x):
    """CSV.jl 0.5.16 text of one Float64 (hmcg_format_float)."""
    buf = C.create_string_buffer(48)
    n = load().hmcg_format_float(C.c_double(float(x)), buf)
    return buf.raw[:n].decode()


def save_results_csv(dir, dates, K, horizons, res, sigvals=None, nsave=0, legacy_trans_header=False, n_threads=0):
    """hmcg_save_results_csv: the five per-window CSV files of saveresults (src/Hmc.jl:724-748) for every window of a
    result dict of estimate_batch_host (C-ABI layouts), written by the library's native writer."""
    W = len(dates)
    H = len(horizons)
    os.makedirs(dir, exist_ok=True)

    def arr(name, shape):
        a = res.get(name)
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == shape, (name, a.shape, shape)
        return a
    nd = next(res[k].shape[-1] for k in ("mu", "sig2", "pi_end", "A", "fcast") if res.get(k) is not None)
    mu, sig2, pe = arr("mu", (W, K, nd)), arr("sig2", (W, K, nd)), arr("pi_end", (W, K, nd))
    A, fc = arr("A", (W, K, K, nd)), arr("fcast", (W, 2 * H, nd))
    dts = (C.c_char_p * W)(*[str(d).encode() for d in dates])
    hz = (C.c_int32 * max(H, 1))(*[int(h) for h in horizons])
    sv = None
    n_samples = nsave_ld = 0
    if sigvals is not None:
        sv = np.ascontiguousarray(sigvals, dtype=np.float64)
        _, n_samples, nsave_ld = sv.shape
    rc = load().hmcg_save_results_csv(str(dir).encode(), C.c_int32(W), dts, C.c_int32(K), C.c_int32(H), hz, C.c_int64(nd),
                                      _np_ptr(mu), _np_ptr(sig2), _np_ptr(pe), _np_ptr(A), _np_ptr(fc), _np_ptr(sv),
                                      C.c_int32(n_samples), C.c_int32(nsave), C.c_int32(nsave_ld),
                                      C.c_int32(1 if legacy_trans_header else 0), C.c_int32(n_threads))
    if rc != 0:
        raise HmcgError("hmcg_save_results_csv rc=%d (directory %s)" % (rc, dir))
