"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (hmc.jl_amd/) never does.

See oracle/hmc_oracle.c for the reference citations (src/Hmc.jl file:line) and
the parity status ("draw-level parity with Julia: parity unpinned"; pinned
statistically by tests/golden/official_*_summary.csv).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hmc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.hmco_estimate_window.restype = C.c_int
        L.hmco_estimate_window_ex.restype = C.c_int
        L.hmco_estimate_batch.restype = C.c_int
        L.hmco_forward_filter.restype = C.c_int
        L.hmco_backward_smoother.restype = None
        L.hmco_forecast.restype = C.c_double
        L.hmco_gamma.restype = C.c_double
        L.hmco_normal.restype = C.c_double
        L.hmco_uniform_x.restype = C.c_double
        L.hmco_max_threads.restype = C.c_int
        L.hmco_sample_summary.restype = None
        _LIB = L
    return _LIB


def _p(a, ty=_dp):
    return None if a is None else a.ctypes.data_as(ty)


def estimate_window(Y, K, burnin, nrun, horizons=(12,), yreal=None, seed=1234, window_id=0,
                    faithful_cost=False, smoother=False, x_init=None, want_smooth=False):
    """gibbssample!/estimatemodel for one window (src/Hmc.jl:517-562, 850-865).

    Returns a dict of Julia-layout arrays converted to numpy (draw index first):
    mu (nrun,K), sig2 (nrun,K), A (nrun,K,K), pi_end (nrun,K), fcast (nrun,2H),
    summary (3K+K^2+2H), x_final (T,) 0-based, pif_final (T,K) unsorted, status.
    """
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    T = Y.shape[0]
    H = len(horizons)
    hz = np.asarray(horizons, dtype=np.int32)
    yr = np.full(H, np.nan) if yreal is None else np.ascontiguousarray(yreal, dtype=np.float64)
    mu = np.empty((K, nrun)); sig2 = np.empty((K, nrun)); A = np.empty((K, K, nrun))
    pe = np.empty((K, nrun)); fc = np.empty((2 * H, nrun))
    summ = np.empty(3 * K + K * K + 2 * H)
    xf = np.empty(T, dtype=np.int32); pf = np.empty((T, K))
    sm = np.empty((K, T, nrun)) if want_smooth else None
    xi = None if x_init is None else np.ascontiguousarray(x_init, dtype=np.int32)
    st = C.c_int(0)
    flags = (1 if faithful_cost else 0) | (2 if smoother else 0)
    rc = lib().hmco_estimate_window(_p(Y), C.c_int(T), C.c_int(K), C.c_int(burnin), C.c_int(nrun),
                                    _p(hz, _ip), C.c_int(H), _p(yr), C.c_uint64(seed), C.c_uint32(window_id),
                                    C.c_int(flags), _p(xi, _ip), _p(mu), _p(sig2), _p(A), _p(pe), _p(fc),
                                    _p(sm), _p(summ), _p(xf, _ip), _p(pf), C.byref(st))
    if rc != 0:
        raise ValueError("hmco_estimate_window rc=%d" % rc)
    out = dict(mu=mu.T.copy(), sig2=sig2.T.copy(), A=A.transpose(2, 1, 0).copy(), pi_end=pe.T.copy(),
               fcast=fc.T.copy(), summary=summ, x_final=xf, pif_final=pf, status=st.value)
    if want_smooth:
        out["pi_smooth"] = sm.transpose(2, 1, 0).copy()  # (nrun, T, K)
    return out


def estimate_signals(Y, K, burnin, nrun, n_samples=1, sig=(0, 0), kappa=1.0, alpha=1.0, nu=1.0, sigma_signal=0.0,
                     save=(0, 0), horizons=(12,), yreal=None, seed=1234, window_id=0, x_init=None, end_pos=-1,
                     blend_mask=0, want_filter_mean=False, want_smooth=False):
    """estimatesignals!'s sampling loop (src/Hmc.jl:868-914) for one window; with n_samples=1 and
    sigma_signal=0 it is the base estimatemodel run on a window that has a signal set.
    sig/save are 0-based half-open position ranges.  end_pos (0-based) selects the position whose smoothed
    probabilities are reported as pi_end (:900; default the last step), blend_mask the horizons reported through
    forecastsignal (:908-909).  Returns draws as (n_samples*nrun, ...) arrays plus sigvals (n_samples, nsave)."""
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    T = Y.shape[0]
    H = len(horizons)
    hz = np.asarray(horizons, dtype=np.int32)
    yr = np.full(H, np.nan) if yreal is None else np.ascontiguousarray(yreal, dtype=np.float64)
    nd = n_samples * nrun
    mu = np.empty((K, nd)); sig2 = np.empty((K, nd)); A = np.empty((K, K, nd))
    pe = np.empty((K, nd)); fc = np.empty((2 * H, nd))
    summ = np.empty(3 * K + K * K + 2 * H)
    nsave = max(save[1] - save[0], 0)
    sv = np.zeros((n_samples, max(nsave, 1)))
    xf = np.empty(T, dtype=np.int32); pf = np.empty((T, K))
    fm = np.zeros((T, K)) if want_filter_mean else None
    sm = np.empty((K, T, nd)) if want_smooth else None
    xi = None if x_init is None else np.ascontiguousarray(x_init, dtype=np.int32)
    st = C.c_int(0)
    rc = lib().hmco_estimate_window_ex(_p(Y), C.c_int(T), C.c_int(K), C.c_int(burnin), C.c_int(nrun),
                                       _p(hz, _ip), C.c_int(H), _p(yr), C.c_uint64(seed), C.c_uint32(window_id),
                                       C.c_int(0), _p(xi, _ip), C.c_int(sig[0]), C.c_int(sig[1]), C.c_double(kappa),
                                       C.c_double(alpha), C.c_double(nu), C.c_int(n_samples), C.c_double(sigma_signal),
                                       C.c_int(save[0]), C.c_int(save[1]), C.c_int(end_pos), C.c_int(blend_mask),
                                       _p(mu), _p(sig2), _p(A), _p(pe), _p(fc),
                                       _p(sm), _p(summ), _p(sv), _p(xf, _ip), _p(pf), _p(fm), C.byref(st))
    if rc != 0:
        raise ValueError("hmco_estimate_window_ex rc=%d" % rc)
    out = dict(mu=mu.T.copy(), sig2=sig2.T.copy(), A=A.transpose(2, 1, 0).copy(), pi_end=pe.T.copy(),
               fcast=fc.T.copy(), summary=summ, sigvals=sv[:, :nsave], x_final=xf, pif_final=pf, status=st.value)
    # runaggregate's (date, signalid) rows of this run (src/Hmc.jl:1053-1075): per noise sample, means of the rounded draws
    ss = np.empty((n_samples, 3 * K + K * K + 2 * H))
    lib().hmco_sample_summary(_p(mu), _p(sig2), _p(A), _p(pe), _p(fc), C.c_int(K), C.c_int(H), C.c_int(n_samples), C.c_int(nrun), _p(ss))
    out["sample_summary"] = ss
    if want_filter_mean:
        out["pi_filter_mean"] = fm               # (T, K) sorted labels, mean over the kept draws
    if want_smooth:
        out["pi_smooth"] = sm.transpose(2, 1, 0).copy()      # (nd, T, K)
    return out


def estimate_batch(Y, T, K, burnin, nrun, horizons=(12,), yreal=None, seed=1234, window_base=0,
                   faithful_cost=False, nthreads=0, want_state=False):
    """Batched oracle, same buffer layouts as include/hmcg.h (window slowest)."""
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    W, ldY = Y.shape
    T = np.ascontiguousarray(T, dtype=np.int32)
    H = len(horizons)
    hz = np.asarray(horizons, dtype=np.int32)
    yr = np.full((W, H), np.nan) if yreal is None else np.ascontiguousarray(yreal, dtype=np.float64)
    mu = np.empty((W, K, nrun)); sig2 = np.empty((W, K, nrun)); A = np.empty((W, K, K, nrun))
    pe = np.empty((W, K, nrun)); fc = np.empty((W, 2 * H, nrun))
    summ = np.empty((W, 3 * K + K * K + 2 * H))
    st = np.zeros(W, dtype=np.int32)
    xf = np.zeros((W, ldY), dtype=np.int32) if want_state else None
    pf = np.zeros((W, ldY, K)) if want_state else None
    rc = lib().hmco_estimate_batch(_p(Y), C.c_int(ldY), _p(T, _ip), C.c_int(W), C.c_int(K), C.c_int(burnin),
                                   C.c_int(nrun), _p(hz, _ip), C.c_int(H), _p(yr), C.c_uint64(seed),
                                   C.c_uint32(window_base), C.c_int(1 if faithful_cost else 0), C.c_int(nthreads),
                                   _p(mu), _p(sig2), _p(A), _p(pe), _p(fc), _p(summ), _p(xf, _ip), _p(pf), _p(st, _ip))
    if rc != 0:
        raise ValueError("hmco_estimate_batch rc=%d" % rc)
    return dict(mu=mu, sig2=sig2, A=A, pi_end=pe, fcast=fc, summary=summ, status=st, x_final=xf, pif_final=pf)


def forward_filter(Y, mu, sig2, rho, A, want_P=False):
    """forwardupdate_P! (src/Hmc.jl:371-440): returns pif (T,K) [, Pf (T,K,K)], status."""
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    K = len(mu)
    T = Y.shape[0]
    pif = np.empty((T, K))
    Pf = np.empty((T, K, K)) if want_P else None
    st = lib().hmco_forward_filter(_p(Y), C.c_int(T), C.c_int(K), _p(np.ascontiguousarray(mu, dtype=np.float64)),
                                   _p(np.ascontiguousarray(sig2, dtype=np.float64)),
                                   _p(np.ascontiguousarray(rho, dtype=np.float64)),
                                   _p(np.ascontiguousarray(A, dtype=np.float64)), _p(pif), _p(Pf))
    return (pif, Pf, st) if want_P else (pif, st)


def backward_smoother(pif, Pf):
    """backwardupdate_P! (src/Hmc.jl:442-457): returns pib (T,K)."""
    T, K = pif.shape
    pib = np.zeros((T, K))
    lib().hmco_backward_smoother(C.c_int(T), C.c_int(K), _p(np.ascontiguousarray(pif)), _p(np.ascontiguousarray(Pf)), _p(pib))
    return pib


def forecast(mu, A, pi_end, h):
    """forecast (src/Hmc.jl:658-667), value only."""
    K = len(mu)
    return lib().hmco_forecast(C.c_int(K), _p(np.ascontiguousarray(mu, dtype=np.float64)),
                               _p(np.ascontiguousarray(A, dtype=np.float64)),
                               _p(np.ascontiguousarray(pi_end, dtype=np.float64)), C.c_int(h))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    lib().hmco_philox(c, C.c_uint32(key[0]), C.c_uint32(key[1]))
    return [int(x) for x in c]


def gamma(seed, window, sweep, site, elem, shape):
    return lib().hmco_gamma(C.c_uint64(seed), C.c_uint32(window), C.c_uint32(sweep), C.c_uint32(site),
                            C.c_uint32(elem), C.c_double(shape))


def normal(seed, window, sweep, site, elem):
    return lib().hmco_normal(C.c_uint64(seed), C.c_uint32(window), C.c_uint32(sweep), C.c_uint32(site), C.c_uint32(elem))


def uniform_x(seed, window, sweep, t):
    return lib().hmco_uniform_x(C.c_uint64(seed), C.c_uint32(window), C.c_uint32(sweep), C.c_uint32(t))


def max_threads():
    return lib().hmco_max_threads()
