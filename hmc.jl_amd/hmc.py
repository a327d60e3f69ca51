"""Host-side mirror of the reference's `module Hmc` surface for the hot path.

Julia is not available in the build image, so the tested host layer is this Python
module; `julia/Hmc.jl` carries the same surface as a Julia module over the same C
ABI.  Names, argument meaning and result layout follow the reference:

    estopt              src/Hmc.jl:17-73     (fields and keyword defaults)
    makey/startdate/enddate/yobs/yend   src/Hmc.jl:85-107
    makedate            src/Hmc.jl:573-582
    estimatemodel       src/Hmc.jl:850-865   -> (mu, sigma, pib, A, forecasts, obsdates)
    estimatesignals     src/Hmc.jl:868-914   (estimatesignals!; signal ranges ending at endIndex)
    forecast            src/Hmc.jl:658-667
    saveresults/basicsave   src/Hmc.jl:707-748   (five per-window CSVs)
    runaggregate layout src/Hmc.jl:1025-1078 (`*_summary.csv`)
    calccorr            src/Hmc.jl:1094-1163 (correlations.xlsx; matrices from the files or from the device)

New (the reference batches by launching one SLURM task per window,
slurmscripts/base_estimation.sh:5): `estimatewindows`, one call for many windows.

All sampling runs in libhmcgibbs (HIP, gfx950).  No CPU fallback.
"""
import datetime as _dt
import math
import os
import numpy as np

from . import _lib

class Samples:
    """The NamedTuple estimatemodel returns upstream (src/Hmc.jl:864): fields μ, σ, πb, A,
    forecasts, obsdates (ASCII aliases mu, sigma, pib), plus the window's status word."""

    def __init__(self, μ, σ, πb, A, forecasts, obsdates, status=0, signalvals=None, signalids=None):
        self.μ, self.σ, self.πb, self.A = μ, σ, πb, A
        self.forecasts, self.obsdates, self.status = forecasts, obsdates, status
        self.signalvals, self.signalids = signalvals, signalids      # estimatesignals! only (src/Hmc.jl:913)

    mu = property(lambda s: s.μ)
    sigma = property(lambda s: s.σ)
    pib = property(lambda s: s.πb)


def makedate(x):
    """Stata monthly date (months since 1960-01) -> date (src/Hmc.jl:573-582)."""
    y = int(x)
    return _dt.date(y // 12 + 1960, y % 12 + 1, 1)


class estopt:
    """Run options; field names and defaults of the reference struct (src/Hmc.jl:17-60).

    Index ranges are 1-based inclusive `range` objects or sequences, as in Julia
    (sampleRange=range(1, 122) is Julia's 1:121).  As in the reference, windows are
    expected to start at index 1."""

    def __init__(self, rawdata, dates, sampleRange=range(1, 122), signalRange=range(2, 2), signalSave=range(2, 2),
                 endIndex=121, horizons=(12,), D=3, burnin=1000, Nrun=1000, signalburnin=1000, signalNrun=1000,
                 noise=0.0, noiseSamples=1, σsignal=0.0, series="offical", seed=1234):
        self.rawdata = np.asarray(rawdata, dtype=np.float64)
        self.dates = list(dates)
        self.sampleRange = list(sampleRange)
        self.signalRange = list(signalRange)
        self.signalSave = list(signalSave)
        if not set(self.signalRange) <= set(self.sampleRange):
            print("ERROR: signalRange is not a subset of sampleRange")     # @error only logs (src/Hmc.jl:61)
        if not set(self.signalSave) <= set(self.signalRange):
            print("ERROR: signalSave is not a subset of signalRange")      # src/Hmc.jl:62
        self.endIndex = int(endIndex)
        self.horizons = list(horizons)
        self.D = int(D)
        self.burnin = int(burnin)
        self.Nrun = int(Nrun)
        self.signalburnin = int(signalburnin)
        self.signalNrun = int(signalNrun)
        self.noise = float(noise)
        self.noiseSamples = int(noiseSamples)
        self.σsignal = float(σsignal)
        self.series = series
        self.seed = int(seed)
        update_itators(self)


def update_itators(opt):
    """src/Hmc.jl:75-83 (the reference's spelling is kept)."""
    sig = set(opt.signalRange)
    opt.obsRange = [i for i in opt.sampleRange if i not in sig]


def makey(opt):
    return opt.rawdata[np.asarray(opt.sampleRange, dtype=np.int64) - 1]


def enddate(opt, extra=0):
    return opt.dates[opt.endIndex + extra - 1]


def startdate(opt):
    return opt.dates[opt.sampleRange[0] - 1]


def yobs(opt, index):
    return opt.rawdata[index - 1]


def yend(opt, extra=0):
    return opt.rawdata[opt.endIndex + extra - 1]


def forecast(μ, A, πb, horizon, Yreal):
    """(pi' A^h) . mu and its error (src/Hmc.jl:658-667).  Host arithmetic on one draw,
    kept for API parity; per-draw forecasts inside estimatemodel come from the GPU."""
    S1 = np.asarray(πb, dtype=np.float64) @ np.linalg.matrix_power(np.asarray(A, dtype=np.float64), int(horizon))
    f = float(S1 @ np.asarray(μ, dtype=np.float64))
    return f, f - Yreal


def _yreal_row(rawdata, endIndex, horizons):
    out = np.full(len(horizons), np.nan)
    for i, h in enumerate(horizons):
        j = endIndex + h
        if 1 <= j <= len(rawdata):
            out[i] = rawdata[j - 1]
    return out


def _check_live_path(opt):
    if opt.sampleRange[0] != 1 or opt.sampleRange != list(range(1, opt.sampleRange[-1] + 1)):
        raise ValueError("sampleRange must be 1:N (the reference indexes window-relative arrays with absolute "
                         "indices, src/Hmc.jl:254,406 -- only windows starting at 1 are meaningful)")
    sr = opt.signalRange
    if len(sr):
        if sr != list(range(sr[0], sr[-1] + 1)) or sr[-1] != opt.sampleRange[-1]:
            raise NotImplementedError("signalRange must be a contiguous tail of sampleRange")
        if not 0 <= sr[-1] - opt.endIndex <= _lib.HMCG_MAXTAIL:
            raise NotImplementedError("sigLen = last(signalRange) - endIndex (src/Hmc.jl:888) must lie in 0..%d"
                                      % _lib.HMCG_MAXTAIL)


def _sig_ranges(opt):
    """0-based half-open (signal, save) position ranges of the window, or (None, None)."""
    if not len(opt.signalRange):
        return None, None
    sig = (opt.signalRange[0] - 1, opt.signalRange[-1])
    sv = (opt.signalSave[0] - 1, opt.signalSave[-1]) if len(opt.signalSave) else (0, 0)
    return sig, sv


def _check_status(status, what):
    """Per-window status words of a finished call: a window the library skipped is an error here, as it is upstream
    (a NaN observation or an empty window makes the reference throw, src/Hmc.jl:435,464); the numerical flags only warn."""
    import warnings
    status = np.atleast_1d(np.asarray(status))
    bad = np.nonzero(status & _lib.ST_SKIPPED)[0]
    if len(bad):
        names = {_lib.ST_NONFINITE: "non-finite observation", _lib.ST_BAD_T: "window length / end position out of range",
                 _lib.ST_BAD_RANGE: "signal or save range out of range"}
        why = sorted({n for w in bad for b, n in names.items() if status[w] & b})
        raise _lib.HmcgError("%s: window(s) %s were not estimated (%s); their outputs are NaN"
                             % (what, ", ".join(str(int(w)) for w in bad[:8]), "; ".join(why)))
    other = np.nonzero(status)[0]
    if len(other):
        warnings.warn("%s: numerical flags raised in %d window(s) (status bits %s): see HMCG_ST_* in include/hmcg.h"
                      % (what, len(other), sorted({int(s) for s in status[other]})), RuntimeWarning, stacklevel=3)


# RNG stream of estimatesignals!' base run (:869-872): window id with the top bit set, so that the base chain and the
# first noise sample -- same seed, same window, same sweep numbers -- do not replay the same Philox counters
# (upstream's single MersenneTwister stream simply continues from one run into the next)
BASE_RUN_STREAM = 0x80000000


def _unpack(res, w, nrun, K, H, obsdate):
    mu = res["mu"][w].T.copy()                       # (nrun, K)
    sig = res["sig2"][w].T.copy()
    pe = res["pi_end"][w].T.copy()
    A = np.transpose(res["A"][w], (2, 1, 0)).copy()  # [d, i, j]
    fc = res["fcast"][w].T.copy() if H else np.zeros((nrun, 0))
    return Samples(mu, sig, pe[:, None, :], A, fc, [obsdate] * nrun, int(res["status"][w]))


def estimatemodel(opt, device=0, smooth=False, window_id=0):
    """Hmc.estimatemodel(opt) (src/Hmc.jl:850-865) on the GPU.

    smooth="draws" returns the reference's full samples.πb[Nrun, N, D] (every kept draw's smoothed probabilities,
    backwardupdate_P!, :442-457, stored per draw :558) -- 8 N D bytes per draw -- besides πb_mean / πf_mean.

    smooth=True additionally runs the full backward pass (backwardupdate_P!, :442-457) every sweep and
    returns, as `samples.πb_mean` (N, D), the mean over the kept draws of the smoothed probabilities
    `samples.πb[:, t, :]` -- what smoothStates/forecastinsample average upstream (:649-654, :696).

    Returns Samples(μ[Nrun,D], σ[Nrun,D] (variances), πb[Nrun,1,D], A[Nrun,D,D],
    forecasts[Nrun,2H], obsdates).  πb keeps only the window's last time step -- the
    only slice the reference's outputs consume (`samples.πb[:,end,:]`, src/Hmc.jl:744,861)
    -- so `samples.πb[:, -1, :]` reads exactly as upstream.

    If opt carries a signal range the call is the base run of estimatesignals! (:869-872): upstream
    builds HyperParams(Y, D) there, i.e. alpha = nu = 1 and kappa = 1.0 whatever opt.noise says."""
    _check_live_path(opt)
    Y = makey(opt)
    sig, sv = _sig_ranges(opt)
    kw = {}
    if sig is not None:
        kw = dict(sig_range=[sig], save_range=[sv], sigma_signal=[0.0], kappa=1.0, n_samples=1)
    res = _lib.estimate_batch_host(Y[None, :], [len(Y)], opt.D, opt.burnin, opt.Nrun, tuple(opt.horizons),
                                   _yreal_row(opt.rawdata, opt.endIndex, opt.horizons)[None, :], seed=opt.seed,
                                   device=device, want_smooth=bool(smooth), want_filter_mean=bool(smooth), window_ids=[window_id],
                                   want_smooth_draws=(smooth == "draws"), **kw)
    _check_status(res["status"], "estimatemodel")
    s = _unpack(res, 0, opt.Nrun, opt.D, len(opt.horizons), enddate(opt))
    if smooth == "draws":
        s.πb = np.ascontiguousarray(np.transpose(res["pi_smooth_draws"][0, :, :len(Y), :], (2, 1, 0)))     # (Nrun, N, D), as upstream
    if smooth:
        s.πb_mean = res["pi_smooth_mean"][0, :len(Y)]
        s.πf_mean = res["pi_filter_mean"][0, :len(Y)]      # draw-averaged filtered probabilities (sorted labels)
    return s


def savesmoothresults(πb_mean, dates, dir):
    """`smoothed_state_probs.csv` in the layout of savesmoothresults (src/Hmc.jl:750-758): Date, state_1.."""
    os.makedirs(dir, exist_ok=True)
    D = πb_mean.shape[1]
    with open(os.path.join(dir, "smoothed_state_probs.csv"), "w") as f:
        f.write(",".join(["Date"] + ["state_%d" % i for i in range(1, D + 1)]) + "\n")
        for d, row in zip(dates, πb_mean):
            f.write(",".join([str(d)] + [_fmt(v) for v in row]) + "\n")


def estimatesignals(opt, device=0):
    """Hmc.estimatesignals!(opt) (src/Hmc.jl:868-914) on the GPU (the `!`: opt.σsignal is set when it was 0).

    opt.noiseSamples chains of signalburnin + signalNrun sweeps run back to back on
    Yfake = Yreal + N(0,1) * σsignal over opt.signalRange, the chain state carried from one noise sample
    to the next as upstream (:889-895), with HyperParams(opt): alpha = nu = 2, kappa = opt.noise (:148-159).
    Returns Samples with (noiseSamples*signalNrun) draws, sample-major, plus signalvals[Ndraws, len(signalSave)]
    and signalids[Ndraws] (1-based), as the reference's NamedTuple (:913).  The noise comes from this
    library's counter-based RNG (site 5), not Julia's stream.

    Signals past the end date (sigLen = last(signalRange) - endIndex > 0, :888; the len_1 / len_12 experiments of
    code/run_hmm.jl:122-158): πb is the SMOOTHED probability at endIndex (:900); a horizon h > sigLen is forecast
    h - sigLen steps from the window's last step (:906-907), h == sigLen goes through forecastsignal (:908-909)
    and h < sigLen is left unset upstream (uninitialised memory) -- NaN here."""
    _check_live_path(opt)
    if not len(opt.signalRange):
        raise ValueError("estimatesignals needs a signalRange")
    if opt.σsignal == 0:                                   # isapprox(opt.σsignal, 0) (:869)
        base = estimatemodel(opt, device=device, window_id=BASE_RUN_STREAM)
        opt.σsignal = float(np.mean(base.σ)) * opt.noise  # :871
    Y = makey(opt)
    sig, sv = _sig_ranges(opt)
    n, ns = opt.signalNrun, opt.noiseSamples
    sigLen = opt.signalRange[-1] - opt.endIndex                                 # :888
    dev_h = [h - sigLen if h > sigLen else 0 for h in opt.horizons]             # :907
    blend = sum(1 << k for k, h in enumerate(opt.horizons) if h == sigLen and sigLen > 0)
    unset = [k for k, h in enumerate(opt.horizons) if h < sigLen]
    kw = dict(end_pos=[opt.endIndex - 1], blend_mask=blend) if sigLen > 0 else {}
    res = _lib.estimate_batch_host(Y[None, :], [len(Y)], opt.D, opt.signalburnin, n, tuple(dev_h),
                                   _yreal_row(opt.rawdata, opt.endIndex, opt.horizons)[None, :], seed=opt.seed,
                                   device=device, sig_range=[sig], save_range=[sv], sigma_signal=[opt.σsignal],
                                   kappa=opt.noise, n_samples=ns, alpha=2.0, nu=2.0, **kw)
    _check_status(res["status"], "estimatesignals")
    for k in unset:
        res["fcast"][0, 2 * k:2 * k + 2] = np.nan
    s = _unpack(res, 0, ns * n, opt.D, len(opt.horizons), enddate(opt))
    nsave = len(opt.signalSave)
    s.signalvals = np.repeat(res["sigvals"][0][:, :nsave], n, axis=0)           # :904
    s.signalids = np.repeat(np.arange(1, ns + 1), n)                            # :903
    return s


def estimatesignalswindows(opts, device=0, window_ids=None, keep_draws=True, summaries=False):
    """estimatesignals! for many windows in one GPU call (what the reference fans out as one SLURM task per end date,
    slurmscripts/base_estimation.sh:5): opts is a list of estopt with signal ranges that share D, horizons, the sweep
    counts, noiseSamples, noise, seed and sigLen.  Windows whose opt.σsignal is 0 get it from a batched base run first
    (:869-872).  Returns the list of Samples in the order of opts.  RNG stream ids default to the position in `opts`
    (pass window_ids = zeros to reproduce single-window estimatesignals calls draw for draw).
    summaries=True additionally returns, as a second value, a dict with `sample_summary` (W, noiseSamples, 3K+K^2+2H) -- per
    noise sample the mean of its signalNrun rounded draws, taken on the device: the rows runaggregate(datadir, var) makes per
    (date, signalid) (src/Hmc.jl:1053-1075); write_signal_summaries writes them as upstream's files -- and `signalvals`
    (W, noiseSamples, len(signalSave)).  With keep_draws=False no draw leaves the GPU and the first value is None."""
    o0 = opts[0]
    W = len(opts)
    for o in opts:
        _check_live_path(o)
        if not len(o.signalRange):
            raise ValueError("estimatesignalswindows needs a signalRange in every window")
        same = (o.D, tuple(o.horizons), o.burnin, o.Nrun, o.signalburnin, o.signalNrun, o.noiseSamples, o.noise, o.seed,
                o.signalRange[-1] - o.endIndex, len(o.signalSave))
        if same != (o0.D, tuple(o0.horizons), o0.burnin, o0.Nrun, o0.signalburnin, o0.signalNrun, o0.noiseSamples, o0.noise,
                    o0.seed, o0.signalRange[-1] - o0.endIndex, len(o0.signalSave)):
            raise ValueError("windows of one call must share D, horizons, sweep counts, noise settings, seed and sigLen")
    Tw = np.array([len(o.sampleRange) for o in opts], dtype=np.int32)
    ld = int(Tw.max())
    Y = np.zeros((W, ld))
    for w, o in enumerate(opts):
        Y[w, :Tw[w]] = makey(o)
    yreal = np.stack([_yreal_row(o.rawdata, o.endIndex, o.horizons) for o in opts])
    sig = np.array([_sig_ranges(o)[0] for o in opts], dtype=np.int32)
    sv = np.array([_sig_ranges(o)[1] for o in opts], dtype=np.int32)
    wid = np.arange(W, dtype=np.uint32) if window_ids is None else np.asarray(window_ids, dtype=np.uint32)
    need = [w for w, o in enumerate(opts) if o.σsignal == 0]
    if need:                                               # base runs (:869-872): kappa = 1, alpha = nu = 1, no noise
        idx = np.array(need)
        base = _lib.estimate_batch_host(Y[idx], Tw[idx], o0.D, o0.burnin, o0.Nrun, tuple(o0.horizons), yreal[idx], seed=o0.seed,
                                        device=device, want_draws=("sig2",), window_ids=wid[idx] | np.uint32(BASE_RUN_STREAM), sig_range=sig[idx],
                                        save_range=sv[idx], sigma_signal=np.zeros(len(idx)), kappa=1.0, n_samples=1)
        _check_status(base["status"], "estimatesignalswindows (base run)")
        for i, w in enumerate(need):
            opts[w].σsignal = float(np.mean(base["sig2"][i].T.copy())) * opts[w].noise     # :871 (summed as estimatesignals does)
    n, ns = o0.signalNrun, o0.noiseSamples
    sigLen = o0.signalRange[-1] - o0.endIndex
    dev_h = [h - sigLen if h > sigLen else 0 for h in o0.horizons]
    blend = sum(1 << k for k, h in enumerate(o0.horizons) if h == sigLen and sigLen > 0)
    kw = dict(end_pos=[o.endIndex - 1 for o in opts], blend_mask=blend) if sigLen > 0 else {}
    res = _lib.estimate_batch_host(Y, Tw, o0.D, o0.signalburnin, n, tuple(dev_h), yreal, seed=o0.seed, device=device,
                                   window_ids=wid, sig_range=sig, save_range=sv, sigma_signal=[o.σsignal for o in opts],
                                   kappa=o0.noise, n_samples=ns, alpha=2.0, nu=2.0, want_draws=bool(keep_draws),
                                   want_sample_summary=bool(summaries), **kw)
    _check_status(res["status"], "estimatesignalswindows")
    nsave = len(o0.signalSave)
    NP = 3 * o0.D + o0.D * o0.D
    extra = None
    if summaries:
        ss = res["sample_summary"].copy()
        for k, h in enumerate(o0.horizons):
            if h < sigLen:
                ss[:, :, NP + 2 * k:NP + 2 * k + 2] = np.nan
        extra = dict(sample_summary=ss, signalvals=res["sigvals"][:, :, :nsave].copy())
    if not keep_draws:
        return None, extra
    for k, h in enumerate(o0.horizons):
        if h < sigLen:
            res["fcast"][:, 2 * k:2 * k + 2] = np.nan
    out = []
    for w, o in enumerate(opts):
        s = _unpack(res, w, ns * n, o.D, len(o.horizons), enddate(o))
        s.signalvals = np.repeat(res["sigvals"][w][:, :nsave], n, axis=0)
        s.signalids = np.repeat(np.arange(1, ns + 1), n)
        out.append(s)
    return (out, extra) if summaries else out


class BatchResult:
    """Result of estimatewindows: per-window posterior summaries (+ optional draws)."""

    def __init__(self, opts, res, keep_draws):
        self.opts = opts
        self.summary = res["summary"]          # (W, 3K+K^2+2H)
        self.status = res["status"]
        self.kernel_ms = res.get("kernel_ms")
        self.corr = res.get("corr")            # (W, NC, NC) with corr=True: calccorr's matrix per window (corrnames)
        self._res = res if keep_draws else None

    def samples(self, w):
        if self._res is None:
            raise ValueError("draws were not kept (keep_draws=False)")
        o = self.opts[w]
        return _unpack(self._res, w, o.Nrun, o.D, len(o.horizons), enddate(o))


def estimatewindows(rawdata, dates, endIndices, startIndex=1, keep_draws=False, device=0, window_ids=None, corr=False, **kwargs):
    """Batched estimatemodel over many expanding windows (one GPU call).

    Window w uses sampleRange = startIndex:endIndices[w], endIndex = endIndices[w]; the
    remaining keyword arguments are estopt's.  RNG stream ids default to the position
    in `endIndices`; pass window_ids to pin them (sharded runs).  corr=True also accumulates, on the device, the
    correlation matrix between each window's per-draw outputs (calccorr, src/Hmc.jl:1094-1163) -> BatchResult.corr."""
    if startIndex != 1:
        raise ValueError("windows must start at index 1 (see estopt)")
    rawdata = np.asarray(rawdata, dtype=np.float64)
    opts = [estopt(rawdata, dates, sampleRange=range(1, int(e) + 1), endIndex=int(e), **kwargs) for e in endIndices]
    o0 = opts[0]
    for o in opts:
        _check_live_path(o)
        if len(o.signalRange):
            raise NotImplementedError("estimatewindows batches estimatemodel; use estimatesignals per window for signal runs")
    W = len(opts)
    Tw = np.array([len(o.sampleRange) for o in opts], dtype=np.int32)
    ld = int(Tw.max())
    Y = np.zeros((W, ld))
    yreal = np.zeros((W, len(o0.horizons)))
    for w, o in enumerate(opts):
        Y[w, :Tw[w]] = makey(o)
        yreal[w] = _yreal_row(rawdata, o.endIndex, o.horizons)
    res = _lib.estimate_batch_host(Y, Tw, o0.D, o0.burnin, o0.Nrun, tuple(o0.horizons), yreal, seed=o0.seed,
                                   device=device, want_draws=keep_draws, window_ids=window_ids, want_corr=corr)
    return BatchResult(opts, res, keep_draws)


# ----------------------------------------------------------------- CSV output --

def _fmt(x):
    """Float text in the style of the reference's committed CSVs (CSV.jl 0.5.16):
    shortest round-trip digits; integral values without a fraction; |x| < 1e-4 as
    <integer mantissa>e-<n> (e.g. 24e-11)."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    if x == int(x) and abs(x) < 1e15:
        return str(int(x))
    r = repr(x)
    if abs(x) < 1e-4:
        mant, exp = ("%r" % x).split("e") if "e" in r else (None, None)
        if mant is None:                       # repr chose positional notation (1e-4 > |x| >= 1e-5 never does)
            return r
        sign = "-" if mant.startswith("-") else ""
        mant = mant.lstrip("-")
        ip, _, fp = mant.partition(".")
        digits = (ip + fp).lstrip("0") or "0"
        e10 = int(exp) - len(fp)
        return "%s%se%d" % (sign, digits, e10)
    return r


def _header(opt, nfc):
    D = opt.D
    h1 = ["state_%d" % i for i in range(1, D + 1)]
    h2 = ["trans_%d_%d" % (i, j) for j in range(1, D + 1) for i in range(1, D + 1)]   # vec of [i,j] column-major (src/Hmc.jl:727)
    h3 = []
    for h in opt.horizons:
        h3 += ["forecast_%d" % h, "forecast_error_%d" % h]
    return h1, h2, h3[:nfc]


def basicsave(data, dates, fname, dataheader, precision=5, signal=None, signalids=None):
    """src/Hmc.jl:707-722: round to `precision` digits, date first; with signals a `signalid` column follows
    the date and `signal_i` columns (rounded to 5 digits) close the row."""
    data = np.asarray(data, dtype=np.float64)
    sc = 10.0 ** precision
    rounded = np.rint(data * sc) / sc
    header = ["date"] + list(dataheader)
    sig = None
    if signal is not None and np.asarray(signal).shape[1] > 0:
        sig = np.rint(np.asarray(signal, dtype=np.float64) * 1e5) / 1e5
        header += ["signal_%d" % (i + 1) for i in range(sig.shape[1])]
    if signalids is not None and len(signalids) > 0:
        header.insert(1, "signalid")
    with open(fname, "w") as f:
        f.write(",".join(header) + "\n")
        for i, (d, row) in enumerate(zip(dates, rounded)):
            cells = [str(d)]
            if sig is not None:                  # the reference only emits signalids together with signals (:715-717)
                cells.append(str(int(signalids[i])))
            cells += [_fmt(v) for v in row]
            if sig is not None:
                cells += [_fmt(v) for v in sig[i]]
            f.write(",".join(cells) + "\n")


def saveresults(samples, opt, dir, hassignals=False, native=True):
    """Five per-window CSVs with the reference's names and columns (src/Hmc.jl:724-748).
    Without signals the reference ignores `dir` and writes to data/output/<series>/ -- which is what its
    only caller passes (code/run_hmm.jl:116,120); with signals it writes under `dir` (:735-739).
    native=True (default) writes through the library's C writer (hmcg_save_results_csv: same bytes, ~100x the speed of
    the interpreted loop); native=False keeps the Python basicsave (used as the cross-check in the tests)."""
    os.makedirs(dir, exist_ok=True)
    h1, h2, h3 = _header(opt, samples.forecasts.shape[1])
    ed = enddate(opt)
    n = samples.μ.shape[0]
    if native:
        K = samples.μ.shape[1]
        H = samples.forecasts.shape[1] // 2
        res = dict(mu=samples.μ.T[None], sig2=samples.σ.T[None], pi_end=samples.πb[:, -1, :].T[None],
                   A=np.transpose(samples.A, (2, 1, 0))[None], fcast=samples.forecasts.T[None] if H else None)
        sv = None
        nsave = 0
        if hassignals and samples.signalvals is not None:
            ids = np.asarray(samples.signalids)
            ns = int(ids.max()) if len(ids) else 1
            per = n // ns
            sv = np.ascontiguousarray(np.asarray(samples.signalvals)[::per][None])      # (1, n_samples, nsave): one row per noise sample
            nsave = sv.shape[2]
        _lib.save_results_csv(dir, [ed], K, tuple(opt.horizons[:H]), res, sigvals=sv, nsave=nsave, n_threads=1)
        return
    kw = dict(signal=samples.signalvals, signalids=samples.signalids) if hassignals else {}
    basicsave(samples.μ, samples.obsdates, os.path.join(dir, "filtered_means_%s.csv" % ed), h1, **kw)
    basicsave(samples.σ, samples.obsdates, os.path.join(dir, "filtered_variances_%s.csv" % ed), h1, **kw)
    basicsave(samples.πb[:, -1, :], samples.obsdates, os.path.join(dir, "filtered_state_probs_%s.csv" % ed), h1, **kw)
    basicsave(samples.A.reshape(n, -1, order="F"), samples.obsdates,
              os.path.join(dir, "filtered_trans_probs_%s.csv" % ed), h2, **kw)
    basicsave(samples.forecasts, samples.obsdates, os.path.join(dir, "forecasts_%s.csv" % ed), h3, **kw)


SUMMARY_FILES = ("filtered_means", "filtered_variances", "filtered_state_probs", "filtered_trans_probs", "forecasts")


def write_summaries(summary, opts, dir, legacy_trans_header=False):
    """`*_summary.csv` files in runaggregate's layout (src/Hmc.jl:1025-1078): header
    `date,<col>_mean,...`, one row per end date in ascending date order.  `summary` is the
    (W, 3K+K^2+2H) block of on-device means of the 5-digit-rounded draws.
    legacy_trans_header reproduces the committed fixtures' trans_<j>_<i> naming
    (code/deprecated/Hmc.jl_08072019bak:711)."""
    os.makedirs(dir, exist_ok=True)
    o0 = opts[0]
    K, H = o0.D, len(o0.horizons)
    h1, h2, h3 = _header(o0, 2 * H)
    if legacy_trans_header:
        h2 = ["trans_%d_%d" % (j, i) for j in range(1, K + 1) for i in range(1, K + 1)]
    cols = [(0, K, h1), (K, 2 * K, h1), (2 * K, 3 * K, h1), (3 * K, 3 * K + K * K, h2),
            (3 * K + K * K, 3 * K + K * K + 2 * H, h3)]
    order = sorted(range(len(opts)), key=lambda w: enddate(opts[w]))
    paths = []
    for name, (a, b, hdr) in zip(SUMMARY_FILES, cols):
        p = os.path.join(dir, name + "_summary.csv")
        with open(p, "w") as f:
            f.write(",".join(["date"] + [h + "_mean" for h in hdr]) + "\n")
            for w in order:
                f.write(",".join([str(enddate(opts[w]))] + [_fmt(v) for v in summary[w, a:b]]) + "\n")
        paths.append(p)
    return paths


def write_signal_summaries(sample_summary, signalvals, opts, dir, legacy_trans_header=False):
    """The five `<var>_summary.csv` files runaggregate(datadir, var) (src/Hmc.jl:1053-1075) writes for a SIGNAL run -- header
    `date,signalid,<col>_mean...,signal_1_mean...`, one row per (date, noise sample), dates in ascending order as the per-draw
    files sort -- straight from the device's per-sample means (estimatesignalswindows(..., summaries=True)): the
    noiseSamples x signalNrun x 5 files of per-draw text per date are not needed.  The signal columns are constant within a
    sample; their "mean" is still taken as the file route takes it (the rounded value summed signalNrun times, divided), so
    that both routes print the same text (upstream's own fixture shows the effect: signal_1_mean = 12.408199999956814)."""
    os.makedirs(dir, exist_ok=True)
    o0 = opts[0]
    K, H = o0.D, len(o0.horizons)
    h1, h2, h3 = _header(o0, 2 * H)
    if legacy_trans_header:
        h2 = ["trans_%d_%d" % (j, i) for j in range(1, K + 1) for i in range(1, K + 1)]
    cols = [(0, K, h1), (K, 2 * K, h1), (2 * K, 3 * K, h1), (3 * K, 3 * K + K * K, h2),
            (3 * K + K * K, 3 * K + K * K + 2 * H, h3)]
    ns, nsave = sample_summary.shape[1], signalvals.shape[2]
    n = o0.signalNrun

    def const_mean(r):                                    # _seq_mean([r] * n) without the list
        if n <= 4096:
            acc = 0.0
            for _ in range(n):
                acc += r
            return acc / n
        return float(np.cumsum(np.full(n, r))[-1]) / n    # (cumsum adds in sequence)
    order = sorted(range(len(opts)), key=lambda w: enddate(opts[w]))
    paths = []
    for name, (a, b, hdr) in zip(SUMMARY_FILES, cols):
        p = os.path.join(dir, name + "_summary.csv")
        with open(p, "w") as f:
            f.write(",".join(["date", "signalid"] + [h + "_mean" for h in hdr] + ["signal_%d_mean" % (i + 1) for i in range(nsave)]) + "\n")
            for w in order:
                for smp in range(ns):
                    f.write(",".join([str(enddate(opts[w])), str(smp + 1)] + [_fmt(v) for v in sample_summary[w, smp, a:b]] +
                                     [_fmt(const_mean(float(np.rint(v * 1e5) / 1e5))) for v in signalvals[w, smp]]) + "\n")
        paths.append(p)
    return paths


# ------------------------------------------- aggregation over per-draw CSVs (host) --
# runaggregate / calcdispersion (src/Hmc.jl:1025-1092) work on the CSV files saveresults wrote, exactly as
# upstream: they are file-in / file-out and never touch the GPU.  For the estimatemodel path the on-device
# `summary` block (write_summaries above) gives the same numbers without the per-draw files.

def _read_csv(path):
    import csv
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    return rows[0], rows[1:]


def _seq_mean(vals):
    s = 0.0
    for v in vals:
        s += v
    return s / len(vals) if vals else float("nan")


def _seq_std(vals):
    """Statistics.std: corrected (n-1) two-pass sample standard deviation."""
    n = len(vals)
    if n < 2:
        return float("nan")
    m = _seq_mean(vals)
    s = 0.0
    for v in vals:
        s += (v - m) * (v - m)
    return math.sqrt(s / (n - 1))          # sqrt, not ** 0.5: pow() is not correctly rounded (3 of 456 lines differed in the last digit)


def _aggregate(header, rows, groups, funcs):
    """DataFrames.aggregate(df, groups, funcs): one row per group in order of first appearance; columns = groups,
    then for each function all non-group columns named <col>_<fname>."""
    gi = [header.index(g) for g in groups]
    vi = [i for i in range(len(header)) if i not in gi]
    order, buckets = [], {}
    for r in rows:
        key = tuple(r[i] for i in gi)
        if key not in buckets:
            buckets[key] = []
            order.append(key)
        buckets[key].append([float(r[i]) for i in vi])
    out_header = list(groups) + ["%s_%s" % (header[i], name) for name, _ in funcs for i in vi]
    out = []
    for key in order:
        cols = list(zip(*buckets[key]))
        out.append(list(key) + [_fmt(fn(list(c))) for _, fn in funcs for c in cols])
    return out_header, out


def _write_csv(path, header, rows):
    with open(path, "w") as f:
        f.write(",".join(header) + "\n")
        for r in rows:
            f.write(",".join(r) + "\n")


def runaggregate(datadir, var=None):
    """runaggregate(datadir) / runaggregate(datadir, var) (src/Hmc.jl:1025-1078): `<var>_summary.csv` = the mean of
    every column of each per-draw file `<var>_<date>.csv`, one row per date -- or per (date, signal) when the files
    carry signal columns.  As upstream, the one-argument form groups signal files by :signal_1 and the two-argument
    form by :signalid (the committed `forecasts_summary.csv` fixtures come from the latter)."""
    import glob
    if not os.path.isdir(datadir):
        raise ValueError("%s is not a valid directory" % datadir)
    first = sorted(glob.glob(os.path.join(datadir, "filtered_means*")))
    first = [f for f in first if "summary" not in f and "dispersion" not in f]
    hassignal = bool(first) and any("signal" in h for h in _read_csv(first[0])[0])
    groups = ["date"] if not hassignal else (["date", "signal_1"] if var is None else ["date", "signalid"])
    written = []
    for v in (SUMMARY_FILES_UPSTREAM_ORDER if var is None else (var,)):
        files = sorted(f for f in glob.glob(os.path.join(datadir, v + "*")) if "summary" not in f and "dispersion" not in f)
        if not files:
            continue
        out_header, out_rows = None, []
        for f in files:
            header, rows = _read_csv(f)
            h, r = _aggregate(header, rows, groups, [("mean", _seq_mean)])
            out_header = out_header or h
            out_rows += r
        path = os.path.join(datadir, v + "_summary.csv")
        _write_csv(path, out_header, out_rows)
        written.append(path)
    return written


SUMMARY_FILES_UPSTREAM_ORDER = ("filtered_means", "filtered_state_probs", "filtered_variances", "filtered_trans_probs", "forecasts")


def calcdispersion(datadir):
    """calcdispersion(datadir) (src/Hmc.jl:1080-1092): for each `<var>_summary.csv`, strip `_mean` from the column
    names and write mean and (n-1) standard deviation of every column per date to `<var>_dispersion.csv`
    (columns: date, all <col>_mean, then all <col>_std -- the signal id included, as upstream)."""
    if not os.path.isdir(datadir):
        raise ValueError("%s is not a valid directory" % datadir)
    written = []
    for v in SUMMARY_FILES_UPSTREAM_ORDER:
        src = os.path.join(datadir, v + "_summary.csv")
        if not os.path.exists(src):
            continue
        header, rows = _read_csv(src)
        header = [h.replace("_mean", "") for h in header]
        h, r = _aggregate(header, rows, ["date"], [("mean", _seq_mean), ("std", _seq_std)])
        path = os.path.join(datadir, v + "_dispersion.csv")
        _write_csv(path, h, r)
        written.append(path)
    return written


# ------------------------------------------------------------ correlation workbook --
# calccorr (src/Hmc.jl:1094-1163): per end date, the correlation matrix between the per-draw columns
# mu_1..K | sigma_1..K | pi_1..K | trans_* (file order) | forecast_<first horizon>, one workbook sheet per date plus, on
# the first sheet, the forecast's row of every date.  The matrices come either from the per-draw CSV files (as upstream:
# file in, file out, no GPU) or straight from the device (estimatewindows(..., corr=True): extras.corr, no per-draw file).

def corrnames(K, horizons):
    """Labels of a correlation matrix: what calccorr derives from the CSV headers (:1104,1109,1114,1122)."""
    return (["μ%d" % i for i in range(1, K + 1)] + ["σ%d" % i for i in range(1, K + 1)] + ["π%d" % i for i in range(1, K + 1)]
            + ["trans_%d_%d" % (i, j) for j in range(1, K + 1) for i in range(1, K + 1)] + ["forecast_%d" % horizons[0]])


def _xlsx_col(c):
    s = ""
    c += 1
    while c:
        c, r = divmod(c - 1, 26)
        s = chr(65 + r) + s
    return s


def _xlsx_sheet(cells):
    """cells: {(row, col): value} (0-based) -> worksheet XML; numbers as numeric cells, NaN / text as inline strings."""
    from xml.sax.saxutils import escape
    rows = {}
    for (r, c), v in cells.items():
        rows.setdefault(r, []).append((c, v))
    out = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
           '<worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"><sheetData>']
    for r in sorted(rows):
        out.append('<row r="%d">' % (r + 1))
        for c, v in sorted(rows[r]):
            ref = "%s%d" % (_xlsx_col(c), r + 1)
            if isinstance(v, (int, float, np.floating)) and np.isfinite(v):
                out.append('<c r="%s"><v>%s</v></c>' % (ref, repr(float(v))))
            else:
                out.append('<c r="%s" t="inlineStr"><is><t>%s</t></is></c>' % (ref, escape(str(v))))
        out.append("</row>")
    out.append("</sheetData></worksheet>")
    return "".join(out)


def write_corr_workbook(path, dates, names, mats):
    """correlations.xlsx as calccorr lays it out (:1140-1161): sheet 1 ("Sheet1") = header row of the labels from B1 and
    one row per date (A: the date, B..: the LAST row of its matrix -- the forecast's correlations); then one sheet
    "yyyy_mm" per date holding the labelled matrix at A1 (corner cell: the date).  mats[i]: (NC, NC) array of dates[i]."""
    import zipfile
    names = list(names)
    sheets = [("Sheet1", {})]
    first = sheets[0][1]
    for c, nm in enumerate(names):
        first[(0, c + 1)] = nm
    for i, (d, m) in enumerate(zip(dates, mats)):
        d = str(d)
        m = np.asarray(m)
        first[(i + 1, 0)] = d
        for c in range(len(names)):
            first[(i + 1, c + 1)] = m[-1, c]
        cells = {(0, 0): d}
        for c, nm in enumerate(names):
            cells[(0, c + 1)] = nm
            cells[(c + 1, 0)] = nm
        for r in range(len(names)):
            for c in range(len(names)):
                cells[(r + 1, c + 1)] = m[r, c]
        sheets.append((d[0:4] + "_" + d[5:7], cells))
    ct = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
          '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
          '<Default Extension="xml" ContentType="application/xml"/>'
          '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>']
    wb = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n<workbook xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" '
          'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets>']
    rel = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">']
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        for i, (nm, cells) in enumerate(sheets, 1):
            ct.append('<Override PartName="/xl/worksheets/sheet%d.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>' % i)
            wb.append('<sheet name="%s" sheetId="%d" r:id="rId%d"/>' % (nm, i, i))
            rel.append('<Relationship Id="rId%d" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet%d.xml"/>' % (i, i))
            z.writestr("xl/worksheets/sheet%d.xml" % i, _xlsx_sheet(cells))
        z.writestr("[Content_Types].xml", "".join(ct) + "</Types>")
        z.writestr("_rels/.rels", '<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>')
        z.writestr("xl/workbook.xml", "".join(wb) + "</sheets></workbook>")
        z.writestr("xl/_rels/workbook.xml.rels", "".join(rel) + "</Relationships>")
    return path


def _corr_from_files(datadir, date):
    """One date's labelled matrix from its five per-draw files, as calccorr reads them (:1100-1125)."""
    cols, names = [], []
    for stem, sym in (("filtered_means_", "μ"), ("filtered_variances_", "σ"), ("filtered_state_probs_", "π"),
                      ("filtered_trans_probs_", None), ("forecasts_", None)):
        header, rows = _read_csv(os.path.join(datadir, stem + date + ".csv"))
        data = np.array([[float(x) for x in r[1:]] for r in rows]).T
        hn = header[1:]
        if stem == "forecasts_":
            data, hn = data[:1], hn[:1]               # df4[!, [2]]: the first forecast column only (:1122)
        cols.append(data)
        names += [h.replace("state_", sym) if sym else h for h in hn]
    with np.errstate(invalid="ignore", divide="ignore"):
        return names, np.corrcoef(np.concatenate(cols))


def calccorr(datadir, startyear=1980, endyear=2018, startmonth=1, endmonth=2, result=None):
    """calccorr(datadir; startyear, endyear, startmonth, endmonth) (src/Hmc.jl:1094-1163): `correlations.xlsx` under
    datadir for the months from (startyear, startmonth) up to but excluding (endyear, endmonth).

    result=None: as upstream, every month's five per-draw files `filtered_*_<yyyy-mm>-01.csv` / `forecasts_<...>.csv` are
    read back and correlated on the host.  result = a BatchResult of estimatewindows(..., corr=True): the matrices were
    accumulated on the device while the draws were in HBM -- no per-draw file is read (or needs to exist); every month
    of the range must be one of the result's end dates.  Returns (path, dates, names, matrices)."""
    dates = []
    year, month = int(startyear), int(startmonth)
    while not (year == endyear and month == endmonth):
        dates.append("%04d-%02d-01" % (year, month))
        month += 1
        if month > 12:
            month, year = 1, year + 1
        if year > endyear + 1:
            raise ValueError("the month range never reaches (endyear, endmonth)")
    if not dates:
        raise ValueError("empty month range (upstream fails on data[1] here, src/Hmc.jl:1155)")
    mats, names = [], None
    if result is None:
        for d in dates:
            names, m = _corr_from_files(datadir, d)
            mats.append(m)
    else:
        if getattr(result, "corr", None) is None:
            raise ValueError("result carries no correlation matrices (estimatewindows(..., corr=True))")
        by_date = {str(enddate(o)): w for w, o in enumerate(result.opts)}
        o0 = result.opts[0]
        names = corrnames(o0.D, o0.horizons)
        for d in dates:
            if d not in by_date:
                raise ValueError("no window ends on %s" % d)
            mats.append(result.corr[by_date[d]])
    os.makedirs(datadir, exist_ok=True)
    path = write_corr_workbook(os.path.join(datadir, "correlations.xlsx"), dates, names, mats)
    return path, dates, names, mats
