"""The Hmc-style host API end to end on the GPU: reads like the reference's own runtests.jl."""
import datetime as dt
import os

import numpy as np
import pytest

from hmc_jl_amd import hmc, synth

pytestmark = pytest.mark.gpu


def test_estimatemodel_like_runtests(hmclib, oracle):
    T = 500
    Y99, _ = synth.generate_window(T, 2, seed=126)
    dates = [hmc.makedate(120 + i) for i in range(T)]
    opt = hmc.estopt(Y99, dates, sampleRange=range(1, T - 24 + 1), signalRange=range(2, 2), endIndex=T - 24,
                     horizons=[12], D=2, burnin=3000, Nrun=1000, signalburnin=1, signalNrun=1, series="test")
    samples = hmc.estimatemodel(opt)
    assert samples.μ.shape == (1000, 2) and samples.σ.shape == (1000, 2) and samples.A.shape == (1000, 2, 2)
    assert samples.πb[:, -1, :].shape == (1000, 2) and samples.forecasts.shape == (1000, 2)
    assert samples.obsdates == [dates[T - 25]] * 1000 and samples.status == 0
    assert np.all(np.abs(samples.μ.mean(axis=0) - [-5.0, 4.0]) < 0.3)          # test/runtests.jl:56
    assert np.all(np.abs(samples.σ.mean(axis=0) - [1.0, 0.5]) < 0.5)           # test/runtests.jl:57
    o = oracle.estimate_window(Y99[:T - 24], 2, 3000, 1000, (12,), [Y99[T - 24 + 11]], seed=opt.seed)
    assert np.max(np.abs(samples.μ - o["mu"])) < 1e-9 and np.max(np.abs(samples.A - o["A"])) < 1e-9
    assert np.max(np.abs(samples.forecasts - o["fcast"])) < 1e-9
    f, e = hmc.forecast(samples.μ[5], samples.A[5], samples.πb[5, -1], 12, hmc.yobs(opt, opt.endIndex + 12))
    assert abs(f - samples.forecasts[5, 0]) < 1e-9 and abs(e - samples.forecasts[5, 1]) < 1e-9


def test_run_hmm_style_flow_and_summary_files(hmclib, inflation, tmp_path, golden_summaries):
    """code/run_hmm.jl's flow for a handful of end dates in one batched call, then the five
    `*_summary.csv` files in the committed layout."""
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    ends = [120, 121, 122, 350]
    res = hmc.estimatewindows(y, dd, ends, horizons=[12], D=3, burnin=2000, Nrun=20000, series="official",
                              keep_draws=False)
    assert (res.status == 0).all()
    paths = hmc.write_summaries(res.summary, res.opts, str(tmp_path), legacy_trans_header=True)
    for p, name in zip(paths, hmc.SUMMARY_FILES):
        lines = open(p).read().splitlines()
        hdr, rows = golden_summaries[name]
        assert lines[0].split(",") == hdr
        for line in lines[1:]:
            cells = line.split(",")
            got = np.array([float(v) for v in cells[1:]])
            np.testing.assert_allclose(got, rows[cells[0]], atol=0.12, rtol=0.03)
    one = hmc.estimatewindows(y, dd, [121], horizons=[12], D=3, burnin=5, Nrun=10, keep_draws=True, window_ids=[1])
    s = one.samples(0)
    hmc.saveresults(s, one.opts[0], str(tmp_path / "official"))
    assert len(os.listdir(tmp_path / "official")) == 5


def test_estimatesignals_like_run_hmm_allsignal(hmclib, oracle, inflation, tmp_path):
    """code/run_hmm.jl:158-175: everything a signal, then Hmc.estimatesignals!(opt) and saveresults(hassignals=true)."""
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    e = 121
    opt = hmc.estopt(y, dd, sampleRange=range(1, e + 1), signalRange=range(1, e + 1), signalSave=range(e - 1, e + 1),
                     endIndex=e, horizons=[12], D=3, burnin=200, Nrun=400, signalburnin=20, signalNrun=30,
                     noiseSamples=5, noise=0.3, series="official")
    samples = hmc.estimatesignals(opt)
    assert opt.σsignal > 0                                             # set by the base run (src/Hmc.jl:869-872)
    assert samples.μ.shape == (150, 3) and samples.A.shape == (150, 3, 3) and samples.forecasts.shape == (150, 2)
    assert samples.signalvals.shape == (150, 2) and list(samples.signalids[:31]) == [1] * 30 + [2]
    assert (samples.signalvals[:30] == samples.signalvals[0]).all() and (samples.signalvals[30] != samples.signalvals[0]).any()
    # the same flow on the oracle: base run (kappa = 1, alpha = nu = 1), then the chained noise samples
    base = oracle.estimate_signals(y[:e], 3, 200, 400, 1, sig=(0, e), kappa=1.0, alpha=1.0, nu=1.0, yreal=[y[e + 11]],
                                   window_id=hmc.BASE_RUN_STREAM)       # the base run has its own RNG stream
    ssig = base["sig2"].mean() * 0.3
    assert abs(ssig - opt.σsignal) < 1e-9 * (1 + ssig)
    o = oracle.estimate_signals(y[:e], 3, 20, 30, 5, sig=(0, e), kappa=0.3, alpha=2.0, nu=2.0, sigma_signal=opt.σsignal,
                                save=(e - 2, e), yreal=[y[e + 11]])
    assert np.max(np.abs(samples.μ - o["mu"])) < 1e-9 and np.max(np.abs(samples.forecasts - o["fcast"])) < 1e-9
    assert np.max(np.abs(samples.signalvals[::30] - o["sigvals"])) < 1e-9
    hmc.saveresults(samples, opt, str(tmp_path / "signals_official_noise_0.3_allsignal"), hassignals=True)
    lines = open(tmp_path / "signals_official_noise_0.3_allsignal" / "forecasts_1980-01-01.csv").read().splitlines()
    assert lines[0] == "date,signalid,forecast_12,forecast_error_12,signal_1,signal_2" and len(lines) == 151
    assert lines[1].startswith("1980-01-01,1,") and lines[-1].startswith("1980-01-01,5,")


@pytest.mark.parametrize("signalLen", [1, 12])
def test_estimatesignals_like_run_hmm_future_signals(hmclib, oracle, inflation, signalLen):
    """code/run_hmm.jl:122-156 (the len_1 / len_12 experiments): the window runs signalLen months past the end date
    and those months are the signals.  h = 12 is forecastsignal for signalLen 12 and a 11-step forecast for
    signalLen 1; h = 6 < 12 is unset upstream (NaN here); h = 24 is forecast 24 - signalLen steps ahead."""
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    e = 200
    opt = hmc.estopt(y, dd, sampleRange=range(1, e + signalLen + 1), signalRange=range(e + 1, e + signalLen + 1),
                     signalSave=range(e + 1, e + signalLen + 1), endIndex=e, horizons=[6, 12, 24], D=3, burnin=100, Nrun=200,
                     signalburnin=10, signalNrun=25, noiseSamples=4, noise=1.0, series="official")
    s = hmc.estimatesignals(opt)
    T = e + signalLen
    base = oracle.estimate_signals(y[:T], 3, 100, 200, 1, sig=(e, T), kappa=1.0, alpha=1.0, nu=1.0, horizons=(6, 12, 24),
                                   yreal=[y[e + 5], y[e + 11], y[e + 23]], window_id=hmc.BASE_RUN_STREAM)
    assert abs(base["sig2"].mean() * 1.0 - opt.σsignal) < 1e-9 * (1 + opt.σsignal)
    dev_h = [h - signalLen if h > signalLen else 0 for h in (6, 12, 24)]
    blend = 2 if signalLen == 12 else 0
    o = oracle.estimate_signals(y[:T], 3, 10, 25, 4, sig=(e, T), kappa=1.0, alpha=2.0, nu=2.0, sigma_signal=opt.σsignal,
                                save=(e, T), horizons=dev_h, yreal=[y[e + 5], y[e + 11], y[e + 23]], end_pos=e - 1,
                                blend_mask=blend)
    assert s.forecasts.shape == (100, 6) and s.signalvals.shape == (100, signalLen)
    assert np.max(np.abs(s.μ - o["mu"])) < 1e-9 and np.max(np.abs(s.πb[:, -1, :] - o["pi_end"])) < 1e-9
    if signalLen == 12:
        assert np.isnan(s.forecasts[:, 0:2]).all()                         # h = 6 < sigLen: never assigned upstream
        assert np.max(np.abs(s.forecasts[:, 2:] - o["fcast"][:, 2:])) < 1e-9
    else:
        assert np.max(np.abs(s.forecasts - o["fcast"])) < 1e-9
    assert np.max(np.abs(s.signalvals[::25] - o["sigvals"])) < 1e-9


def test_estimatesignalswindows_equals_single_window_calls(hmclib, inflation):
    """The batched signal entry (many end dates in one GPU call) against one estimatesignals call per date, with the
    single calls' RNG stream id (0) pinned: the same draws, signal values and sigma_signal (set by each base run)."""
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]

    def mk(e):
        return hmc.estopt(y, dd, sampleRange=range(1, e + 2), signalRange=range(e + 1, e + 2), signalSave=range(e + 1, e + 2),
                          endIndex=e, horizons=[1, 12], D=3, burnin=60, Nrun=90, signalburnin=8, signalNrun=20,
                          noiseSamples=3, noise=0.5, series="official")
    ends = [150, 201, 333]
    single = []
    for e in ends:
        o = mk(e)
        single.append((hmc.estimatesignals(o), o.σsignal))
    opts = [mk(e) for e in ends]
    batch = hmc.estimatesignalswindows(opts, window_ids=np.zeros(len(ends)))
    # (a batch is scanned with the steps-per-thread variant of its longest window, so a shorter window's floats
    #  can differ from its single-window run in the last bits: 1e-9 like every other GPU comparison, not bitwise)
    for (s1, ssig), s2, o in zip(single, batch, opts):
        assert abs(o.σsignal - ssig) < 1e-12 * ssig
        for k in ("μ", "σ", "πb", "A", "forecasts", "signalvals"):
            assert np.max(np.abs(getattr(s1, k) - getattr(s2, k))) < 1e-9, k
        assert list(s1.signalids) == list(s2.signalids) and s1.obsdates == s2.obsdates
    # default stream ids: window w runs on stream w -- different draws, same posterior region
    other = hmc.estimatesignalswindows([mk(e) for e in ends])
    assert np.array_equal(other[0].μ, batch[0].μ) and np.max(np.abs(other[1].μ - batch[1].μ)) > 1e-3


@pytest.mark.parametrize("noise", ["0.1", "0.6"])
def test_gpu_signal_path_vs_reference_dispersion_outputs(hmclib, inflation, noise):
    """The GPU chain against the reference's committed allsignal dispersion outputs (see the oracle test of the
    same name for what the fixture holds): 100 noise samples x (2000 + 6000) sweeps."""
    import csv
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    path = os.path.join(os.path.dirname(__file__), "golden", "signals_noise_%s_allsignal_forecasts_dispersion.csv" % noise)
    fx = {k: float(v) for k, v in next(r for r in csv.DictReader(open(path)) if r["date"] == "1980-01-01").items() if k != "date"}
    e, ns, n = 121, 100, 6000
    opt = hmc.estopt(y, dd, sampleRange=range(1, e + 1), signalRange=range(1, e + 1), signalSave=range(e - 1, e + 1),
                     endIndex=e, horizons=[12], D=3, signalburnin=2000, signalNrun=n, noiseSamples=ns, noise=float(noise),
                     σsignal=0.5 * (fx["signal_1_std"] + fx["signal_2_std"]))
    s = hmc.estimatesignals(opt)
    f = s.forecasts[:, 0].reshape(ns, n).mean(axis=1)
    se = np.hypot(f.std(ddof=1) / 10.0, fx["forecast_12_std"] / 10.0)
    assert abs(f.mean() - fx["forecast_12_mean"]) < 4 * se + 0.02, (f.mean(), fx["forecast_12_mean"])
    assert 0.7 < f.std(ddof=1) / fx["forecast_12_std"] < 1.45


def test_smoothed_state_probabilities_host_api(hmclib, oracle, inflation, tmp_path):
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    e = 200
    opt = hmc.estopt(y, dd, sampleRange=range(1, e + 1), endIndex=e, horizons=[12], D=3, burnin=50, Nrun=100)
    s = hmc.estimatemodel(opt, smooth=True)
    assert s.πb_mean.shape == (e, 3) and np.max(np.abs(s.πb_mean.sum(axis=1) - 1)) < 1e-12
    o = oracle.estimate_window(y[:e], 3, 50, 100, (12,), [y[e + 11]], want_smooth=True)
    assert np.max(np.abs(s.πb_mean - o["pi_smooth"].mean(axis=0))) < 1e-9
    hmc.savesmoothresults(s.πb_mean, dd[:e], str(tmp_path))
    lines = open(tmp_path / "smoothed_state_probs.csv").read().splitlines()
    assert lines[0] == "Date,state_1,state_2,state_3" and len(lines) == e + 1 and lines[1].startswith("1970-01-01,")


def test_signal_summaries_from_the_device_equal_the_file_route(hmclib, tmp_path):
    """runaggregate(datadir, var) on a signal run (src/Hmc.jl:1053-1075: group the per-draw rows by date and signalid, mean)
    two ways: (a) as upstream -- saveresults(hassignals=true) writes the per-draw files, runaggregate reads them back;
    (b) estimatesignalswindows(..., summaries=True) takes the per-sample means on the device (extras.sample_summary) and
    write_signal_summaries prints them.  The five files must be identical byte for byte; (b) with keep_draws=False as well,
    where no draw leaves the GPU."""
    raw = np.array([5.0 + np.sin(i / 7.0) * 2 + (i % 13) * 0.3 for i in range(260)])
    dates = [hmc.makedate(120 + i) for i in range(260)]
    opts = [hmc.estopt(raw, dates, sampleRange=range(1, e + 1), signalRange=range(e - 20, e + 1), signalSave=range(e - 1, e + 1),
                       endIndex=e, horizons=[12], D=3, burnin=30, Nrun=60, signalburnin=5, signalNrun=23, noiseSamples=4, noise=0.3,
                       series="t") for e in (150, 201, 177)]
    samples, extra = hmc.estimatesignalswindows(opts, summaries=True)
    a, b = tmp_path / "files", tmp_path / "device"
    for s_, o in zip(samples, opts):
        hmc.saveresults(s_, o, str(a), hassignals=True)
    for var in hmc.SUMMARY_FILES:
        hmc.runaggregate(str(a), var)
    hmc.write_signal_summaries(extra["sample_summary"], extra["signalvals"], opts, str(b))
    for var in hmc.SUMMARY_FILES:
        ta, tb = open(a / (var + "_summary.csv")).read(), open(b / (var + "_summary.csv")).read()
        assert ta == tb, (var, ta[:300], tb[:300])
        assert ta.count("\n") == 1 + 3 * 4 and ta.startswith("date,signalid,")
    opts2 = [hmc.estopt(raw, dates, sampleRange=range(1, e + 1), signalRange=range(e - 20, e + 1), signalSave=range(e - 1, e + 1),
                        endIndex=e, horizons=[12], D=3, burnin=30, Nrun=60, signalburnin=5, signalNrun=23, noiseSamples=4, noise=0.3,
                        series="t") for e in (150, 201, 177)]
    none, extra2 = hmc.estimatesignalswindows(opts2, keep_draws=False, summaries=True)
    assert none is None and np.array_equal(extra2["sample_summary"], extra["sample_summary"])


def test_estimatemodel_returns_the_full_pib_on_request(hmclib, oracle):
    """samples.πb[Nrun, N, D] as upstream's estimatemodel returns it (src/Hmc.jl:864): smooth="draws"."""
    T = 300
    Y, _ = synth.generate_window(T, 3, seed=77)
    dates = [hmc.makedate(120 + i) for i in range(T)]
    opt = hmc.estopt(Y, dates, sampleRange=range(1, T - 12 + 1), endIndex=T - 12, horizons=[12], D=3, burnin=5, Nrun=9, series="t")
    s = hmc.estimatemodel(opt, smooth="draws")
    N = T - 12
    assert s.πb.shape == (9, N, 3) and np.max(np.abs(s.πb.sum(axis=2) - 1)) < 1e-12
    o = oracle.estimate_window(Y[:N], 3, 5, 9, (12,), [Y[N + 11]], seed=opt.seed, window_id=0, want_smooth=True)
    assert np.max(np.abs(s.πb - o["pi_smooth"])) < 1e-9 and np.max(np.abs(s.πb.mean(axis=0) - s.πb_mean)) < 1e-12
    f, e = hmc.forecast(s.μ[0], s.A[0], s.πb[0, -1, :], 12, Y[N + 11])             # :860-862 reads πb[idx, end, :]
    assert abs(f - s.forecasts[0, 0]) < 1e-9 and abs(e - s.forecasts[0, 1]) < 1e-9
