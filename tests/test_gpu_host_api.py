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
