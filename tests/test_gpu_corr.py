"""extras.corr: correlations between the per-draw outputs, accumulated on the device (calccorr, src/Hmc.jl:1094-1163).

Checker: numpy.corrcoef of the 5-digit-rounded draws the same call returned -- the matrix upstream's calccorr computes
from the per-draw CSV cells (`cor(Matrix(df))`, :1125; columns mu | sigma | pi | vec(A) | first forecast, :1122).
Tolerance 1e-10 absolute: the device accumulates second moments about the first draw in fp64, numpy centres on the mean."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _round5(a):
    return np.round(a, 5)


def _expected(res, w, K):
    cols = [res["mu"][w, k] for k in range(K)] + [res["sig2"][w, k] for k in range(K)] + [res["pi_end"][w, k] for k in range(K)]
    cols += [res["A"][w].reshape(K * K, -1)[q] for q in range(K * K)]
    cols.append(res["fcast"][w, 0])
    with np.errstate(invalid="ignore", divide="ignore"):       # a constant column: NaN, as Statistics.cor gives
        return np.corrcoef(_round5(np.stack(cols)))


@pytest.mark.parametrize("K,T,nrun,W", [(3, 400, 1500, 5), (2, 300, 700, 3), (4, 250, 600, 2), (8, 600, 300, 2)])
def test_corr_matches_numpy(K, T, nrun, W):
    from hmc_jl_amd import _lib, synth
    Y, Tw, fut = synth.generate_panel(W, T, K)
    res = _lib.estimate_batch_host(Y, Tw, K, 50, nrun, horizons=(12, 3), yreal=np.stack([fut[:, 11], fut[:, 2]], 1), want_corr=True)
    assert (res["status"] == 0).all()
    NC = 3 * K + K * K + 1
    assert res["corr"].shape == (W, NC, NC)
    for w in range(W):
        exp = _expected(res, w, K)
        got = res["corr"][w]
        ok = np.isfinite(exp)
        assert np.array_equal(np.isfinite(got), ok)
        assert np.abs(got[ok] - exp[ok]).max() < 1e-10
        assert np.array_equal(got[ok], got.T[ok])
        assert (np.diag(got)[np.isfinite(np.diag(got))] == 1.0).all()


def test_corr_is_chunk_invariant_and_needs_no_draw_copy(monkeypatch):
    """The same bits whether the run is one launch or many chunks, with or without the draws travelling to the host."""
    from hmc_jl_amd import _lib, synth
    K, W, T, nrun = 3, 4, 300, 2000
    Y, Tw, fut = synth.generate_panel(W, T, K)
    kw = dict(horizons=(12,), yreal=fut[:, 11:12], want_corr=True)
    a = _lib.estimate_batch_host(Y, Tw, K, 20, nrun, **kw)
    monkeypatch.setenv("HMCG_CHUNK_DRAWS", "137")
    b = _lib.estimate_batch_host(Y, Tw, K, 20, nrun, **kw)
    c = _lib.estimate_batch_host(Y, Tw, K, 20, nrun, want_draws=False, **kw)
    monkeypatch.delenv("HMCG_CHUNK_DRAWS")
    monkeypatch.setenv("HMCG_NO_CHUNKS", "1")
    d = _lib.estimate_batch_host(Y, Tw, K, 20, nrun, **kw)
    assert b["launches"] > a["launches"] >= 1 and d["launches"] == 1
    for o in (b, c, d):
        assert np.array_equal(a["corr"], o["corr"], equal_nan=True)
    assert "mu" not in c and np.array_equal(a["summary"], c["summary"])


def test_corr_argument_errors():
    from hmc_jl_amd import _lib, synth
    Y, Tw, fut = synth.generate_panel(2, 200, 3)
    with pytest.raises(_lib.HmcgError):        # no forecast column
        _lib.estimate_batch_host(Y, Tw, 3, 10, 100, horizons=(), want_corr=True)
    with pytest.raises(_lib.HmcgError):        # a partial run
        _lib.estimate_batch_host(Y, Tw, 3, 10, 100, horizons=(12,), yreal=fut[:, 11:12], want_corr=True, sweep_count=50)


def test_corr_skipped_window_is_nan():
    from hmc_jl_amd import _lib, synth
    Y, Tw, fut = synth.generate_panel(3, 200, 3)
    Y[1, 5] = np.nan
    res = _lib.estimate_batch_host(Y, Tw, 3, 10, 200, horizons=(12,), yreal=fut[:, 11:12], want_corr=True)
    assert res["status"][1] != 0 and np.isnan(res["corr"][1]).all()
    assert np.isfinite(res["corr"][0]).all() and np.isfinite(res["corr"][2]).all()


def test_calccorr_device_matrices_equal_the_file_route(inflation, tmp_path):
    """code/run_hmm.jl's windows for three consecutive end dates: the workbook calccorr builds from the per-draw files
    (upstream's route) and the one built from the device's matrices (no file read) agree cell by cell."""
    import datetime as dt
    from hmc_jl_amd import hmc
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    ends = [200, 201, 202]
    res = hmc.estimatewindows(y, dd, ends, horizons=[12], D=3, burnin=300, Nrun=3000, series="official", keep_draws=True, corr=True)
    assert (res.status == 0).all() and res.corr.shape == (3, 19, 19)
    for w in range(3):
        hmc.saveresults(res.samples(w), res.opts[w], str(tmp_path))
    d0, d1 = dd[ends[0] - 1], dd[ends[-1]]           # half-open month range
    pf, dates_f, names_f, mats_f = hmc.calccorr(str(tmp_path), d0.year, d1.year, d0.month, d1.month)
    files = open(pf, "rb").read()
    pd_, dates_d, names_d, mats_d = hmc.calccorr(str(tmp_path / "dev"), d0.year, d1.year, d0.month, d1.month, result=res)
    assert dates_f == dates_d == [str(dd[e - 1]) for e in ends] and names_f == names_d
    for a, b in zip(mats_f, mats_d):
        ok = np.isfinite(a)
        assert np.array_equal(ok, np.isfinite(b)) and np.abs(a[ok] - b[ok]).max() < 1e-10
    assert len(files) > 1000 and len(open(pd_, "rb").read()) > 1000
    # without the draws: same matrices, nothing but summaries and correlations crosses PCIe
    lean = hmc.estimatewindows(y, dd, ends, horizons=[12], D=3, burnin=300, Nrun=3000, series="official", keep_draws=False, corr=True)
    assert np.array_equal(lean.corr, res.corr, equal_nan=True)


def test_corr_device_entry_equals_host_entry():
    """hmcg_estimate_batch_device with extras.corr (device pointers, one launch) against the chunked host entry: the same
    bits -- the moments are accumulated in draw order whatever the chunking.  K = 8 runs the LDS-resident kernel, whose pdf
    scratch shares the device context with the moment tables."""
    from hmc_jl_amd import _lib, synth
    from hmc_jl_amd.device import DevicePanel
    for K, T, nrun in ((3, 500, 800), (8, 700, 200)):
        W = 6
        Y, Tw, fut = synth.generate_panel(W, T, K)
        host = _lib.estimate_batch_host(Y, Tw, K, 20, nrun, (12,), fut[:, 11:12], want_corr=True)
        p = DevicePanel(Y, Tw, K, nrun, (12,), fut[:, 11:12], corr=True)
        p.run(burnin=20)
        p.run(burnin=20, timed=False)          # enqueue-only second run on the same context: ordered behind the first
        p.sync()
        assert np.array_equal(p.corr.cpu().numpy(), host["corr"], equal_nan=True)
        assert np.array_equal(p.mu.cpu().numpy(), host["mu"])
    with pytest.raises(_lib.HmcgError):        # the device entry cannot make up the draw arrays
        DevicePanel(Y, Tw, K, nrun, (12,), fut[:, 11:12], corr=True, keep_draws=False).run(burnin=20)
