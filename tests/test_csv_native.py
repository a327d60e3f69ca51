"""The library's native per-draw CSV writer (hmcg_save_results_csv / hmcg_write_table_csv / hmcg_format_float) -- host
code, no GPU.  SURVEY 8f rows 3-4: byte-faithful CSV text, per-draw dumps at the reference's scale (250 000 rows x 5
files per window, code/run_hmm.jl:103-104; basicsave/saveresults src/Hmc.jl:707-748)."""
import csv
import os
import time

import numpy as np

from hmc_jl_amd import _lib, hmc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_float_text_is_csvjl_style_and_equals_the_python_formatter():
    cases = {0.0: "0", -0.0: "0", 1.0: "1", -3.0: "-3", 5.0: "5", 100000.0: "100000", 0.5: "0.5", 2.5457720896000287: "2.5457720896000287",
             1e-4: "0.0001", 0.00012: "0.00012", 1e-5: "1e-5", 2.4e-10: "24e-11", 4.6210605925778346e-7: "46210605925778346e-23",
             -7.188335999993575e-05: "-7188335999993575e-20", 123456.78901: "123456.78901", 8.52553: "8.52553",
             float("nan"): "NaN", float("inf"): "Inf", float("-inf"): "-Inf"}
    for x, want in cases.items():
        assert _lib.format_float(x) == want, (x, _lib.format_float(x), want)
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.normal(0, 10, 20000), rng.uniform(0, 1, 20000), 10.0 ** rng.uniform(-12, 12, 20000) * rng.choice([-1, 1], 20000),
                         np.rint(rng.normal(0, 3, 20000) * 1e5) / 1e5, rng.integers(-1000, 1000, 5000).astype(float)])
    for x in xs:
        assert _lib.format_float(x) == hmc._fmt(x), (repr(x), _lib.format_float(x), hmc._fmt(x))
        assert float(_lib.format_float(x)) == x                      # round trip


def test_float_text_reproduces_the_committed_fixture_cells():
    """Every numeric cell of the reference's committed summary files re-prints as itself."""
    n = 0
    for name in ("filtered_means", "filtered_variances", "filtered_state_probs", "filtered_trans_probs", "forecasts"):
        for row in list(csv.reader(open(os.path.join(GOLDEN, "official_%s_summary.csv" % name))))[1:]:
            for cell in row[1:]:
                assert _lib.format_float(float(cell)) == cell, (name, row[0], cell)
                n += 1
    assert n > 9000


def fake_result(W, K, H, nd, seed=0):
    rng = np.random.default_rng(seed)
    res = dict(mu=rng.normal(4, 3, (W, K, nd)), sig2=rng.gamma(2.0, 1.0, (W, K, nd)), pi_end=rng.dirichlet(np.ones(K), (W, nd)).transpose(0, 2, 1).copy(),
               A=rng.dirichlet(np.ones(K), (W, K, nd)).transpose(0, 3, 1, 2).copy(), fcast=rng.normal(3, 1, (W, 2 * H, nd)))
    res["pi_end"][0, 0, :5] = [1.0, 0.0, 2.4e-10, 1e-5, 0.99999]      # text edge cases
    return res


def test_native_files_equal_the_python_basicsave_byte_for_byte(tmp_path):
    import datetime as dt
    W, K, H, nd = 3, 3, 2, 400
    res = fake_result(W, K, H, nd)
    dates = [dt.date(1980, 1 + w, 1) for w in range(W)]
    _lib.save_results_csv(str(tmp_path / "native"), dates, K, (6, 12), res, n_threads=2)
    for w in range(W):
        opt = hmc.estopt(np.arange(600.0), [dt.date(1970, 1, 1)] * 600, sampleRange=range(1, 122), endIndex=121, horizons=[6, 12], D=K, Nrun=nd)
        opt.dates[120] = dates[w]
        s = hmc._unpack(dict(res, status=np.zeros(W, dtype=np.int32)), w, nd, K, H, dates[w])
        hmc.saveresults(s, opt, str(tmp_path / "py"), native=False)
        for stem in ("filtered_means", "filtered_variances", "filtered_state_probs", "filtered_trans_probs", "forecasts"):
            a = open(tmp_path / "native" / ("%s_%s.csv" % (stem, dates[w])), "rb").read()
            b = open(tmp_path / "py" / ("%s_%s.csv" % (stem, dates[w])), "rb").read()
            assert a == b, (stem, w)
    hdr = open(tmp_path / "native" / "filtered_trans_probs_1980-01-01.csv").readline().strip()
    assert hdr == "date,trans_1_1,trans_2_1,trans_3_1,trans_1_2,trans_2_2,trans_3_2,trans_1_3,trans_2_3,trans_3_3"
    # signal files: signalid after the date, signal_j columns at the end (src/Hmc.jl:715-717)
    ns, nsave = 4, 2
    sv = np.random.default_rng(1).normal(5, 1, (W, ns, nsave + 1))
    _lib.save_results_csv(str(tmp_path / "sig"), dates, K, (6, 12), res, sigvals=sv, nsave=nsave, n_threads=1)
    lines = open(tmp_path / "sig" / "forecasts_1980-02-01.csv").read().splitlines()
    assert lines[0] == "date,signalid,forecast_6,forecast_error_6,forecast_12,forecast_error_12,signal_1,signal_2"
    assert lines[1].startswith("1980-02-01,1,") and lines[nd // ns + 1].startswith("1980-02-01,2,") and len(lines) == nd + 1
    assert lines[1].endswith("," + hmc._fmt(np.rint(sv[1, 0, 0] * 1e5) / 1e5) + "," + hmc._fmt(np.rint(sv[1, 0, 1] * 1e5) / 1e5))
    # the host API writes through the native writer by default
    s = hmc._unpack(dict(res, status=np.zeros(W, dtype=np.int32)), 0, nd, K, H, dates[0])
    hmc.saveresults(s, opt, str(tmp_path / "api"))
    assert open(tmp_path / "api" / "filtered_means_1980-03-01.csv", "rb").read()[:4] == b"date"


def test_per_draw_dump_at_the_reference_scale(tmp_path):
    """One window at upstream's production size: 250 000 kept draws x 5 files (K = 3, one horizon -> 20 value columns,
    ~37 MB of text).  Timed; the interpreted writer this replaces managed ~0.1 M cells/s."""
    W, K, H, nd = 1, 3, 1, 250000
    res = fake_result(W, K, H, nd, seed=2)
    t0 = time.perf_counter()
    _lib.save_results_csv(str(tmp_path), ["2018-03-01"], K, (12,), res, n_threads=1)
    dt_ = time.perf_counter() - t0
    cells = nd * (3 * K + K * K + 2 * H)
    print("native per-draw dump: %.2f s for %d rows x 5 files, %.1f M cells/s single-threaded" % (dt_, nd, cells / dt_ / 1e6))
    assert dt_ < 20.0
    with open(tmp_path / "forecasts_2018-03-01.csv") as f:
        assert sum(1 for _ in f) == nd + 1
    last = open(tmp_path / "filtered_means_2018-03-01.csv").read().splitlines()[-1].split(",")
    assert [float(v) for v in last[1:]] == list(np.rint(res["mu"][0, :, -1] * 1e5) / 1e5)
