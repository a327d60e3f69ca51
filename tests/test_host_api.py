"""Host layer (hmc.py mirror of module Hmc), C-ABI surface and sharding logic -- no GPU compute."""
import ctypes as C
import datetime as dt
import os
import re

import numpy as np
import pytest

import hmc_jl_amd
from hmc_jl_amd import _lib, hmc, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hmcg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hmcg_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = hmc_jl_amd.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hmcg_version() == int(re.search(r"#define HMCG_VERSION (\d+)", hdr).group(1))


def test_no_cpu_fallback_and_struct_validation():
    lib = hmc_jl_amd.load()
    if lib.hmcg_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu tests")
    Y = np.zeros((1, 16)); T = np.array([16], dtype=np.int32)
    with pytest.raises(_lib.HmcgError, match="no HIP device"):
        _lib.estimate_batch_host(Y, T, 3, 1, 1)
    cfg = _lib.make_config(1, 3, 16, 16, 1, 1, (12,))
    cfg.struct_size = 7
    st = np.zeros(1, dtype=np.int32)
    rc = lib.hmcg_estimate_batch(C.byref(cfg), C.c_void_p(Y.ctypes.data), C.c_void_p(T.ctypes.data), None, None, None,
                                 None, None, None, None, C.c_void_p(st.ctypes.data), None, None)
    assert rc == -1 and b"struct_size" in lib.hmcg_last_error()
    for bad in (dict(K=1), dict(K=9), dict(W=0), dict(nrun=-1)):
        kw = dict(W=1, K=3, ldY=16, max_T=16, burnin=1, nrun=1, horizons=(12,))
        kw.update(bad)
        cfg = _lib.make_config(**kw)
        rc = lib.hmcg_estimate_batch(C.byref(cfg), C.c_void_p(Y.ctypes.data), C.c_void_p(T.ctypes.data), None, None,
                                     None, None, None, None, None, C.c_void_p(st.ctypes.data), None, None)
        assert rc == -1, bad


def test_estopt_defaults_and_accessors():
    raw = np.arange(1.0, 201.0)
    dates = [hmc.makedate(120 + i) for i in range(200)]
    o = hmc.estopt(raw, dates)
    assert (o.sampleRange[0], o.sampleRange[-1], o.endIndex, o.D, o.burnin, o.Nrun) == (1, 121, 121, 3, 1000, 1000)
    assert o.signalRange == [] and o.horizons == [12] and o.seed == 1234 and o.series == "offical"
    assert o.noiseSamples == 1 and o.signalburnin == 1000 and o.σsignal == 0.0
    assert o.obsRange == o.sampleRange
    assert hmc.makedate(120) == dt.date(1970, 1, 1) and hmc.makedate(131) == dt.date(1970, 12, 1)
    assert hmc.startdate(o) == dt.date(1970, 1, 1) and hmc.enddate(o) == dates[120] and hmc.enddate(o, 12) == dates[132]
    assert len(hmc.makey(o)) == 121 and hmc.makey(o)[-1] == 121.0
    assert hmc.yobs(o, 133) == 133.0 and hmc.yend(o, 2) == 123.0
    o2 = hmc.estopt(raw, dates, sampleRange=range(1, 51), signalRange=range(49, 51), endIndex=48)
    assert o2.obsRange == list(range(1, 49))
    raw3 = np.arange(1.0, 401.0)
    dates3 = [hmc.makedate(120 + i) for i in range(400)]
    o3 = hmc.estopt(raw3, dates3, sampleRange=range(1, 361), signalRange=range(61, 361), endIndex=60)   # sigLen = 300 > HMCG_MAXTAIL
    with pytest.raises(NotImplementedError):
        hmc.estimatemodel(o3)
    o4 = hmc.estopt(raw, dates, sampleRange=range(1, 51), signalRange=range(40, 45), endIndex=50)     # not a tail of the window
    with pytest.raises(NotImplementedError):
        hmc.estimatemodel(o4)


def test_forecast_host_helper(oracle):
    A = np.array([[0.9, 0.1], [0.3, 0.7]]); mu = np.array([1.0, 5.0]); pe = np.array([0.25, 0.75])
    f, e = hmc.forecast(mu, A, pe, 12, 3.0)
    assert abs(f - oracle.forecast(mu, A, pe, 12)) < 1e-12 and abs(e - (f - 3.0)) < 1e-15


def test_float_text_matches_fixture_style():
    assert hmc._fmt(2.4e-10) == "24e-11"
    assert hmc._fmt(1.6000000000000002e-10) == "16000000000000002e-26"
    assert hmc._fmt(9.847011999991433e-05) == "9847011999991433e-20"
    assert hmc._fmt(0.00010049371999990497) == "0.00010049371999990497"
    assert hmc._fmt(0.0) == "0" and hmc._fmt(1.0) == "1" and hmc._fmt(3.977017290719999) == "3.977017290719999"
    # every numeric cell of a committed fixture survives a parse -> format round trip byte for byte
    path = os.path.join(ROOT, "tests", "golden", "official_filtered_state_probs_summary.csv")
    for line in open(path).read().splitlines()[1:]:
        for cell in line.split(",")[1:]:
            assert hmc._fmt(float(cell)) == cell


def test_summary_and_draw_csv_layout(tmp_path, oracle, inflation, golden_summaries):
    y, dates = inflation
    dd = [dt.date.fromisoformat(d) for d in dates]
    ends = [122, 120, 121]
    opts = [hmc.estopt(y, dd, sampleRange=range(1, e + 1), endIndex=e, burnin=2, Nrun=6) for e in ends]
    rows = [oracle.estimate_window(y[:e], 3, 2, 6, yreal=[y[e + 11]], window_id=i) for i, e in enumerate(ends)]
    summary = np.array([r["summary"] for r in rows])
    paths = hmc.write_summaries(summary, opts, str(tmp_path), legacy_trans_header=True)
    for p, name in zip(paths, hmc.SUMMARY_FILES):
        got = open(p).read().splitlines()
        assert got[0].split(",") == golden_summaries[name][0]               # header byte-identical to the fixture
        assert [g.split(",")[0] for g in got[1:]] == ["1979-12-01", "1980-01-01", "1980-02-01"]   # ascending dates
        assert open(p).read().endswith("\n")
    cur = open(hmc.write_summaries(summary, opts, str(tmp_path / "cur"))[3]).readline().strip().split(",")
    assert cur[1:4] == ["trans_1_1_mean", "trans_2_1_mean", "trans_3_1_mean"]   # current naming, i fastest (src/Hmc.jl:727)
    # per-draw files (saveresults): names, headers, 5-digit rounding, column-major A
    r = rows[1]
    s = hmc.Samples(r["mu"], r["sig2"], r["pi_end"][:, None, :], r["A"], r["fcast"], [hmc.enddate(opts[1])] * 6)
    hmc.saveresults(s, opts[1], str(tmp_path / "draws"))
    names = sorted(os.listdir(tmp_path / "draws"))
    assert names == ["filtered_means_1979-12-01.csv", "filtered_state_probs_1979-12-01.csv",
                     "filtered_trans_probs_1979-12-01.csv", "filtered_variances_1979-12-01.csv",
                     "forecasts_1979-12-01.csv"]
    tp = open(tmp_path / "draws" / "filtered_trans_probs_1979-12-01.csv").read().splitlines()
    assert tp[0] == "date,trans_1_1,trans_2_1,trans_3_1,trans_1_2,trans_2_2,trans_3_2,trans_1_3,trans_2_3,trans_3_3"
    first = [float(v) for v in tp[1].split(",")[1:]]
    np.testing.assert_allclose(first, np.round(r["A"][0].reshape(-1, order="F"), 5), atol=1e-12)
    fc = open(tmp_path / "draws" / "forecasts_1979-12-01.csv").readline().strip()
    assert fc == "date,forecast_12,forecast_error_12"
    assert len(tp) == 7 and all(line.startswith("1979-12-01,") for line in tp[1:])


def test_partition_windows():
    T = np.arange(120, 580)                       # the production expanding windows
    for G in (1, 2, 4, 8):
        parts = shard.partition_windows(T, G)
        allw = sorted(w for p in parts for w in p)
        assert allw == list(range(len(T)))
        loads = [int(T[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= 2 * T.max()
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard.contiguous_blocks(10, 4) == [[0, 1], [2, 3, 4], [5, 6], [7, 8, 9]]
    assert shard.partition_windows([1000] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]


def test_ctypes_structs_match_the_header(tmp_path):
    """The Python mirror of hmcg_config / hmcg_extras / hmcg_timing has the C compiler's sizes."""
    import ctypes as C
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%d %%d", sizeof(hmcg_config), '
                   'sizeof(hmcg_extras), sizeof(hmcg_timing), HMCG_MAXH, HMCG_MAXTAIL);return 0;}\n'
                   % os.path.join(ROOT, "include", "hmcg.h"))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_lib.Config), C.sizeof(_lib.Extras), C.sizeof(_lib.Timing), _lib.HMCG_MAXH, _lib.HMCG_MAXTAIL]


def test_calcdispersion_reproduces_reference_dispersion_text(tmp_path):
    """calcdispersion (src/Hmc.jl:1080-1092) on the head of the reference's committed
    signals_official_noise_0.3_allsignal/forecasts_summary.csv (5 dates x 100 signal ids) must give the first rows of
    its committed forecasts_dispersion.csv character for character (header, CSV.jl float text, n-1 std).
    On the full 45 500-row file 453 of 456 lines are identical and 3 differ in the last digit (summation order)."""
    import shutil
    g = os.path.join(ROOT, "tests", "golden")
    shutil.copy(os.path.join(g, "signals_noise_0.3_allsignal_forecasts_summary_head.csv"), tmp_path / "forecasts_summary.csv")
    out = hmc.calcdispersion(str(tmp_path))
    assert [os.path.basename(p) for p in out] == ["forecasts_dispersion.csv"]
    got = open(out[0]).read().splitlines()
    exp = open(os.path.join(g, "signals_noise_0.3_allsignal_forecasts_dispersion.csv")).read().splitlines()[:6]
    assert got == exp


def test_runaggregate_layouts(tmp_path):
    """runaggregate (src/Hmc.jl:1025-1078) over per-draw files written by basicsave: plain files group by date,
    signal files by (date, signal_1) in the one-argument form and by (date, signalid) in the two-argument form."""
    rng = np.random.default_rng(5)
    plain, sigd = tmp_path / "plain", tmp_path / "sig"
    plain.mkdir(); sigd.mkdir()
    dates = [dt.date(1980, 1, 1), dt.date(1980, 2, 1)]
    data = {}
    for d in dates:
        x = rng.normal(size=(6, 3))
        data[d] = np.rint(x * 1e5) / 1e5
        for name in ("filtered_means", "forecasts"):
            hmc.basicsave(x, [d] * 6, str(plain / ("%s_%s.csv" % (name, d))), ["state_1", "state_2", "state_3"])
        sig = np.repeat(np.array([[1.5, 2.5], [3.25, 4.0]]), 3, axis=0)
        hmc.basicsave(x, [d] * 6, str(sigd / ("filtered_means_%s.csv" % d)), ["state_1", "state_2", "state_3"],
                      signal=sig, signalids=[1, 1, 1, 2, 2, 2])
    out = hmc.runaggregate(str(plain))
    assert sorted(os.path.basename(p) for p in out) == ["filtered_means_summary.csv", "forecasts_summary.csv"]
    lines = open(plain / "filtered_means_summary.csv").read().splitlines()
    assert lines[0] == "date,state_1_mean,state_2_mean,state_3_mean" and len(lines) == 3
    vals = np.array([[float(v) for v in l.split(",")[1:]] for l in lines[1:]])
    assert np.max(np.abs(vals - np.array([data[d].mean(axis=0) for d in dates]))) < 1e-15
    hmc.runaggregate(str(sigd), "filtered_means")
    lines = open(sigd / "filtered_means_summary.csv").read().splitlines()
    assert lines[0] == "date,signalid,state_1_mean,state_2_mean,state_3_mean,signal_1_mean,signal_2_mean" and len(lines) == 5
    assert lines[1].startswith("1980-01-01,1,") and lines[2].startswith("1980-01-01,2,") and lines[2].endswith(",3.25,4")
    hmc.runaggregate(str(sigd))                               # upstream's one-argument form groups by signal_1
    lines = open(sigd / "filtered_means_summary.csv").read().splitlines()
    assert lines[0] == "date,signal_1,signalid_mean,state_1_mean,state_2_mean,state_3_mean,signal_2_mean"
    assert lines[1].startswith("1980-01-01,1.5,1,")
    # and the dispersion of the two-argument summary: mean and n-1 std over the signal ids of a date
    hmc.runaggregate(str(sigd), "filtered_means")
    hmc.calcdispersion(str(sigd))
    lines = open(sigd / "filtered_means_dispersion.csv").read().splitlines()
    assert lines[0].startswith("date,signalid_mean,state_1_mean") and lines[0].endswith("signal_1_std,signal_2_std")
    row = [float(v) for v in lines[1].split(",")[1:]]
    assert row[0] == 1.5 and abs(row[6] - np.std([1.0, 2.0], ddof=1)) < 1e-15


def test_calcdispersion_last_digit_cases(tmp_path):
    """The three dates of the noise-0.3 fixture on which `x ** 0.5` and `sqrt(x)` differ in the last printed digit of a
    standard deviation (1992-10-01, 2009-09-01, 2015-07-01): with the correctly rounded square root calcdispersion
    reproduces the reference's lines character for character -- as it does for all 456 / 456 / 459 lines of the three
    committed dispersion files when run on the reference's full summary files (tools/check_dispersion_full.py)."""
    import shutil
    GOLDEN = os.path.join(ROOT, "tests", "golden")
    shutil.copy(os.path.join(GOLDEN, "signals_noise_0.3_allsignal_forecasts_summary_sqrtcases.csv"), tmp_path / "forecasts_summary.csv")
    hmc.calcdispersion(str(tmp_path))
    got = open(tmp_path / "forecasts_dispersion.csv").read()
    want = open(os.path.join(GOLDEN, "signals_noise_0.3_allsignal_forecasts_dispersion_sqrtcases.csv")).read()
    assert got == want
