"""calccorr (src/Hmc.jl:1094-1163) on the host: the correlation workbook from the per-draw CSV files.  CPU only -- the
files come from the library's native CSV writer (host code), the checker is numpy.corrcoef of the 5-digit-rounded
columns, and the workbook is read back with zipfile + ElementTree (no spreadsheet package in this image)."""
import re
import zipfile
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from hmc_jl_amd import _lib, hmc

NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}


def _read_workbook(path):
    """{sheet name: {(row, col): value}} (0-based)."""
    out = {}
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        names = [s.attrib["name"] for s in wb.find("m:sheets", NS)]
        for i, nm in enumerate(names, 1):
            cells = {}
            for c in ET.fromstring(z.read("xl/worksheets/sheet%d.xml" % i)).iter("{%s}c" % NS["m"]):
                col, row = re.match(r"([A-Z]+)(\d+)", c.attrib["r"]).groups()
                ci = 0
                for ch in col:
                    ci = ci * 26 + ord(ch) - 64
                v = c.find("m:v", NS)
                cells[(int(row) - 1, ci - 1)] = float(v.text) if v is not None else c.find("m:is/m:t", NS).text
            out[nm] = cells
    return names, out


def _fake_draws(rng, K, H, n):
    mu = np.sort(rng.normal(0, 3, (K, n)), axis=0)
    sig2 = rng.gamma(2.0, 1.0, (K, n))
    pi = rng.dirichlet(np.ones(K), n).T
    A = rng.dirichlet(np.ones(K), (K, n))                  # [i][d][j]
    A = np.transpose(A, (2, 0, 1))                         # C-ABI block: [j][i][d]
    fc = rng.normal(2, 1, (2 * H, n))
    return dict(mu=mu[None].copy(), sig2=sig2[None].copy(), pi_end=pi[None].copy(), A=A[None].copy(), fcast=fc[None].copy())


def test_calccorr_from_files_matches_numpy(tmp_path):
    rng = np.random.default_rng(5)
    K, H, n = 3, 2, 400
    dates = ["1990-11-01", "1990-12-01", "1991-01-01"]
    want = {}
    for d in dates:
        res = _fake_draws(rng, K, H, n)
        if d == dates[1]:
            res["pi_end"][0, 2] = 0.25                     # a constant column: NaN row / column, as Statistics.cor gives
        _lib.save_results_csv(str(tmp_path), [d], K, (12, 3), res, n_threads=1)
        cols = np.concatenate([res["mu"][0], res["sig2"][0], res["pi_end"][0], res["A"][0].reshape(K * K, n), res["fcast"][0][:1]])
        with np.errstate(invalid="ignore", divide="ignore"):
            want[d] = np.corrcoef(np.round(cols, 5))
    path, got_dates, names, mats = hmc.calccorr(str(tmp_path), startyear=1990, endyear=1991, startmonth=11, endmonth=2)
    assert got_dates == dates and path.endswith("correlations.xlsx")
    assert names == hmc.corrnames(K, [12, 3])
    for d, m in zip(dates, mats):
        ok = np.isfinite(want[d])
        assert np.array_equal(np.isfinite(m), ok) and np.abs(m[ok] - want[d][ok]).max() < 1e-12
    sheets, wb = _read_workbook(path)
    assert sheets == ["Sheet1", "1990_11", "1990_12", "1991_01"]          # :1145-1147
    NC = len(names)
    first = wb["Sheet1"]
    assert [first[(0, c + 1)] for c in range(NC)] == names and (0, 0) not in first        # sheet["B1", dim=2] (:1155)
    for i, d in enumerate(dates):
        assert first[(i + 1, 0)] == d                                                   # :1158
        for c in range(NC):
            v, w = first[(i + 1, c + 1)], want[d][-1, c]                                # data[i][end, 2:end] (:1159)
            assert (v == "nan" and np.isnan(w)) or abs(v - w) < 1e-12
        sh = wb[d[:4] + "_" + d[5:7]]
        assert sh[(0, 0)] == d and [sh[(0, c + 1)] for c in range(NC)] == names and [sh[(r + 1, 0)] for r in range(NC)] == names
        assert abs(sh[(NC, NC)] - 1.0) < 1e-15 and abs(sh[(1, 2)] - want[d][0, 1]) < 1e-12


def test_calccorr_month_range_is_half_open():
    with pytest.raises(FileNotFoundError):
        hmc.calccorr("/nonexistent-dir-for-test", startyear=2000, endyear=2000, startmonth=1, endmonth=2)
    with pytest.raises(ValueError):        # start == end: the reference's while loop does not run and data[1] throws
        hmc.calccorr("/tmp", startyear=2000, endyear=2000, startmonth=3, endmonth=3)
