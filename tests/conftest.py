import csv
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# arms the library's diagnostic switches (HMCG_FORCE_STREAM, HMCG_VIRTUAL_DEVICES, HMCG_NO_BUCKETS, ...): read once, when the
# library is first used, so it is set before anything loads it; child processes the tests start inherit it
os.environ.setdefault("HMCG_DIAG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def inflation():
    """data/processed/inflation.csv, offic_inf column, float32 -> float64 as the reference's
    .dta stores it (6.18 is really 6.179999828338623)."""
    rows = list(csv.DictReader(open(os.path.join(GOLDEN, "inflation.csv"))))
    y = np.array([np.float32(r["offic_inf"]) for r in rows]).astype(np.float64)
    dates = [r["date"] for r in rows]
    return y, dates


def load_summary(name):
    rows = list(csv.reader(open(os.path.join(GOLDEN, "official_%s_summary.csv" % name))))
    return rows[0], {r[0]: np.array([float(v) for v in r[1:]]) for r in rows[1:]}


@pytest.fixture(scope="session")
def golden_summaries():
    return {n: load_summary(n) for n in ("filtered_means", "filtered_variances", "filtered_state_probs",
                                          "filtered_trans_probs", "forecasts")}


@pytest.fixture(scope="session")
def hmclib():
    """The HIP library through its C ABI; GPU tests fail loudly if it cannot compute."""
    import hmc_jl_amd
    lib = hmc_jl_amd.load()
    assert lib.hmcg_device_count() >= 1, "no GPU: -m gpu tests need an MI355X"
    return hmc_jl_amd
