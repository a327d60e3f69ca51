"""Seeded random shapes through the host entry against the oracle: states bit-exact, floats within 1e-9 -- a net under the
per-variant tests for what no hand-picked case hits (window lengths around the chunk and wave boundaries of every kernel,
odd horizon sets, short chains, explicit RNG stream ids, forced flavours, chunked launches).  HMCG_FUZZ_N scales the number of
cases (default 96 base + 32 signal + 24 smoothing; a 12 000 + 4 000 + 3 000 soak of the round-3 final build passed in 76 s)."""
import os

import numpy as np
import pytest

from hmc_jl_amd import synth
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def _case(rng):
    K = int(rng.integers(2, 9))
    W = int(rng.integers(1, 5))
    # lengths: mostly near a boundary of the kernel ladder (256 threads x 1, 2, 4, 8, 16 steps; the LDS kernel beyond)
    top = 2300 if K <= 4 else 1400
    anchors = [2, 3, 64, 65, 256, 257, 512, 513, 1024, 1025, 2048, 2049]
    lens = []
    for _ in range(W):
        if rng.random() < 0.6:
            a = int(rng.choice([x for x in anchors if x <= top]))
            lens.append(int(np.clip(a + rng.integers(-3, 4), 2, top)))
        else:
            lens.append(int(rng.integers(2, top + 1)))
    H = int(rng.integers(0, 3))
    horizons = tuple(int(h) for h in rng.choice(np.arange(1, 41), size=H, replace=False))
    burnin, nrun = int(rng.integers(0, 4)), int(rng.integers(1, 7))
    if K >= 5 and rng.random() < 1.0 / 6.0:
        # one case in six of the LDS-resident kernel takes ONE long window (8 .. 20 steps per thread: the every-eight-steps
        # rescale, the in-place four-row products beyond eight steps, the pdf scratch at every depth), few sweeps -- the
        # oracle's cost stays small
        a = int(rng.choice([2048, 2049, 4096, 4097, 5000]))
        lens = [int(np.clip(a + rng.integers(-3, 4), 1793, 5000)) if rng.random() < 0.7 else int(rng.integers(1793, 5001))]
        burnin, nrun = int(rng.integers(0, 2)), int(rng.integers(1, 4))
    return K, lens, horizons, burnin, nrun


@pytest.mark.parametrize("seed", range(int(os.environ.get("HMCG_FUZZ_N", "96"))))
def test_random_shapes_against_oracle(hmclib, oracle, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    K, lens, horizons, burnin, nrun = _case(rng)
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), K, horizon_pad=41, ragged=lens, window_base=seed * 7)
    yreal = np.stack([fut[:, h - 1] for h in horizons], axis=1) if horizons else None
    wids = None
    if seed % 3 == 0:
        wids = rng.integers(0, 2**31, size=len(lens)).astype(np.int64)
    if seed % 4 == 1 and K <= 4:
        monkeypatch.setenv("HMCG_FLAVOUR", ["p1", "p2", "h"][seed % 3])
    if seed % 5 == 2 or (K >= 5 and max(lens) > 1792):
        monkeypatch.setenv("HMCG_CHUNK_DRAWS", "2")
    g = check_against_oracle(oracle, Y, Tw, K, burnin, nrun, horizons, yreal, window_ids=wids, seed=4321 + seed)
    assert (g["status"] == 0).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("HMCG_FUZZ_N", "96")) // 3))
def test_random_signal_runs_against_oracle(hmclib, oracle, seed):
    """The same for the signal Monte-Carlo path (estimatesignals!, sigLen = 0): random signal tails, save ranges, noise
    levels, kappa and numbers of chained noise samples, K = 2..8 across the steps-per-thread variants of the register-resident
    kernel and, for K >= 5 and one case in seven with longer windows, the SIG form of the LDS-resident one."""
    from test_gpu_parity import check_signals_against_oracle
    rng = np.random.default_rng(5000 + seed)
    K = int(rng.integers(2, 9))                          # K >= 5 (and longer K <= 4 windows): the SIG form of the LDS-resident kernel
    W = int(rng.integers(1, 4))
    top = [1000, 2040, 1000, 700, 700, 600, 600][K - 2] if seed % 7 else [2600, 4500, 2600, 1500, 1500, 1200, 1200][K - 2]
    lens = [int(rng.integers(8, top + 1)) for _ in range(W)]
    Y, Tw, fut = synth.generate_panel(W, max(lens), K, ragged=lens, window_base=seed)
    sig = np.zeros((W, 2), dtype=np.int32)
    save = np.zeros((W, 2), dtype=np.int32)
    for w, T in enumerate(lens):
        slen = int(rng.integers(1, T + 1)) if rng.random() < 0.7 else T          # a tail, or everything a signal
        sig[w] = (T - slen, T)
        ns = int(rng.integers(1, min(slen, 6) + 1))
        save[w] = (T - ns, T)
    ssig = rng.uniform(0.0, 2.0, size=W)
    kappa = float(rng.choice([0.1, 0.3, 0.6, 1.0]))
    check_signals_against_oracle(oracle, Y, Tw, K, int(rng.integers(0, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 4)),
                                 sig, save, kappa, 2.0, 2.0, ssig, fut[:, 11:12])


@pytest.mark.parametrize("seed", range(int(os.environ.get("HMCG_FUZZ_N", "96")) // 4))
def test_random_smoothing_runs_against_oracle(hmclib, oracle, seed):
    """extras.pi_smooth_mean / pi_filter_mean on random shapes, K = 2..8: the SMOOTH variants of the register-resident
    kernel, the smoothing variant of the LDS-resident one beyond their range, against the oracle's literal Pb recursion."""
    from hmc_jl_amd import _lib
    from test_gpu_parity import TOL, close
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(2, 9))
    W = int(rng.integers(1, 4))
    top = 2300 if K <= 4 else 900
    lens = [int(rng.integers(2, top + 1)) for _ in range(W)]
    Y, Tw, fut = synth.generate_panel(W, max(lens), K, ragged=lens, window_base=seed)
    burnin, nrun = int(rng.integers(0, 3)), int(rng.integers(1, 5))
    g = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), fut[:, 11:12], want_state=True, want_smooth=True, want_filter_mean=True)
    for w in range(W):
        T = int(Tw[w])
        o = oracle.estimate_window(Y[w, :T], K, burnin, nrun, (12,), fut[w, 11:12], window_id=w, want_smooth=True)
        assert g["status"][w] == o["status"] == 0
        assert np.array_equal(g["x_final"][w, :T], o["x_final"])
        assert np.max(np.abs(g["pi_smooth_mean"][w, :T] - o["pi_smooth"].mean(axis=0))) < TOL
        assert np.max(np.abs(g["pi_filter_mean"][w, :T].sum(axis=1) - 1)) < 1e-12
        assert close(g["mu"][w].T, o["mu"]) < TOL and close(g["pif_final"][w, :T], o["pif_final"]) < TOL
