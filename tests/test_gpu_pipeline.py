"""The host entry's chunk pipeline, the multi-device entry and the configs[2] workload.

The host entries cut a long chain into chunks and stream each chunk's per-draw outputs to the caller while the next
chunk samples; the chain state crosses chunk boundaries through the checkpoint block.  Everything here is an
equality test: a chunked run, a differently chunked run, a single launch and the device-resident entry must agree
bit for bit."""
import numpy as np
import pytest

from hmc_jl_amd import _lib, shard, synth

pytestmark = pytest.mark.gpu
KEYS = ("mu", "sig2", "A", "pi_end", "fcast", "summary", "status", "x_final", "pif_final")


def same(a, b, keys=KEYS):
    for k in keys:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("K,T", [(3, 1000), (2, 300), (4, 700), (8, 600)])
def test_chunked_host_entry_equals_single_launch(hmclib, monkeypatch, K, T):
    Y, Tw, fut = synth.generate_panel(5, T, K, ragged=[T, T - 1, T // 2, 17, T - 100])
    args = (Y, Tw, K, 7, 130, (1, 12), fut[:, [0, 11]])
    monkeypatch.setenv("HMCG_NO_CHUNKS", "1")
    one = _lib.estimate_batch_host(*args, want_state=True)
    assert one["launches"] == 1
    monkeypatch.delenv("HMCG_NO_CHUNKS")
    auto = _lib.estimate_batch_host(*args, want_state=True)
    assert auto["launches"] >= 3
    same(one, auto)
    for cap in ("1", "7", "64"):                        # more chunks than ring slots, odd sizes
        monkeypatch.setenv("HMCG_CHUNK_DRAWS", cap)
        r = _lib.estimate_batch_host(*args, want_state=True)
        assert r["launches"] >= 130 // int(cap) // 2
        same(one, r)


def test_chunked_signal_and_smooth_paths(hmclib, monkeypatch):
    """Chunk boundaries inside a noise sample (the noisy observations are regenerated from the counter-based RNG) and
    the running sums of the smoothed / filtered means carried across chunks."""
    K, T = 3, 500
    Y, Tw, fut = synth.generate_panel(3, T, K)
    sig = np.stack([Tw - 30, Tw], axis=1).astype(np.int32)
    kw = dict(sig_range=sig, save_range=sig, sigma_signal=np.array([0.4, 0.9, 0.1]), kappa=0.6, n_samples=4, alpha=2.0, nu=2.0,
              want_sample_summary=True)
    monkeypatch.setenv("HMCG_NO_CHUNKS", "1")
    a = _lib.estimate_batch_host(Y, Tw, K, 5, 21, (12,), fut[:, 11:12], want_state=True, **kw)
    b = _lib.estimate_batch_host(Y, Tw, K, 5, 21, (12,), fut[:, 11:12], want_state=True, want_smooth=True, want_filter_mean=True)
    monkeypatch.delenv("HMCG_NO_CHUNKS")
    for cap in (None, "5", "13"):
        if cap:
            monkeypatch.setenv("HMCG_CHUNK_DRAWS", cap)
        a2 = _lib.estimate_batch_host(Y, Tw, K, 5, 21, (12,), fut[:, 11:12], want_state=True, **kw)
        b2 = _lib.estimate_batch_host(Y, Tw, K, 5, 21, (12,), fut[:, 11:12], want_state=True, want_smooth=True, want_filter_mean=True)
        assert cap is None or (a2["launches"] > 1 and b2["launches"] > 1)
        same(a, a2, KEYS + ("sigvals", "sample_summary"))
        same(b, b2, KEYS + ("pi_smooth_mean", "pi_filter_mean"))


def test_host_entry_equals_device_entry(hmclib):
    from hmc_jl_amd.device import DevicePanel
    Y, Tw, fut = synth.generate_panel(32, 800, 3)
    host = _lib.estimate_batch_host(Y, Tw, 3, 10, 200, (12,), fut[:, 11:12])
    p = DevicePanel(Y, Tw, 3, 200, (12,), fut[:, 11:12])
    p.run(burnin=10)
    for k in ("mu", "sig2", "A", "pi_end", "fcast", "summary"):
        assert np.array_equal(getattr(p, k).cpu().numpy(), host[k]), k


def test_multi_device_entry_with_one_device_equals_plain_entry(hmclib):
    """hmcg_estimate_batch_multi, n_devices = 1: the LPT partition reorders the windows inside the call (longest
    first), global ids keep every row what it is in the plain call; subsets of outputs may be NULL."""
    lens = [700, 50, 400, 699, 3, 256, 257, 64]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    ids = np.array([5, 900, 17, 3, 2 ** 31 + 5, 8, 1, 0], dtype=np.uint32)
    a = _lib.estimate_batch_host(Y, Tw, 3, 5, 40, (12,), fut[:, 11:12], want_state=True, window_ids=ids)
    b = _lib.estimate_batch_host(Y, Tw, 3, 5, 40, (12,), fut[:, 11:12], want_state=True, window_ids=ids, devices=[0])
    same(a, b)
    assert b["per_device"][0]["windows"] == len(lens) and b["per_device"][0]["device"] == 0
    c = _lib.estimate_batch_host(Y, Tw, 3, 5, 40, (12,), fut[:, 11:12], want_draws=("A",), devices=[0], window_ids=ids)
    assert np.array_equal(c["A"], a["A"]) and np.array_equal(c["summary"], a["summary"])
    with pytest.raises(_lib.HmcgError, match="listed twice"):
        _lib.estimate_batch_host(Y, Tw, 3, 1, 2, (12,), fut[:, 11:12], devices=[0, 0])
    with pytest.raises(_lib.HmcgError, match="out of range"):
        _lib.estimate_batch_host(Y, Tw, 3, 1, 2, (12,), fut[:, 11:12], devices=[0, 63])


def test_cfg3_workload_2048_windows_in_8_shards(hmclib, oracle):
    """BASELINE configs[2]: 3-state, T=1000, 2048 windows sharded over 8 GPUs with a final gather.  On the one-GPU box
    the eight shards of `partition_windows` run one after the other, each as its own call with global window ids, and
    are reassembled through `shard.gather_blocks` (world = 1 path): the result must equal the unsharded call bit for
    bit.  Size-independent properties on all 2048 windows; oracle parity on a few rows."""
    import torch
    W, T, K, burnin, nrun = 2048, 1000, 3, 2, 10
    Y, Tw, fut = synth.generate_panel(W, T, K)
    yreal = fut[:, 11:12]
    full = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, (12,), yreal, want_state=True)
    assert (full["status"] == 0).all() and full["helper_waves"] == 0          # more windows than CUs: capped plain kernels
    parts = shard.partition_windows(Tw, 8)
    assert sorted(sum(parts, [])) == list(range(W)) and {len(p) for p in parts} == {256}
    NS = full["summary"].shape[1]
    gathered = torch.zeros((W, NS + K * nrun), dtype=torch.float64)
    for ids in parts:
        ids = np.array(ids)
        r = _lib.estimate_batch_host(Y[ids], Tw[ids], K, burnin, nrun, (12,), yreal[ids], window_ids=ids)
        assert r["helper_waves"] == 4                                         # a 256-window shard gets a CU per window
        block = torch.from_numpy(np.concatenate([r["summary"], r["mu"].reshape(len(ids), -1)], axis=1))
        gathered += shard.gather_blocks(block, ids, W)
    got = gathered.numpy()
    assert np.array_equal(got[:, :NS], full["summary"])
    assert np.array_equal(got[:, NS:], full["mu"].reshape(W, -1))
    mu = np.transpose(full["mu"], (0, 2, 1)); A = np.transpose(full["A"], (0, 3, 2, 1))
    assert (np.diff(mu, axis=2) > 0).all() and np.max(np.abs(A.sum(axis=3) - 1)) < 1e-12
    assert np.max(np.abs(full["pif_final"].sum(axis=2) - 1)) < 1e-12 and np.isfinite(full["fcast"]).all()
    for w in (0, 777, 2047):
        o = oracle.estimate_window(Y[w], K, burnin, nrun, (12,), yreal[w], window_id=w)
        assert np.array_equal(full["x_final"][w], o["x_final"])
        assert np.max(np.abs(full["mu"][w].T - o["mu"])) < 1e-9 and np.max(np.abs(full["summary"][w] - o["summary"])) < 1e-9


def test_workspaces_survive_shape_changes(hmclib):
    """Grow-only workspaces: a small call after a large one and a large one after that give the same rows."""
    Y, Tw, fut = synth.generate_panel(64, 600, 3)
    big = _lib.estimate_batch_host(Y, Tw, 3, 3, 50, (12,), fut[:, 11:12])
    small = _lib.estimate_batch_host(Y[:3], Tw[:3], 3, 3, 50, (12,), fut[:3, 11:12])
    big2 = _lib.estimate_batch_host(Y, Tw, 3, 3, 50, (12,), fut[:, 11:12])
    assert np.array_equal(big["mu"][:3], small["mu"]) and np.array_equal(big["mu"], big2["mu"])
    _lib.load().hmcg_shutdown()                     # contexts are rebuilt on demand
    again = _lib.estimate_batch_host(Y[:3], Tw[:3], 3, 3, 50, (12,), fut[:3, 11:12])
    assert np.array_equal(again["mu"], small["mu"])


def test_shutdown_then_large_call_scatters_every_chunk(hmclib):
    """After hmcg_shutdown() the scatter helpers are new threads while the pool's generation counter has moved on: a
    fresh worker must neither run a phantom job nor be counted as done for one (`run()` could return while a
    helper was still copying into the caller's arrays).  Chunks above the 1 MiB hand-off threshold, twice, against the
    single-threaded scatter."""
    import os
    Y, Tw, fut = synth.generate_panel(96, 400, 3)
    args = (Y, Tw, 3, 2, 1200, (12,), fut[:, 11:12])           # 96 x 20 x 600 x 8 B = 9 MB in the first chunk
    ref = _lib.estimate_batch_host(*args)
    for _ in range(3):
        _lib.load().hmcg_shutdown()
        r = _lib.estimate_batch_host(*args)
        assert r["launches"] >= 3
        for k in ("mu", "sig2", "A", "pi_end", "fcast", "summary"):
            assert np.array_equal(r[k], ref[k]), k
