"""Length-bucketed dispatch of ragged batches (csrc/hmcg.hip, make_plan / launch_kernel).

The reference's production run is 460 expanding windows of 120..579 months (code/run_hmm.jl:79-109; one SLURM task per
window, slurmscripts/base_estimation.sh:5).  Until round 3 one launch ran them all on the steps-per-thread variant of the
longest window; now every window runs on the variant its own length selects (one launch per length class, side by side).
What must hold:
  * a window's result does not depend on what else is in the call: bit-identical to a call with that window alone
    (same global window id), whatever the batch;
  * against the unbucketed launch (HMCG_NO_BUCKETS=1) the state paths are identical and every float agrees to 1e-9
    (a different steps-per-thread variant associates its scans differently: the last bits move, nothing else);
  * oracle parity on windows of each class (T = 120, 300, 579);
  * the device entry with the min_T hint runs the same classes as the host entry (bit-identical)."""
import numpy as np
import pytest

from hmc_jl_amd import _lib, device as hdev, synth

pytestmark = pytest.mark.gpu
TOL = 1e-9
K, HOR = 3, (1, 12)
NAMES = ("mu", "sig2", "A", "pi_end", "fcast", "summary")


def close(g, o):
    return float(np.max(np.abs(g - o) / (1.0 + np.abs(o)))) if g.size else 0.0


@pytest.fixture(scope="module")
def panel460():
    lens = list(range(120, 580))
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), K, ragged=lens)
    return Y, Tw, fut[:, [0, 11]]


def test_production_panel_bucketed_vs_single_launch(hmclib, panel460, monkeypatch):
    Y, Tw, yreal = panel460
    burnin, nrun = 3, 9
    g = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, HOR, yreal, want_state=True)
    assert g["buckets"] == 3 and g["steps_per_thread"] == 3          # classes 1, 2 and 3 steps per thread; the longest reports
    assert (g["status"] == 0).all()
    monkeypatch.setenv("HMCG_NO_BUCKETS", "1")
    u = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, HOR, yreal, want_state=True)
    assert u["buckets"] == 1 and u["steps_per_thread"] == 3
    assert np.array_equal(g["x_final"], u["x_final"]), "state paths differ between the bucketed and the single launch"
    for name in NAMES:
        assert close(g[name], u[name]) < TOL, name
    # the longest class runs the same variant in both: its windows are bit-identical
    top = Tw > 512
    for name in NAMES:
        assert np.array_equal(g[name][top], u[name][top]), name


def test_a_window_does_not_depend_on_its_batch(hmclib, oracle, panel460):
    Y, Tw, yreal = panel460
    burnin, nrun = 3, 9
    g = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, HOR, yreal, want_state=True)
    for w in (0, 136, 137, 180, 392, 393, 459):                       # T = 120, 256, 257, 300, 512, 513, 579: both sides of every cut
        T = int(Tw[w])
        s = _lib.estimate_batch_host(Y[w:w + 1, :T], Tw[w:w + 1], K, burnin, nrun, HOR, yreal[w:w + 1], want_state=True,
                                     window_ids=np.array([w]))
        assert s["steps_per_thread"] == (1 if T <= 256 else (2 if T <= 512 else 3))
        assert np.array_equal(s["x_final"][0, :T], g["x_final"][w, :T])
        for name in NAMES:
            assert np.array_equal(s[name][0], g[name][w]), (name, w)
    for w in (0, 180, 459):                                           # T = 120, 300, 579 against the oracle
        T = int(Tw[w])
        o = oracle.estimate_window(Y[w, :T], K, burnin, nrun, HOR, yreal[w], window_id=w)
        assert np.array_equal(g["x_final"][w, :T], o["x_final"]), "state path differs from the oracle (window %d)" % w
        assert close(g["mu"][w].T, o["mu"]) < TOL and close(g["sig2"][w].T, o["sig2"]) < TOL
        assert close(np.transpose(g["A"][w], (2, 1, 0)), o["A"]) < TOL
        assert close(g["pi_end"][w].T, o["pi_end"]) < TOL and close(g["fcast"][w].T, o["fcast"]) < TOL


def test_device_entry_with_the_min_T_hint(hmclib, panel460):
    import torch
    Y, Tw, yreal = panel460
    nrun = 8
    g = _lib.estimate_batch_host(Y, Tw, K, 0, nrun, HOR, yreal)
    p = hdev.DevicePanel(Y, Tw, K, nrun, HOR, yreal)
    p.run(burnin=0)
    assert p.last_timing.buckets == 3 and p.min_T == 120
    for name in ("mu", "sig2", "A", "pi_end", "fcast", "summary"):
        assert np.array_equal(getattr(p, name).cpu().numpy(), g[name]), name
    assert int((p.status != 0).sum().item()) == 0
    p.run(burnin=0, bucketed=False)                                   # no hint: one launch, as before
    assert p.last_timing.buckets == 1 and p.last_timing.steps_per_thread == 3
    assert close(p.mu.cpu().numpy(), g["mu"]) < TOL and close(p.summary.cpu().numpy(), g["summary"]) < TOL
    torch.cuda.synchronize()


def test_windows_outside_the_hint_and_bad_lengths(hmclib):
    """A hint that is wrong only changes which variant a window runs on; T < 2 and T > max_T are flagged once."""
    lens = [300, 90, 600, 1, 450]
    Y, Tw, fut = synth.generate_panel(len(lens), 600, K, ragged=[max(l, 2) for l in lens])
    Tw = np.array(lens, dtype=np.int32)
    g = _lib.estimate_batch_host(Y, Tw, K, 1, 4, HOR, fut[:, [0, 11]], nan_fill=False)
    assert g["status"][3] == _lib.ST_BAD_T and (np.delete(g["status"], 3) == 0).all()
    assert g["buckets"] == 3 and not g["mu"][3].any()


def test_shuffled_batch_equals_sorted_batch_on_both_entries(hmclib, panel460, monkeypatch):
    """The order of the windows in a call is the caller's business: a shuffled batch (every length class scattered over the
    grid -- the launches then run their compacted window lists, KernelParams::order) gives every window the bits it has in
    the sorted batch (classes in one run each: block b = window b), on the host entry and on the device entry; and so does
    a sorted batch with the compaction switched off (HMCG_NO_BUCKET_LISTS)."""
    Y, Tw, yreal = panel460
    W = Y.shape[0]
    burnin, nrun = 2, 7
    ids = np.arange(W)
    ref = _lib.estimate_batch_host(Y, Tw, K, burnin, nrun, HOR, yreal, want_state=True, window_ids=ids)
    assert ref["buckets"] == 3 and (ref["status"] == 0).all()
    perm = np.random.default_rng(5).permutation(W)
    g = _lib.estimate_batch_host(np.ascontiguousarray(Y[perm]), Tw[perm], K, burnin, nrun, HOR, np.ascontiguousarray(yreal[perm]),
                                 want_state=True, window_ids=ids[perm])
    assert g["buckets"] == 3
    for k in NAMES + ("x_final", "status"):
        assert np.array_equal(g[k], ref[k][perm]), k
    p = hdev.DevicePanel(np.ascontiguousarray(Y[perm]), Tw[perm], K, nrun, HOR, np.ascontiguousarray(yreal[perm]), window_ids=ids[perm])
    p.run(burnin=burnin)
    assert p.last_timing.buckets == 3
    assert np.array_equal(p.mu.cpu().numpy(), ref["mu"][perm]) and np.array_equal(p.summary.cpu().numpy(), ref["summary"][perm])
    assert np.array_equal(p.status.cpu().numpy(), ref["status"][perm])
    monkeypatch.setenv("HMCG_NO_BUCKET_LISTS", "1")
    h = _lib.estimate_batch_host(np.ascontiguousarray(Y[perm]), Tw[perm], K, burnin, nrun, HOR, np.ascontiguousarray(yreal[perm]),
                                 want_state=True, window_ids=ids[perm])
    for k in NAMES + ("x_final", "status"):
        assert np.array_equal(h[k], g[k]), k
