"""hmcg_estimate_batch_multi with G > 1 for everything the multi entry accepts, on four VIRTUAL devices
(HMCG_VIRTUAL_DEVICES=4: four device ids, each with its own context -- streams, arenas, scatter helpers -- on the one
physical GPU of the box).  Round 3 ran the K = 3 base path this way (tests/test_gpu_cdriver.py); here: the LDS-resident
kernel (K = 8, T = 5000: per-context pdf scratch, per-device hipFuncSetAttribute), its HBM-streaming form, the signal
path with sample_summary and sigvals, the smoothed means and per-draw smoothed probabilities, and extras.corr.  Every
output must equal the single-device call bit for bit (windows keep their global RNG ids and their own length class
whatever the partition).  Reference role: the SLURM fan-out, slurmscripts/base_estimation.sh:5,17."""
import numpy as np
import pytest

from hmc_jl_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEVS = [0, 1, 2, 3]
SKIP = {"kernel_ms", "call_ms", "per_device", "launches", "threads_per_window", "steps_per_thread", "lds_bytes", "helper_waves", "buckets", "streaming"}


def both(monkeypatch, *args, **kw):
    one = _lib.estimate_batch_host(*args, **kw)
    monkeypatch.setenv("HMCG_VIRTUAL_DEVICES", "4")
    four = _lib.estimate_batch_host(*args, devices=DEVS, **kw)
    monkeypatch.delenv("HMCG_VIRTUAL_DEVICES")
    assert sorted(d["device"] for d in four["per_device"]) == DEVS and all(d["windows"] > 0 for d in four["per_device"])
    for k, v in one.items():
        if k in SKIP:
            continue
        assert np.array_equal(v, four[k], equal_nan=True), k
    assert (one["status"] == 0).all()
    return one, four


def test_lds_resident_kernel_k8_t5000(hmclib, monkeypatch):
    lens = [5000, 4800, 5000, 3100, 4999, 2600, 5000]
    Y, Tw, fut = synth.generate_panel(len(lens), 5000, 8, ragged=lens)
    one, four = both(monkeypatch, Y, Tw, 8, 2, 6, (1, 12), fut[:, [0, 11]], want_state=True)
    assert four["steps_per_thread"] == 20 and four["threads_per_window"] == 256


def test_streaming_form(hmclib, monkeypatch):
    lens = [9000, 7700, 8200, 9000, 8999]                     # beyond the LDS at K = 3: the HBM-streaming form by itself
    Y, Tw, fut = synth.generate_panel(len(lens), 9000, 3, ragged=lens)
    one, four = both(monkeypatch, Y, Tw, 3, 1, 4, (12,), fut[:, 11:12], want_state=True)
    assert four["streaming"]                                 # per-step arrays are not in the LDS
    monkeypatch.setenv("HMCG_FORCE_STREAM", "1")             # and forced on a K = 8 batch
    lens = [1500, 1200, 900, 1499, 1000]
    Y, Tw, fut = synth.generate_panel(len(lens), 1500, 8, ragged=lens)
    both(monkeypatch, Y, Tw, 8, 1, 4, (12,), fut[:, 11:12], want_state=True)


def test_signal_path_with_sample_summary_and_sigvals(hmclib, monkeypatch):
    lens = [400, 333, 520, 260, 579, 128, 300, 450, 200]      # three length classes: bucketed on every device
    W = len(lens)
    Y, Tw, fut = synth.generate_panel(W, max(lens), 3, ragged=lens)
    sig = np.stack([Tw - 25, Tw], axis=1).astype(np.int32)
    save = np.stack([Tw - 3, Tw], axis=1).astype(np.int32)
    ssig = np.linspace(0.3, 1.1, W)
    one, four = both(monkeypatch, Y, Tw, 3, 3, 7, (1, 12), fut[:, [0, 11]], sig_range=sig, save_range=save, sigma_signal=ssig,
                     kappa=0.6, n_samples=4, alpha=2.0, nu=2.0, want_sample_summary=True, want_state=True)
    assert one["sample_summary"].shape == (W, 4, 3 * 3 + 9 + 4) and np.isfinite(four["sigvals"]).all()
    # ... and on the LDS-resident kernel (K = 6)
    Y, Tw, fut = synth.generate_panel(W, max(lens), 6, ragged=lens)
    both(monkeypatch, Y, Tw, 6, 2, 5, (12,), fut[:, 11:12], sig_range=sig, save_range=save, sigma_signal=ssig,
         kappa=0.4, n_samples=3, alpha=2.0, nu=2.0, want_sample_summary=True)


def test_smoothed_means_and_per_draw_smoothed_probabilities(hmclib, monkeypatch):
    lens = [300, 512, 140, 700, 256, 257, 90]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    one, four = both(monkeypatch, Y, Tw, 3, 2, 9, (12,), fut[:, 11:12], want_smooth=True, want_filter_mean=True, want_smooth_draws=True)
    for w, T in enumerate(lens):
        assert np.abs(four["pi_smooth_mean"][w, :T].sum(axis=1) - 1).max() < 1e-12
        assert np.abs(four["pi_smooth_draws"][w, :, :T, :].sum(axis=0) - 1).max() < 1e-12
    lens = [2000, 1700, 1999, 1234]                           # K = 8: the smoothing variant of the LDS-resident kernel
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 8, ragged=lens)
    both(monkeypatch, Y, Tw, 8, 1, 5, (12,), fut[:, 11:12], want_smooth=True, want_smooth_draws=True)


def test_correlation_matrices(hmclib, monkeypatch):
    lens = [300, 260, 579, 120, 400, 350, 510, 515]
    Y, Tw, fut = synth.generate_panel(len(lens), max(lens), 3, ragged=lens)
    monkeypatch.setenv("HMCG_CHUNK_DRAWS", "150")            # several chunks per device: the moments accumulate across them
    one, four = both(monkeypatch, Y, Tw, 3, 10, 600, (12,), fut[:, 11:12], want_corr=True)
    assert np.isfinite(four["corr"][:, 0, 0]).all() and np.abs(np.diagonal(four["corr"], axis1=1, axis2=2)[np.isfinite(np.diagonal(four["corr"], axis1=1, axis2=2))] - 1).max() == 0
