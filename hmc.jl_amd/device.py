"""Device-resident batched estimation: torch is used only for HBM allocation, streams
and torch.distributed plumbing; the sampling is libhmcgibbs (hmcg_estimate_batch_device).
"""
import ctypes as C

import numpy as np

from . import _lib


class DevicePanel:
    """A window panel resident in HBM plus its output buffers.

    Y (W, ldY) float64, T (W,), yreal (W, H); outputs in the C-ABI layouts
    (include/hmcg.h): mu/sig2/pi_end (W,K,nrun), A (W,K,K,nrun), fcast (W,2H,nrun),
    summary (W, 3K+K^2+2H), status (W,)."""

    def __init__(self, Y, T, K, nrun, horizons=(12,), yreal=None, device=0, window_ids=None, keep_draws=True, corr=False):
        import torch
        if not torch.cuda.is_available():
            raise _lib.HmcgError("no GPU visible to torch: hmc.jl_amd has no CPU fallback")
        _lib.load()
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.device_index = device
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.W, self.ldY = Y.shape
        self.K, self.nrun, self.horizons = int(K), int(nrun), tuple(horizons)
        H = len(self.horizons)
        self.NS = 3 * K + K * K + 2 * H
        self.max_T = int(np.max(T))
        Tv = np.asarray(T)[np.asarray(T) >= 2]
        self.min_T = int(Tv.min()) if Tv.size else 0      # hint for the length-bucketed dispatch (hmcg_config.min_T)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.Y = torch.from_numpy(Y).to(self.dev)
        self.T = torch.from_numpy(np.ascontiguousarray(T, dtype=np.int32)).to(self.dev)
        self.yreal = None if yreal is None else torch.from_numpy(
            np.ascontiguousarray(yreal, dtype=np.float64).reshape(self.W, H)).to(self.dev)
        self.window_ids = None if window_ids is None else torch.from_numpy(
            np.ascontiguousarray(window_ids, dtype=np.int64).astype(np.int32)).to(self.dev)
        W = self.W
        if keep_draws:
            self.mu = torch.zeros((W, K, nrun), **f64); self.sig2 = torch.zeros((W, K, nrun), **f64)
            self.A = torch.zeros((W, K, K, nrun), **f64); self.pi_end = torch.zeros((W, K, nrun), **f64)
            self.fcast = torch.zeros((W, 2 * H, nrun), **f64)
        else:
            self.mu = self.sig2 = self.A = self.pi_end = self.fcast = None
        self.summary = torch.zeros((W, self.NS), **f64)
        self.status = torch.zeros(W, dtype=torch.int32, device=self.dev)
        # extras.corr: (W, NC, NC) correlations between the per-draw outputs (calccorr), needs the draws on the device
        NC = 3 * K + K * K + 1
        self.corr = torch.zeros((W, NC, NC), **f64) if corr else None
        self.last_timing = None
        torch.cuda.synchronize(self.dev)   # fills above ran on torch's stream; the library uses its own

    @staticmethod
    def _ptr(t):
        return 0 if t is None else t.data_ptr()

    def run(self, burnin, seed=1234, window_base=0, threads_per_window=0, timed=True, stream=None, bucketed=True):
        """One batched estimate call (burnin + nrun sweeps per window).  With timed=True
        the call waits for completion and returns the HIP-event kernel time in ms.
        bucketed=False withholds the min_T hint: one launch sized for the longest window."""
        cfg = _lib.make_config(self.W, self.K, self.ldY, self.max_T, burnin, self.nrun, self.horizons, seed,
                               window_base, self.device_index, 0, threads_per_window,
                               min_T=self.min_T if bucketed else 0)
        ex = None
        if self.window_ids is not None or self.corr is not None:
            ex = _lib.Extras()
            ex.struct_size = C.sizeof(_lib.Extras)
            if self.window_ids is not None:
                ex.window_ids = self.window_ids.data_ptr()
            if self.corr is not None:
                ex.corr = self.corr.data_ptr()
        tm = _lib.estimate_batch_device(cfg, self.Y.data_ptr(), self.T.data_ptr(), self._ptr(self.yreal),
                                        self._ptr(self.mu), self._ptr(self.sig2), self._ptr(self.A),
                                        self._ptr(self.pi_end), self._ptr(self.fcast), self.summary.data_ptr(),
                                        self.status.data_ptr(), ex, stream, timed)
        self.last_timing = tm
        return tm.kernel_ms if tm is not None else None

    def sync(self):
        """Wait for the library stream (untimed runs are asynchronous)."""
        self.torch.cuda.synchronize(self.dev)
