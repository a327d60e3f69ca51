"""Seeded synthetic Gaussian-HMM panels.

Same structure as the reference's generateData (src/Hmc.jl:210-229): X_1 = state 1,
X_t ~ Categorical(A[X_{t-1}, :]), Y_t ~ Normal(mu[X_t], sqrt(sig2[X_t])) -- but on
our own generator (the reference's is tied to Julia's MersenneTwister): numpy's
Philox bit generator, raw 64-bit outputs only (stable across numpy versions), with
explicit inverse-CDF / Box-Muller transforms.  Window w uses seed 20240000 + w.
"""
import numpy as np

K3 = dict(A=np.array([[.92, .05, .03], [.04, .92, .04], [.03, .05, .92]]),
          mu=np.array([2.0, 4.5, 9.0]), sig2=np.array([0.5, 0.7, 4.0]))


def k8_params():
    K = 8
    return dict(A=0.86 * np.eye(K) + 0.02 * (1 - np.eye(K)), mu=2.0 * np.arange(1, K + 1), sig2=np.full(K, 0.5))


def params_for(K):
    if K == 3:
        return K3
    if K == 8:
        return k8_params()
    if K == 2:  # the reference unit test's truth (test/runtests.jl:23-26)
        return dict(A=np.array([[0.5, 0.5], [0.2, 0.8]]), mu=np.array([-5.0, 4.0]), sig2=np.array([1.0, 0.5]))
    A = np.full((K, K), 0.1 / (K - 1)) + (0.9 - 0.1 / (K - 1)) * np.eye(K)
    return dict(A=A, mu=2.5 * np.arange(1, K + 1), sig2=np.full(K, 0.6))


def _uniforms(seed, n):
    raw = np.random.Philox(int(seed)).random_raw(n)
    return (raw >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def generate_window(T, K=3, seed=20240000, params=None):
    """Returns (Y[T], X[T] 0-based)."""
    p = params or params_for(K)
    A, mu, sig2 = p["A"], p["mu"], p["sig2"]
    u = _uniforms(seed, 3 * T)
    cdf = np.cumsum(A, axis=1)
    X = np.zeros(T, dtype=np.int64)
    for t in range(1, T):
        X[t] = min(int(np.searchsorted(cdf[X[t - 1]], u[t], side="right")), K - 1)
    z = np.sqrt(-2.0 * np.log(1.0 - u[T:2 * T])) * np.cos(2.0 * np.pi * u[2 * T:3 * T])
    Y = mu[X] + np.sqrt(sig2[X]) * z
    return Y, X


def _generate_same_length(seeds, T, K, params=None):
    """generate_window for many seeds at once (same length): identical values, vectorised over the windows."""
    p = params or params_for(K)
    A, mu, sig2 = p["A"], p["mu"], p["sig2"]
    n = len(seeds)
    U = np.stack([_uniforms(sd, 3 * T) for sd in seeds])          # (n, 3T)
    cdf = np.cumsum(A, axis=1)
    X = np.zeros((n, T), dtype=np.int64)
    for t in range(1, T):
        X[:, t] = np.minimum((cdf[X[:, t - 1]] <= U[:, t, None]).sum(axis=1), K - 1)   # searchsorted(..., side="right")
    z = np.sqrt(-2.0 * np.log(1.0 - U[:, T:2 * T])) * np.cos(2.0 * np.pi * U[:, 2 * T:3 * T])
    return mu[X] + np.sqrt(sig2[X]) * z


def generate_panel(W, T, K=3, horizon_pad=12, window_base=0, ragged=None):
    """Panel for the benchmark configs: W windows of length T (+horizon_pad extra
    points per window, returned separately as the realised future values).

    ragged: optional array of per-window lengths (<= T).
    Returns Y (W, T) float64 zero-padded, Tw (W,) int32, future (W, horizon_pad)."""
    Y = np.zeros((W, T))
    fut = np.zeros((W, horizon_pad))
    Tw = np.full(W, T, dtype=np.int32) if ragged is None else np.asarray(ragged, dtype=np.int32)
    for length in np.unique(Tw):
        ws = np.nonzero(Tw == length)[0]
        if len(ws) == 1:
            y = generate_window(int(length) + horizon_pad, K, 20240000 + window_base + int(ws[0]))[0][None, :]
        else:
            y = _generate_same_length([20240000 + window_base + int(w) for w in ws], int(length) + horizon_pad, K)
        Y[ws, :length] = y[:, :length]
        fut[ws] = y[:, length:length + horizon_pad]
    return Y, Tw, fut
