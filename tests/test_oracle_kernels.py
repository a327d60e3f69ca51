"""Known-answer tests of the oracle's deterministic kernels against independent numpy
restatements written here (textbook forward-backward), tiny hand-checkable cases included."""
import numpy as np


def normpdf(mu, sd, x):
    return np.exp(-0.5 * ((x - mu) / sd) ** 2) / (sd * np.sqrt(2 * np.pi))


def textbook_forward(Y, mu, sig2, rho, A):
    sd = np.sqrt(sig2)
    a = rho.copy()
    out = []
    for y in Y:
        a = (a @ A) * normpdf(mu, sd, y)     # the reference applies A once before the first emission (quirk 3)
        a = a / a.sum()
        out.append(a)
    return np.array(out)


def textbook_smoother(Y, mu, sig2, rho, A):
    """Scaled forward-backward: gamma_t = alpha_t * beta_t / sum."""
    sd = np.sqrt(sig2)
    al = textbook_forward(Y, mu, sig2, rho, A)
    T, K = al.shape
    be = np.ones((T, K))
    for t in range(T - 2, -1, -1):
        be[t] = A @ (normpdf(mu, sd, Y[t + 1]) * be[t + 1])
        be[t] /= be[t].sum()
    g = al * be
    return g / g.sum(axis=1, keepdims=True)


def test_forward_hand_case(oracle):
    # K=2, T=3, worked by hand: sd=1, mu=(0,2), A=[[.9,.1],[.2,.8]], rho=(.5,.5), Y=(0,2,2)
    mu = np.array([0.0, 2.0]); sig2 = np.array([1.0, 1.0]); rho = np.array([0.5, 0.5])
    A = np.array([[0.9, 0.1], [0.2, 0.8]]); Y = np.array([0.0, 2.0, 2.0])
    pif, st = oracle.forward_filter(Y, mu, sig2, rho, A)
    e = np.exp(-2.0)
    # step 1: prior (.55,.45); likelihoods (1, e) up to the common constant
    p1 = np.array([0.55 * 1.0, 0.45 * e]); p1 /= p1.sum()
    pr2 = p1 @ A
    p2 = pr2 * np.array([e, 1.0]); p2 /= p2.sum()
    pr3 = p2 @ A
    p3 = pr3 * np.array([e, 1.0]); p3 /= p3.sum()
    assert st == 0
    np.testing.assert_allclose(pif, np.array([p1, p2, p3]), rtol=1e-13)


def test_forward_and_smoother_vs_textbook(oracle):
    rng = np.random.default_rng(5)
    for K, T in ((2, 50), (3, 400), (5, 120)):
        A = rng.dirichlet(np.ones(K) * 3, size=K)
        mu = np.sort(rng.normal(0, 4, K)); sig2 = rng.uniform(0.3, 2.0, K); rho = rng.dirichlet(np.ones(K))
        Y = rng.normal(0, 4, T)
        pif, Pf, st = oracle.forward_filter(Y, mu, sig2, rho, A, want_P=True)
        np.testing.assert_allclose(pif, textbook_forward(Y, mu, sig2, rho, A), rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(Pf.sum(axis=(1, 2)), 1.0, rtol=1e-12)
        np.testing.assert_allclose(Pf.sum(axis=1), pif, rtol=1e-12, atol=1e-300)    # pif[t,s] = sum_r Pf[t,r,s]
        pib = oracle.backward_smoother(pif, Pf)
        np.testing.assert_allclose(pib[-1], pif[-1], rtol=0, atol=0)                 # src/Hmc.jl:448
        np.testing.assert_allclose(pib, textbook_smoother(Y, mu, sig2, rho, A), rtol=1e-8, atol=1e-12)


def test_forward_underflow_flag(oracle):
    mu = np.array([0.0, 1.0]); sig2 = np.array([1e-4, 1e-4]); rho = np.array([0.5, 0.5]); A = np.full((2, 2), 0.5)
    pif, st = oracle.forward_filter(np.array([0.0, 50.0, 1.0]), mu, sig2, rho, A)
    assert st & 2
    np.testing.assert_allclose(pif[1], pif[0] @ A)          # the underflowed observation is treated as missing
    assert np.isfinite(pif).all()


def test_forecast(oracle):
    rng = np.random.default_rng(2)
    for K in (2, 3, 6):
        A = rng.dirichlet(np.ones(K), size=K); mu = rng.normal(0, 3, K); pe = rng.dirichlet(np.ones(K))
        for h in (0, 1, 2, 3, 12, 24, 37):
            ref = pe @ np.linalg.matrix_power(A, h) @ mu
            assert abs(oracle.forecast(mu, A, pe, h) - ref) < 1e-12 * (1 + abs(ref))
        assert abs(oracle.forecast(mu, np.eye(K), pe, 12) - pe @ mu) < 1e-15 * (1 + abs(pe @ mu))


def test_teacher_forced_sweep_statistics(oracle):
    """One sweep from a fixed X: the conjugate updates must centre on their posterior means."""
    rng = np.random.default_rng(3)
    T, K = 600, 3
    X = rng.integers(0, K, T)
    mus = np.array([-3.0, 1.0, 6.0]); Y = mus[X] + rng.normal(0, 0.7, T)
    draws = [oracle.estimate_window(Y, K, 0, 1, horizons=(), seed=s, x_init=X) for s in range(300)]
    m = np.mean([d["mu"][0] for d in draws], axis=0)
    v = np.mean([d["sig2"][0] for d in draws], axis=0)
    # exact conditional posterior means given X (src/Hmc.jl:313-314,331; beta=1 on the first sweep, alpha=nu=1, xi=mean(Y))
    xi = Y.mean()
    for i in range(K):
        yi = Y[X == i]
        N = len(yi)
        a = 1.0 + 0.5 * N
        b = 1.0 + 0.5 * ((yi - yi.mean()) ** 2).sum() + 0.5 * N / (N + 1.0) * (yi.mean() - xi) ** 2
        e_sig = b / (a - 1.0)
        assert abs(v[i] - e_sig) < 5 * e_sig / np.sqrt(a - 2.0) / np.sqrt(300), (i, v[i], e_sig)
        e_mu = (yi.sum() + xi) / (N + 1.0)
        assert abs(m[i] - e_mu) < 5 * np.sqrt(e_sig / (N + 1.0)) / np.sqrt(300), (i, m[i], e_mu)
    Am = np.mean([d["A"][0] for d in draws], axis=0)
    C = np.ones((K, K))
    for a, b in zip(X[:-1], X[1:]):
        C[a, b] += 1
    np.testing.assert_allclose(Am, C / C.sum(axis=1, keepdims=True), atol=0.02)


def test_signals_past_the_end_date_reporting(oracle):
    """sigLen > 0 (src/Hmc.jl:888-910): the reported probabilities are the backward smoother's row at endIndex, the
    blended forecast is a*signal + (1-a)*(pi_last . mu) with a = 1/(1 + sigma_signal), and neither changes the chain."""
    from hmc_jl_amd import synth
    K, T, sigLen, ssig = 3, 240, 5, 0.7
    Y, _, fut = synth.generate_panel(1, T, K)
    y = Y[0]
    kw = dict(sig=(T - sigLen, T), kappa=0.5, alpha=2.0, nu=2.0, sigma_signal=ssig, save=(T - 1, T), seed=77)
    plain = oracle.estimate_signals(y, K, 5, 20, 2, horizons=(0, 7), yreal=[0.0, 1.0], **kw)
    rep = oracle.estimate_signals(y, K, 5, 20, 2, horizons=(0, 7), yreal=[0.0, 1.0], end_pos=T - 1 - sigLen, blend_mask=1, **kw)
    for k in ("mu", "sig2", "A", "x_final", "sigvals"):
        assert np.array_equal(plain[k], rep[k]), k
    assert np.array_equal(plain["fcast"][:, 2:], rep["fcast"][:, 2:])          # the unblended horizon is untouched
    a = (1.0 / ssig) / (1.0 + 1.0 / ssig)
    ysig = np.repeat(rep["sigvals"][:, 0], 20)                                  # Yfake[T-1] of each noise sample
    assert np.max(np.abs(rep["fcast"][:, 0] - (a * ysig + (1 - a) * plain["fcast"][:, 0]))) < 1e-12
    assert np.max(np.abs(rep["pi_end"].sum(axis=1) - 1)) < 1e-12
    assert np.max(np.abs(rep["pi_end"] - plain["pi_end"])) > 1e-3              # smoothed at endIndex != filtered at the end
    # the smoother row itself: rerun one sweep's filter + smoother from the final parameters is not available
    # draw by draw, so check the defining property on the last draw's neighbours instead: with sigLen = 0 both agree
    same = oracle.estimate_signals(y, K, 5, 20, 2, horizons=(0, 7), yreal=[0.0, 1.0], end_pos=T - 1, **kw)
    assert np.array_equal(same["pi_end"], plain["pi_end"])
