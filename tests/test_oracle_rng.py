"""Oracle RNG spec: Philox4x32-10 known answers (Random123 kat_vectors) and sampler moments."""
import numpy as np


def test_philox_known_answers(oracle):
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_range_and_streams(oracle):
    u = np.array([oracle.uniform_x(1234, 0, s, t) for s in range(20) for t in range(100)])
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    # counter-based: same (seed, window, sweep, t) -> same value; any coordinate change -> different
    assert oracle.uniform_x(1234, 3, 7, 11) == oracle.uniform_x(1234, 3, 7, 11)
    base = oracle.uniform_x(1234, 3, 7, 11)
    assert len({base, oracle.uniform_x(1235, 3, 7, 11), oracle.uniform_x(1234, 4, 7, 11),
                oracle.uniform_x(1234, 3, 8, 11), oracle.uniform_x(1234, 3, 7, 12)}) == 5


def test_gamma_moments(oracle):
    n = 20000
    for shape in (0.5, 1.0, 1.5, 4.0, 60.5, 400.0):
        g = np.array([oracle.gamma(99, 0, s, 0, 0, shape) for s in range(n)])
        assert (g > 0).all()
        se = np.sqrt(shape / n)
        assert abs(g.mean() - shape) < 5 * se, (shape, g.mean())
        assert abs(g.var() / shape - 1) < 0.08, (shape, g.var())


def test_normal_moments(oracle):
    z = np.array([oracle.normal(7, 1, s, 1, 0) for s in range(40000)])
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.03
    assert abs(np.mean(z ** 3)) < 0.06 and abs(np.mean(z ** 4) - 3) < 0.15
