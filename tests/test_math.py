"""include/pdmpc_math.h (the sin/cos shared by oracle and kernel) against libm."""
import numpy as np

from oracle import oracle


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64).view(np.int64)
    b = np.asarray(b, dtype=np.float64).view(np.int64)
    a = np.where(a < 0, np.int64(-(2**63)) - a, a)
    b = np.where(b < 0, np.int64(-(2**63)) - b, b)
    return np.abs(a - b)


def test_sincos_within_one_ulp_of_libm():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.uniform(-np.pi, np.pi, 200000),
        rng.uniform(-100, 100, 200000),
        rng.uniform(-1e5, 1e5, 100000),
        np.linspace(-7, 7, 20001),
        np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 1e-300, 1e-10, 2.0**-27, 2.0**-28]),
    ])
    s, c = oracle.sincos(x)
    assert ulp_diff(s, np.sin(x)).max() <= 1
    assert ulp_diff(c, np.cos(x)).max() <= 1
    assert np.abs(s * s + c * c - 1).max() < 4e-16


def test_sincos_special_values():
    s, c = oracle.sincos(np.array([0.0, -0.0, np.inf, -np.inf, np.nan]))
    assert s[0] == 0 and c[0] == 1 and np.signbit(s[1]) and c[1] == 1
    assert np.isnan(s[2:]).all() and np.isnan(c[2:]).all()


def test_sincos_symmetry():
    x = np.random.default_rng(1).uniform(-50, 50, 10000)
    s1, c1 = oracle.sincos(x)
    s2, c2 = oracle.sincos(-x)
    assert np.array_equal(s1, -s2) and np.array_equal(c1, c2)
