"""Accuracy of the lean device math (csrc/rp_math.h) against NumPy, measured in ulp."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "commonroad-reactive-planner_amd", "lib", "librp_mathtest.so")


def _run(kind, x):
    lib = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    lib.rpt_math.argtypes = [C.c_int, C.c_int, dp, dp, dp]
    x = np.ascontiguousarray(x, dtype=np.float64)
    a, b = np.empty_like(x), np.empty_like(x)
    assert lib.rpt_math(kind, len(x), x.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp)) == 0
    return a, b


def _run_n(kind, packed, n):
    lib = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    lib.rpt_math.argtypes = [C.c_int, C.c_int, dp, dp, dp]
    packed = np.ascontiguousarray(packed, dtype=np.float64)
    a, b = np.empty(n), np.empty(n)
    assert lib.rpt_math(kind, n, packed.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp)) == 0
    return a, b


def _ulp(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_atan():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-0.45, 0.45, 200000), rng.uniform(-3, 3, 200000), rng.normal(size=50000) * 1e3,
                        10.0 ** rng.uniform(-300, 300, 20000), [0.0, -0.0, 0.4375, 0.6875, 1.1875, 2.4375, 1e-320, np.inf, -np.inf]])
    a, _ = _run(0, x)
    ref = np.arctan(x)
    nz = ref != 0
    assert _ulp(a[nz], ref[nz]).max() <= 1.0
    assert np.all(a[~nz] == 0)
    assert np.isnan(_run(0, np.array([np.nan]))[0][0])


def test_sincos():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, 400000), rng.uniform(-1e3, 1e3, 100000),
                        np.arange(-16, 17) * (np.pi / 4), [0.0, -0.0, 1e-300]])
    s, c = _run(1, x)
    rs, rc = np.sin(x), np.cos(x)
    # absolute error bound 1e-16-ish near zeros of sin/cos (Cody-Waite with a 3-part pi/2), ulp elsewhere
    assert np.max(np.abs(s - rs)) <= 2.3e-16 and np.max(np.abs(c - rc)) <= 2.3e-16
    big = np.abs(rs) > 1e-3
    assert _ulp(s[big], rs[big]).max() <= 1.0
    big = np.abs(rc) > 1e-3
    assert _ulp(c[big], rc[big]).max() <= 1.0


def test_rcp_rsqrt():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0.01, 100, 300000), 10.0 ** rng.uniform(-100, 100, 50000)])
    r, _ = _run(2, x)
    assert _ulp(r, 1.0 / x).max() <= 1.0
    r, _ = _run(2, -x)
    assert _ulp(r, -1.0 / x).max() <= 1.0
    q, _ = _run(3, x)
    assert _ulp(q, 1.0 / np.sqrt(x)).max() <= 2.0


def test_extension_travel_closed_form():
    """ext_travel (closed form of dt * sum_{t=1..m} max(0, v + t dt a), used by the bounds of the collision broad
    phase) against the loop it replaces (CartesianSample.enlarge, trajectories.py:182-196)."""
    rng = np.random.default_rng(5)
    n = 200000
    v = np.concatenate([rng.uniform(-3, 25, n // 2), rng.uniform(-0.01, 0.01, n // 4), np.zeros(n // 4)])
    a = np.concatenate([rng.uniform(-12, 12, n // 2), rng.uniform(-1e-3, 1e-3, n // 4), rng.uniform(-12, 12, n // 8), np.zeros(n // 8)])
    m = rng.integers(0, 130, n).astype(float)
    dt = rng.choice([0.05, 0.1, 0.2], n)
    packed = np.concatenate([v, a, m, dt])
    got, _ = _run_n(4, packed, n)
    ref = np.zeros(n)
    for t in range(1, 130):
        vt = v + (t * dt) * a
        ref += np.where(t <= m, dt * np.maximum(vt, 0.0), 0.0)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # monotone in (v, a): what pair_step_bound relies on when it evaluates the extremes of the pair
    got2, _ = _run_n(4, np.concatenate([v + 0.3, a + 0.2, m, dt]), n)
    assert np.all(got2 >= got - 1e-12)
