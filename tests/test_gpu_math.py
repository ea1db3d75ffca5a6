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
