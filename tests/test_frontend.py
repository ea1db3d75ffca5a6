"""Reference-path front end: rp_build_reference / rp_project / rp_initial_state (host-only entry points of librp_amd.so,
csrc/rp_frontend.h) and the oracle's NumPy restatement (oracle/frontend.py), both against fixtures produced by the
reference's own CoordinateSystem.__init__ and ReactivePlanner._compute_initial_states
(tests/golden/make_frontend_golden.py; utils_coordinate_system.py:74-118, reactive_planner.py:446-512).

Tolerances: vertices / arc length / orientation 1e-9, curvature 1e-9, curvature rate 1e-8 (second and third differences of a
spline that SciPy and the C++ restatement solve by different eliminations), Frenet states 1e-8 relative to 1 + |value|."""
import glob
import os

import numpy as np
import pytest

from commonroad_rp_amd import _capi
from oracle import frontend as ofe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "frontend_*.npz")))


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def test_fixture_inventory():
    assert len(CASES) >= 8 and any(c.endswith("_raw") for c in CASES)


def _check_tables(got, z):
    ref, pos, th, cu, cud = got
    assert ref.shape == z["reference"].shape
    np.testing.assert_allclose(ref, z["reference"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pos, z["ref_pos"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(th, z["ref_theta"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(cu, z["ref_curv"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(cud, z["ref_curv_d"], rtol=0, atol=1e-8)


def _check_states(fn, z):
    for st, lon, lat, code in zip(z["states"], z["x0_lon"], z["x0_lat"], z["code"]):
        x, y, th, v, a, steer, low = st
        if code == -5:
            with pytest.raises(ValueError):
                fn(x, y, th, v, a, steer, bool(low))
        elif code == -6:
            with pytest.raises(Exception) as e:
                fn(x, y, th, v, a, steer, bool(low))
            assert not isinstance(e.value, ValueError)
        else:
            got_lon, got_lat = fn(x, y, th, v, a, steer, bool(low))
            np.testing.assert_allclose(got_lon, lon, rtol=1e-8, atol=1e-8)
            np.testing.assert_allclose(got_lat, lat, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("name", CASES)
def test_oracle_restatement_matches_the_reference(name):
    z = _load(name)
    tables = ofe.build_reference(z["polyline"], smooth=bool(int(z["smooth"])))
    _check_tables(tables, z)
    wb = float(z["wheelbase"])
    _check_states(lambda x, y, th, v, a, st, low: ofe.initial_state(tables, x, y, th, v, a, st, wb, low), z)


@pytest.mark.parametrize("name", CASES)
def test_native_front_end_matches_the_reference(name):
    z = _load(name)
    tables = _capi.build_reference(z["polyline"], smooth=bool(int(z["smooth"])))
    _check_tables(tables, z)
    wb = float(z["wheelbase"])
    _check_states(lambda x, y, th, v, a, st, low: _capi.initial_state(*tables, x, y, th, v, a, st, wb, low), z)
    # (s, d) of every state the reference could project, against the oracle's projection
    ref, pos = tables[0], tables[1]
    for st, code in zip(z["states"], z["code"]):
        if code == -5:
            with pytest.raises(ValueError):
                _capi.project(ref, pos, st[0], st[1])
        else:
            np.testing.assert_allclose(_capi.project(ref, pos, st[0], st[1]), ofe.project(ref, pos, st[0], st[1]), rtol=0, atol=1e-9)


def test_known_answers_straight_line_and_circle():
    # straight line, heading atan(3/4), 0.5 m vertices, every vertex twice: duplicates go, 1 m resampling, zero curvature
    s = np.repeat(np.arange(0.0, 60.5, 0.5), 2)
    ref, pos, th, cu, cud = _capi.build_reference(np.stack((2.0 + 0.8 * s, -1.0 + 0.6 * s), axis=1))
    assert len(pos) == 61
    np.testing.assert_allclose(pos, np.arange(61.0), atol=1e-9)
    np.testing.assert_allclose(ref, np.stack((2.0 + 0.8 * pos, -1.0 + 0.6 * pos), axis=1), atol=1e-9)
    np.testing.assert_allclose(th, np.arctan2(0.6, 0.8), atol=1e-9)
    np.testing.assert_allclose(cu, 0.0, atol=1e-9)
    np.testing.assert_allclose(cud, 0.0, atol=1e-9)
    # a point 1.5 m left of the line at arc length 20.25 m
    x, y = 2.0 + 0.8 * 20.25 - 0.6 * 1.5, -1.0 + 0.6 * 20.25 + 0.8 * 1.5
    np.testing.assert_allclose(_capi.project(ref, pos, x, y), [20.25, 1.5], atol=1e-9)
    lon, lat = _capi.initial_state(ref, pos, th, cu, cud, x, y, np.arctan2(0.6, 0.8) + 0.1, 10.0, 0.5, 0.0, 2.5, False)
    np.testing.assert_allclose(lon, [20.25, 10.0 * np.cos(0.1), 0.5 * np.cos(0.1)], atol=1e-9)   # straight reference: s'' = a cos(theta)
    np.testing.assert_allclose(lat[:2], [1.5, 10.0 * np.sin(0.1)], atol=1e-9)
    # quarter circle of radius 80 m (left turn), 0.25 m vertices: curvature 1 / 80 away from the ends, heading s / R
    R = 80.0
    a = np.arange(0.0, 0.5 * np.pi * R, 0.25) / R
    ref, pos, th, cu, cud = _capi.build_reference(np.stack((R * np.sin(a), R * (1.0 - np.cos(a))), axis=1))
    rad = np.hypot(ref[:, 0], ref[:, 1] - R)   # the spline is sampled 200 times and resampled linearly: chords of 0.63 m, sagitta 6e-4 m
    assert np.all(rad <= R + 1e-6) and np.all(rad >= R - 1e-3)
    np.testing.assert_allclose(cu[3:-3], 1.0 / R, rtol=2e-2)   # (second differences over a linearly resampled polyline: +-1 %)
    np.testing.assert_allclose(th[:-2], (pos[:-2] + 0.5) / R, atol=2e-3)                      # heading of a 1 m chord = angle at its middle
    assert np.all(np.diff(th) >= -1e-12)
    # unusable input
    with pytest.raises(ValueError):
        _capi.build_reference(np.array([[0.0, 0.0], [0.0, 0.0], [0.0, 0.0]]))
    with pytest.raises(ValueError):
        _capi.build_reference(np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0]]), smooth=True)    # splprep needs m > k
    assert len(_capi.build_reference(np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0]]), smooth=False)[1]) == 3


def test_projection_on_a_nearly_straight_stretch_is_the_root():
    """The foot point solves ((P - p0) - lam e) . ((1 - lam) t0 + lam t1) = 0 per segment: a quadratic in lam whose leading coefficient
    vanishes as the stretch straightens.  With the textbook root formula both the oracle and the library were 5e-5 m off in s there
    (and apart from each other: the C++ side's fma rounds the cancelling difference another way) -- found by the live sweep of round 5
    (tests/sweeps/live_frontend.py).  Here: a polyline that bends by 1e-12 rad per metre, far from the origin; the returned (s, d) must
    satisfy the defining equation, and the two implementations must agree."""
    i = np.arange(0.0, 12.0)
    ref = np.stack((4000.0 + i, -3000.0 + 0.5e-12 * i * i), axis=1)
    pos = ofe.compute_pathlength_from_polyline(ref)
    tan = ofe.compute_vertex_tangents(ref)
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(200):
        x, y = 4000.0 + float(rng.uniform(0.5, 10.5)), -3000.0 + float(rng.uniform(-3, 3))
        s_o, d_o = ofe.project(ref, pos, x, y)
        s_n, d_n = _capi.project(ref, pos, x, y)
        assert abs(s_o - s_n) < 1e-11 and abs(d_o - d_n) < 1e-11
        k = min(int(np.searchsorted(pos, s_n, side="right")) - 1, len(ref) - 2)
        lam = (s_n - pos[k]) / (pos[k + 1] - pos[k])
        e, q = ref[k + 1] - ref[k], np.array([x, y]) - ref[k]
        t = (1.0 - lam) * tan[k] + lam * tan[k + 1]
        worst = max(worst, abs(float((q - lam * e) @ t)))
        assert abs(s_n - (x - 4000.0)) < 1e-9 and abs(d_n - (y + 3000.0)) < 1e-9   # (the stretch is straight to 1e-10 m)
    assert worst < 1e-12
