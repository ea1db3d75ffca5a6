"""The CPU oracle (oracle/rp_oracle.c) against the golden vectors produced by the reference's own
Python (tests/golden/make_golden.py).  This is what pins the oracle.

Tolerances: labels / reasons / winner / counters exact; polynomial coefficients 1e-9 relative
(LAPACK vs. hand-rolled pivoted LU); states 1e-9 absolute (contract: 1e-6, BASELINE.json);
costs 1e-10 relative.
"""
import numpy as np
import pytest

from _golden import Golden, case_names
from oracle import oracle

STATE_ATOL = 1e-9
NAMES = case_names()


def test_fixture_inventory():
    assert len(NAMES) >= 18


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    g = Golden(name)
    run = oracle.plan(g.inputs, g.oracle_tables(), want_states=True)
    C = g.inputs.n_candidates
    assert C == len(g["label"])
    label = run.status & 3
    reason = (run.status >> 4) & 7

    # polynomial coefficients (sampling.py:202-242, polynomial_trajectory.py:292-360)
    np.testing.assert_allclose(run.coeffs[:, :6], g["lon_coeffs"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(run.coeffs[:, 6:12], g["lat_coeffs"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(run.coeffs[:, 12], g["lat_T"], rtol=1e-12)

    # first-failure reasons and the counters built from them (reactive_planner.py:796-805, 971-1017)
    np.testing.assert_array_equal(reason, g["reason"])
    np.testing.assert_array_equal(run.out.reason_counts[1:6], g["reason_counts"])
    assert run.out.n_infeasible_kinematics == int(g["n_infeasible_kinematics"])

    # labels: kinematic labels exact; collision labels are eager here, lazy in the reference
    ref_label = g["label"]
    eager = g["collide_all"].astype(bool)
    exp = ref_label.copy()
    exp[(ref_label == 1) & eager] = 3
    np.testing.assert_array_equal(label, exp)

    # costs of every candidate the reference costed
    has_cost = ~np.isnan(g["cost"])
    np.testing.assert_allclose(run.cost[has_cost], g["cost"][has_cost], rtol=1e-10)
    assert np.all(np.isnan(run.cost[~has_cost]))

    # winner + lazy collision counter (trajectories.py:502-510, reactive_planner.py:1031-1063)
    assert run.out.best_index == int(g["winner"])
    assert run.out.n_collision_before_best == int(g["n_infeasible_collision"])
    if run.out.best_index >= 0:
        np.testing.assert_allclose(run.out.best_cost, float(g["winner_cost"]), rtol=1e-10)

    # full state blocks
    idx = g["state_index"]
    np.testing.assert_allclose(run.states[idx], g["states"], rtol=0, atol=STATE_ATOL)
    if run.out.best_index >= 0:
        np.testing.assert_array_equal(run.out.best_states, run.states[run.out.best_index])


def test_np_sum_is_numpys_pairwise_sum():
    import ctypes as C
    rng = np.random.default_rng(3)
    lib = oracle.lib()
    for n in (1, 5, 8, 9, 21, 31, 61, 101, 128, 129, 300, 1000):
        for _ in range(20):
            a = rng.normal(size=n) * 10.0 ** rng.integers(-3, 6, size=n)
            got = lib.rpo_np_sum(a.ctypes.data_as(C.POINTER(C.c_double)), n)
            assert got == float(np.sum(a)), n


def test_coefficient_solves_against_numpy():
    import ctypes as C
    rng = np.random.default_rng(5)
    lib = oracle.lib()
    c = np.zeros(6)
    for _ in range(200):
        p0, v0, a0, pf, T = rng.uniform(-50, 50), rng.uniform(0, 20), rng.uniform(-3, 3), rng.uniform(-50, 80), rng.uniform(0.2, 60)
        lib.rpo_quintic_coeffs(p0, v0, a0, pf, 0.0, 0.0, T, c.ctypes.data_as(C.POINTER(C.c_double)))
        A = np.array([[T ** 3, T ** 4, T ** 5], [3 * T ** 2, 4 * T ** 3, 5 * T ** 4], [6 * T, 12 * T ** 2, 20 * T ** 3]])
        b = np.array([pf - (p0 + v0 * T + .5 * a0 * T * T), 0 - (v0 + a0 * T), 0 - a0])
        x = np.linalg.solve(A, b)
        # compare through the polynomial's values, which is what the planner consumes
        tt = np.linspace(0, T, 7)
        ref = p0 + v0 * tt + .5 * a0 * tt ** 2 + x[0] * tt ** 3 + x[1] * tt ** 4 + x[2] * tt ** 5
        got = c[0] + c[1] * tt + c[2] * tt ** 2 + c[3] * tt ** 3 + c[4] * tt ** 4 + c[5] * tt ** 5
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-9)
