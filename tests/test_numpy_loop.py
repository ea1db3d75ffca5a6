"""The loop-faithful NumPy restatement (oracle/numpy_loop.py: the reference's execution model, timed by bench.py as the
second CPU baseline) against the golden vectors produced by the reference's own Python, and against the C oracle.

Tolerances as in test_oracle_golden.py: labels / reasons / winner / lazy collision counter exact, costs 1e-10 relative,
states 1e-9 absolute, coefficients 1e-9 relative.
"""
import numpy as np
import pytest

from _golden import Golden, case_names
from oracle import numpy_loop, oracle

NAMES = case_names()


@pytest.mark.parametrize("name", NAMES)
def test_numpy_loop_matches_reference(name):
    g = Golden(name)
    C = g.inputs.n_candidates
    r = numpy_loop.plan(g.inputs, g.oracle_tables())
    label = r["status"] & 3
    reason = (r["status"] >> 4) & 7
    np.testing.assert_array_equal(reason, g["reason"])
    # collision labels are lazy here exactly as in the reference: only what the sorted walk visited is marked
    np.testing.assert_array_equal(label, g["label"])
    has_cost = ~np.isnan(g["cost"])
    np.testing.assert_allclose(r["cost"][has_cost], g["cost"][has_cost], rtol=1e-10)
    assert np.all(np.isnan(r["cost"][~has_cost]))
    assert r["best_index"] == int(g["winner"])
    assert int(np.sum(label == 3)) == int(g["n_infeasible_collision"])
    if r["best_index"] >= 0:
        np.testing.assert_allclose(r["best_cost"], float(g["winner_cost"]), rtol=1e-10)
    idx = g["state_index"]
    for j, k in enumerate(idx):
        k = int(k)
        if k in r["states"]:
            np.testing.assert_allclose(r["states"][k], g["states"][j], atol=1e-9, rtol=0)
    for k in range(0, C, max(1, C // 16)):   # polynomials of a few candidates (np.linalg.solve, as the reference)
        lon, lat, tl = numpy_loop.sample(g.inputs.params, g.inputs, k)
        np.testing.assert_allclose(lon, g["lon_coeffs"][k], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lat, g["lat_coeffs"][k], rtol=1e-9, atol=1e-12)


def test_numpy_loop_matches_c_oracle_on_a_bench_slice():
    """what bench.py times: the first candidates of cfg2 in draw mode; eager collision labels as the C oracle has them"""
    from commonroad_rp_amd import workloads
    w = workloads.cfg2(flags=1)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    ref = oracle.plan(w.inputs, tb, 0, 400, want_states=True)
    r = numpy_loop.plan(w.inputs, tb, 0, 400, eager_collision=True)
    np.testing.assert_array_equal(r["status"], ref.status)
    np.testing.assert_allclose(r["cost"], ref.cost, rtol=1e-12, equal_nan=True)
    assert r["best_index"] == ref.out.best_index
    for k, st in r["states"].items():
        np.testing.assert_allclose(st, ref.states[k], atol=1e-11, rtol=0)
