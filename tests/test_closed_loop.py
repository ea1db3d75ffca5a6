"""Closed-loop driver (commonroad_rp_amd.harness.run_closed_loop, the loop of run_planner.py:61-107):
the stand-alone planner and -- where the reference is mounted -- the reference planner itself must trace
the same state sequence over several replanning cycles (winner hand-over ``optimal[k][1 + temp]``,
re-sampling of the velocity grid from the current speed, table re-use across ``reset``)."""
import os
import sys

import numpy as np
import pytest

from _golden import build_planner_from_plan_golden
from commonroad_rp_amd.harness import run_closed_loop

HERE = os.path.dirname(os.path.abspath(__file__))
HAVE_REF = os.path.isdir("/root/reference/commonroad_rp")


def _trace(states):
    return np.array([[s.time_step, s.position[0], s.position[1], s.orientation, s.velocity, s.acceleration or 0.0]
                     for s in states], dtype=float)


@pytest.mark.parametrize("name", ["plan_arc_hv_obs", "plan_scurve_lv"])
def test_closed_loop_cpu_glue_matches_reference(name):
    from _oracle_ctx import OracleContext
    rp, z = build_planner_from_plan_golden(name, OracleContext)
    res = run_closed_loop(rp, max_steps=12, desired_velocity=float(z["desired_speed"]))
    assert res.completed and res.n_replans == 4 and len(res.states) == 13
    tr = _trace(res.states)
    assert np.all(np.diff(tr[:, 0]) == 1)            # consecutive time steps
    if not HAVE_REF:
        return
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    case = {c["name"]: c for c in mg.plan_cases()}[name]
    ref = mg.make_planner(dict(case))
    ref.record_state_and_input = lambda state: ref._record_state_list.append(state)   # InputState is a stand-in here
    ref_res = run_closed_loop(ref, max_steps=12, desired_velocity=float(z["desired_speed"]))
    assert ref_res.completed
    np.testing.assert_allclose(tr, _trace(ref_res.states), rtol=0, atol=1e-8)


@pytest.mark.gpu
def test_closed_loop_gpu():
    from commonroad_rp_amd._capi import RpContext
    from _oracle_ctx import OracleContext
    a, z = build_planner_from_plan_golden("plan_arc_hv_obs", RpContext)
    b, _ = build_planner_from_plan_golden("plan_arc_hv_obs", OracleContext)
    ra = run_closed_loop(a, max_steps=12, desired_velocity=float(z["desired_speed"]))
    rb = run_closed_loop(b, max_steps=12, desired_velocity=float(z["desired_speed"]))
    assert ra.completed and rb.completed and ra.n_replans == rb.n_replans == 4
    np.testing.assert_allclose(_trace(ra.states), _trace(rb.states), rtol=0, atol=1e-6)
    # the fifth replanning cycle of this scenario (hard braking behind the obstacle) has no feasible
    # candidate at any sampling level: both back ends must give up at the same cycle with the same trace
    a.close()
    a, z = build_planner_from_plan_golden("plan_arc_hv_obs", RpContext)
    b, _ = build_planner_from_plan_golden("plan_arc_hv_obs", OracleContext)
    ra = run_closed_loop(a, max_steps=15, desired_velocity=float(z["desired_speed"]))
    rb = run_closed_loop(b, max_steps=15, desired_velocity=float(z["desired_speed"]))
    assert not ra.completed and not rb.completed and ra.n_replans == rb.n_replans == 5
    np.testing.assert_allclose(_trace(ra.states), _trace(rb.states), rtol=0, atol=1e-6)
    a.close()


def test_closed_loop_overlap_defers_bookkeeping_into_device_time():
    """``overlap=True``: the on_step call of cycle k runs between the launch and the wait of cycle k + 1; the loop's
    states are the same as without it and every cycle's callback still runs, in order."""
    from _oracle_ctx import OracleContext
    events = []

    class Ctx(OracleContext):
        def plan_begin(self, inp, *a, **k):
            events.append("begin")
            return super().plan_begin(inp, *a, **k)

        def plan_wait(self):
            events.append("wait")
            return super().plan_wait()

    def on_step(k, planner, optimal):
        events.append(f"step{k}")
        assert optimal[0].state_list[1].time_step == planner.record_state_list[k + 1].time_step   # (objects built on access)

    a, z = build_planner_from_plan_golden("plan_arc_hv_obs", Ctx)
    ra = run_closed_loop(a, max_steps=6, replanning_frequency=1, desired_velocity=float(z["desired_speed"]), on_step=on_step, overlap=True)
    b, _ = build_planner_from_plan_golden("plan_arc_hv_obs", OracleContext)
    rb = run_closed_loop(b, max_steps=6, replanning_frequency=1, desired_velocity=float(z["desired_speed"]))
    np.testing.assert_array_equal(_trace(ra.states), _trace(rb.states))
    steps = [e for e in events if e.startswith("step")]
    assert steps == [f"step{k}" for k in range(len(steps))] and len(steps) == len(ra.states) - 1
    # the callback of cycle k sits between the launch and the wait of cycle k + 1
    i = events.index("step0")
    assert events[i - 1] == "begin" and events[i + 1] == "wait"


@pytest.mark.gpu
def test_plan_begin_wait_gpu():
    """rp_plan_begin / rp_plan_wait: the same result as rp_plan; two plans in flight on two contexts; call-sequence errors."""
    from _golden import Golden
    from commonroad_rp_amd._capi import RpContext, RpError
    g1, g2 = Golden("arc_hv_l2_obs"), Golden("scurve_hv_l3")
    c1, c2 = RpContext(0), RpContext(0)
    g1.setup_context(c1)
    g2.setup_context(c2)
    want1, want2 = c1.plan(g1.inputs), c2.plan(g2.inputs)
    for _ in range(3):
        c1.plan_begin(g1.inputs)
        c2.plan_begin(g2.inputs)          # both on the device before either is collected
        with pytest.raises(RpError):
            c1.plan_begin(g1.inputs)      # one plan in flight per context
        with pytest.raises(RpError):
            c1.fetch_status()             # "the last plan" is the last COLLECTED one: nothing may read past a plan in flight
        o2, o1 = c2.plan_wait(), c1.plan_wait()
        for o, w in ((o1, want1), (o2, want2)):
            assert (o.best_index, o.best_cost, o.n_feasible, o.n_collision_before_best) == (w.best_index, w.best_cost, w.n_feasible, w.n_collision_before_best)
            np.testing.assert_array_equal(o.best_states, w.best_states)
            np.testing.assert_array_equal(o.best_lat_coeffs, w.best_lat_coeffs)
    st, _ = c1.fetch_status()
    assert len(st) == g1.inputs.n_candidates
    c1.close()
    c2.close()


@pytest.mark.gpu
def test_levels_ahead_gpu():
    """The stand-alone planner's cycle on the device when level 1 has no winner: first cycle level by level, the next one with two
    levels in flight (second context); same results as the oracle-backed planner both times."""
    from _oracle_ctx import OracleContext
    from commonroad_rp_amd._capi import RpContext
    a, _ = build_planner_from_plan_golden("plan_all_collide", RpContext)
    b, _ = build_planner_from_plan_golden("plan_all_collide", OracleContext)
    for cycle in range(3):
        ra, rb = a.plan(), b.plan()
        assert ra is None and rb is None
        assert (a.infeasible_count_kinematics, a.infeasible_count_collision) == (b.infeasible_count_kinematics, b.infeasible_count_collision)
        assert dict(a.infeasible_reason_dict) == dict(b.infeasible_reason_dict)
    assert a._levels_ahead and getattr(a, "_rp_ctx2", None) is not None
    a.close()
