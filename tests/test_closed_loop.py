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
        def plan_levels_begin(self, *a, **k):
            events.append("begin")
            return super().plan_levels_begin(*a, **k)

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
@pytest.mark.parametrize("policy", ["chain", "adaptive", "sequential"])
@pytest.mark.parametrize("name", ["plan_all_collide", "plan_arc_hv_obs", "plan_scurve_lv", "plan_arc_stop"])
def test_level_loop_in_one_round_trip_gpu(name, policy):
    """The stand-alone planner's cycle on the device: the level loop of plan() (reactive_planner.py:616-636) through rp_plan_levels
    -- every level's kernels on the stream at once, a level that finds a winner stops the ones behind it.  Same trajectory, counters
    and reasons as the oracle-backed planner, which visits the levels one by one; plan_all_collide visits all three."""
    from _oracle_ctx import OracleContext
    from commonroad_rp_amd._capi import RpContext
    a, _ = build_planner_from_plan_golden(name, RpContext)
    b, _ = build_planner_from_plan_golden(name, OracleContext)
    assert a._fast_path_ok()
    a.level_policy = policy
    for cycle in range(3):
        ra, rb = a.plan(), b.plan()
        assert (ra is None) == (rb is None)
        assert (a.infeasible_count_kinematics, a.infeasible_count_collision) == (b.infeasible_count_kinematics, b.infeasible_count_collision)
        assert dict(a.infeasible_reason_dict) == dict(b.infeasible_reason_dict)
        if ra is not None:
            for k in (0, 5, -1):
                np.testing.assert_allclose(ra[0].state_list[k].position, rb[0].state_list[k].position, rtol=0, atol=1e-6)
                np.testing.assert_allclose(ra[2][k], rb[2][k], rtol=0, atol=1e-6)
    a.close()


@pytest.mark.gpu
def test_plan_levels_entry_gpu():
    """rp_plan_levels against rp_plan level by level: (a) the first level wins -- the levels behind it never run, the context's "last
    plan" is that level; (b) a first level without a winner (everything collides) -- the result, counters and rows are the second
    level's; (c) no level has a winner -- the last level's counters; (d) an empty level in the chain is passed over; (e) a level too
    large for the chain is planned behind it, inside the same call; (f) the two-halves form."""
    from _golden import Golden
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext, PlanInputs
    from commonroad_rp_amd.collision import ObstacleTables
    ctx, ref = RpContext(0), RpContext(0)

    def grids(inp):
        return (inp.T, inp.traj_len, inp.L, inp.D)

    def same(res, out):
        assert (res.best_index, res.n_candidates, res.n_feasible, res.n_collision_before_best) == \
            (out.best_index, out.n_candidates, out.n_feasible, out.n_collision_before_best)
        np.testing.assert_array_equal(np.array(res.reason_counts[:]), out.reason_counts)
        if out.best_index >= 0:
            assert res.best_cost == out.best_cost
    ws = [W.cfg1(level=k) for k in (1, 2, 3)]
    for c in (ctx, ref):
        ws[0].setup(c)
    p, cost = ws[0].inputs.params, ws[0].inputs.cost
    outs = [ref.plan(w.inputs) for w in ws]
    assert outs[0].best_index >= 0
    # (a)
    res, lvl, blk, buf = ctx.plan_levels_packed(p, cost, [grids(w.inputs) for w in ws])
    assert lvl == 0 and ctx.last_level() == 0
    same(res, outs[0])
    np.testing.assert_array_equal(blk, outs[0].best_states)
    st, cs = ctx.fetch_status()
    assert len(st) == ws[0].n_candidates
    ref.plan(ws[0].inputs)
    np.testing.assert_array_equal(st, ref.fetch_status()[0])
    # (b) an obstacle over the whole road: level 1 ... and with a far smaller one only the coarse level fails
    wall = ObstacleTables(static_circ=[[ws[0].coordinate_system.reference[60][0], ws[0].coordinate_system.reference[60][1], 40.0]])
    for c in (ctx, ref):
        c.set_obstacles(wall)
    outs_w = [ref.plan(w.inputs) for w in ws]
    assert all(o.best_index < 0 for o in outs_w)
    # (c)
    res, lvl, blk, buf = ctx.plan_levels_packed(p, cost, [grids(w.inputs) for w in ws])
    assert lvl == 2 and blk is None
    same(res, outs_w[2])
    for c in (ctx, ref):
        ws[0].setup(c)
    # (b) a first level without a winner (target speeds three times the limit: every candidate fails its kinematics), the second delivers
    i0 = ws[0].inputs
    sub = PlanInputs(p, cost, i0.T, i0.traj_len, i0.L * 3.0 + 30.0, i0.D)
    o_sub, o2 = ref.plan(sub), ref.plan(ws[1].inputs)
    assert o_sub.best_index < 0 and o2.best_index >= 0
    res, lvl, blk, buf = ctx.plan_levels_packed(p, cost, [grids(sub), grids(ws[1].inputs), grids(ws[2].inputs)])
    assert lvl == 1 and ctx.last_level() == 1
    same(res, o2)
    np.testing.assert_array_equal(blk, o2.best_states)
    st, cs = ctx.fetch_status()                 # "the last plan" is the level that delivered
    assert len(st) == ws[1].n_candidates
    np.testing.assert_array_equal(st, ref.fetch_status()[0])
    one, s1, c1 = ctx.eval_one(o2.best_index)   # ... also for launches that read its grids again
    np.testing.assert_array_equal(one, o2.best_states)
    # (d) empty levels in the chain are passed over
    empty = (i0.T[:0], i0.traj_len[:0], i0.L, i0.D)
    res, lvl, blk, buf = ctx.plan_levels_packed(p, cost, [empty, grids(sub), empty, grids(ws[1].inputs)])
    assert lvl == 3
    same(res, o2)
    res, lvl, blk, buf = ctx.plan_levels_packed(p, cost, [empty, empty])
    assert res.best_index < 0 and res.n_candidates == 0 and blk is None
    # (e) a level beyond the chain's size behind a failing one: cfg2's grid (7 440 candidates, 61 samples fit; cfg3's does not)
    w3 = W.cfg3()
    for c in (ctx, ref):
        w3.setup(c)
    p3, c3 = w3.inputs.params, w3.inputs.cost
    big = ref.plan(w3.inputs)
    i3 = w3.inputs
    none = PlanInputs(p3, c3, i3.T[:1], i3.traj_len[:1], i3.L[-1:] * 3.0, i3.D[:4])   # (target speed three times the limit: infeasible)
    o_none = ref.plan(none)
    assert o_none.best_index < 0
    res, lvl, blk, buf = ctx.plan_levels_packed(p3, c3, [grids(none), grids(i3)])
    assert lvl == 1
    same(res, big)
    # (f) begin / wait
    ctx.plan_levels_begin(p3, c3, [grids(none), grids(none), grids(i3)])
    out = ctx.plan_wait()
    assert ctx.last_level() == 2 and out.best_index == big.best_index and out.n_feasible == big.n_feasible
    ctx.close()
    ref.close()
