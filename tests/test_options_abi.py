"""Per-context options (rp_set_option), wait modes (rp_set_wait_mode), struct_size checks (RP_EABI) and the folded plan entries of
ABI version 2 -- through the C ABI on the device."""
import ctypes as C

import numpy as np
import pytest

from _golden import Golden
from commonroad_rp_amd import _capi
from commonroad_rp_amd._capi import PLAN_BEGIN, PLAN_PACKED, PLAN_ROWS, RpContext, RpError, RpGrids, RpResult

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx_case():
    _capi.set_default_options(None)
    g = Golden("arc_hv_l2_obs")
    ctx = RpContext(0)
    g.setup_context(ctx)
    yield ctx, g
    ctx.close()


def test_options_by_name(ctx_case):
    ctx, g = ctx_case
    assert ctx.get_option("lanes") == 0 and ctx.get_option("lazy") == -1 and ctx.get_option("fused_lon") == 1
    ctx.set_option("lanes", 32)
    assert ctx.get_option("lanes") == 32
    with pytest.raises(RpError, match="unknown option"):
        ctx.set_option("no_such_switch", 1)
    with pytest.raises(RpError, match="out of range"):
        ctx.set_option("lazy", 7)
    # the process-wide set (tests/_paths.py) is applied to live contexts and taken back
    _capi.set_default_options({"fused_lon": 0, "lanes": 64})
    assert ctx.get_option("fused_lon") == 0 and ctx.get_option("lanes") == 64
    _capi.set_default_options(None)
    assert ctx.get_option("fused_lon") == 1 and ctx.get_option("lanes") == 32   # (back to what the context had before the set)
    # a plan in flight: options stay as they are
    ctx.plan_begin(g.inputs)
    with pytest.raises(RpError, match="in flight"):
        ctx.set_option("lanes", 16)
    ctx.plan_wait()


@pytest.mark.parametrize("mode", [_capi.WAIT_SPIN, _capi.WAIT_YIELD, _capi.WAIT_EVENT])
def test_wait_modes_deliver_the_same_result(ctx_case, mode):
    ctx, g = ctx_case
    ref = ctx.plan(g.inputs)
    ctx.set_wait_mode(mode)
    assert ctx.wait_mode() == mode and ctx.wait_mode_name() == ("spin", "yield", "event")[mode]
    for path in ({}, {"fused_lon": 0}, {"fused_lon": 0, "auto_materialize": 0}):   # single launch | two kernels | + winner re-evaluation
        for k, v in path.items():
            ctx.set_option(k, v)
        out = ctx.plan(g.inputs)
        assert out.best_index == ref.best_index == int(g["winner"])
        np.testing.assert_allclose(out.best_cost, ref.best_cost, rtol=1e-12)
        assert out.n_feasible == ref.n_feasible and out.n_collision_before_best == ref.n_collision_before_best
        np.testing.assert_allclose(out.best_states, ref.best_states, rtol=0, atol=1e-9)   # (launch paths differ in the last bits of a row)
        ctx.plan_begin(g.inputs)
        two = ctx.plan_wait()
        assert two.best_index == ref.best_index
        np.testing.assert_array_equal(two.best_states, out.best_states)                    # (the same path in two halves: the same bits)
    with pytest.raises(RpError):
        ctx.set_wait_mode(9)


def test_struct_size_is_checked(ctx_case):
    ctx, g = ctx_case
    lib, h = ctx._lib, ctx._h
    inp = g.inputs
    best = np.empty((_capi.N_ARRAYS, inp.params.N + 1))
    res = RpResult()

    def call(p, k, gr, r):
        return lib.rp_plan(h, C.byref(p), C.byref(k), C.byref(gr), 0, -1, 0, C.byref(r), best.ctypes.data)
    assert call(inp.params, inp.cost, inp.grids(), res) == 0
    for which in ("params", "cost", "grids", "result"):
        p, k, gr, r = _capi.copy_params(inp.params), _capi.RpCost.from_buffer_copy(bytes(inp.cost)), RpGrids.from_buffer_copy(bytes(inp.grids())), RpResult()
        {"params": p, "cost": k, "grids": gr, "result": r}[which].struct_size -= 8   # (a caller built against a smaller struct)
        assert call(p, k, gr, r) == _capi.E_ABI, which
        assert b"struct_size" in lib.rp_last_error(h)
    assert call(inp.params, inp.cost, inp.grids(), res) == 0 and res.best_index == int(g["winner"])
    assert lib.rp_plan(h, C.byref(inp.params), C.byref(inp.cost), C.byref(inp.grids()), 0, -1, 1 << 9, C.byref(res), None) == -1   # unknown flag


def test_folded_entries_agree(ctx_case):
    """rp_plan with RP_PLAN_BEGIN / RP_PLAN_PACKED and rp_plan_levels with them: the same result as the plain call"""
    ctx, g = ctx_case
    inp = g.inputs
    ref = ctx.plan(inp)
    res, blk, buf = ctx.plan_packed(inp.params, inp.cost, inp.T, inp.traj_len, inp.L, inp.D)
    assert res.best_index == ref.best_index
    np.testing.assert_array_equal(blk, ref.best_states)
    np.testing.assert_array_equal(buf, _capi.pack_trajectory(ref.best_states, inp.params.dt, inp.params.wheelbase, inp.params.x0_orientation))
    levels = [(inp.T, inp.traj_len, inp.L, inp.D)] * 2
    res2, lvl, blk2, buf2 = ctx.plan_levels_packed(inp.params, inp.cost, levels)
    assert lvl == 0 and res2.best_index == ref.best_index
    np.testing.assert_array_equal(blk2, ref.best_states)
    ctx.plan_levels_begin(inp.params, inp.cost, levels)
    out = ctx.plan_wait()
    assert out.best_index == ref.best_index and ctx.last_level() == 0
    # packed calls plan whole grids
    gd = RpGrids()
    gd.nT, gd.nL, gd.nD = len(inp.T), len(inp.L), len(inp.D)
    raw = (C.c_double * ((14 + 13) * (inp.params.N + 1)))()
    assert ctx._lib.rp_plan(ctx._h, C.byref(inp.params), C.byref(inp.cost), C.byref(gd), 1, -1, PLAN_PACKED, C.byref(ctx._res), raw) == -1
    assert PLAN_BEGIN | PLAN_ROWS == 3


def test_sweep_takes_over_from_exhausted_lists():
    """cfg4 with its road boundary: the three lists of the cost-ordered stage run dry (the winner sorts behind ~30 000 colliding
    candidates); on a batch of this size the bounded sweep, not the eager kernel, finds it -- same winner, same counters."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    _capi.set_default_options(None)
    w = W.cfg4(road_boundary=True)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    o = oracle.plan(w.inputs, tb, want_states=False, nthreads=8)
    ctx = RpContext(0)
    try:
        w.setup(ctx)
        ctx.set_collision_path(_capi.COLLISION_COST_ORDERED)
        out = ctx.plan(w.inputs, want_best_states=False)
        assert ctx.last_path() == 3
        assert out.best_index == o.out.best_index and out.n_collision_before_best == o.out.n_collision_before_best
        assert out.n_feasible == o.out.n_feasible and out.n_collision == out.n_collision_before_best
        np.testing.assert_array_equal(out.reason_counts, o.out.reason_counts)
        status, cost = ctx.fetch_status()
        before = (cost < out.best_cost) | ((cost == out.best_cost) & (np.arange(len(cost)) < out.best_index))
        np.testing.assert_array_equal((status & 3)[before], (o.status & 3)[before])      # every label in front of the winner
        d = (status & 3) != (o.status & 3)
        assert np.all(((o.status & 3)[d] == 3) & ((status & 3)[d] == 1))                # behind it: colliding candidates it never looked at
        ctx.set_option("sweep", 0)                                                       # the eager kernel instead: labels everything
        out2 = ctx.plan(w.inputs, want_best_states=False)
        assert ctx.last_path() == 2 and out2.best_index == out.best_index and out2.n_collision == o.out.n_collision
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["plan_arc_hv_obs", "plan_scurve_lv", "plan_standstill", "plan_all_collide", "plan_arc_stop"])
def test_plan_cycle_is_the_cycle_of_plan(name):
    """rp_plan_cycle (one foreign call per replanning cycle: initial state, low-velocity flag, lateral samples + current offset, level
    loop, output packing) against the same cycle through the separate calls -- with the curvilinear state handed over, and worked
    out from the Cartesian one inside the call."""
    from _golden import build_planner_from_plan_golden, compare_plan_result, plan_case_names
    if name not in plan_case_names():
        pytest.skip("fixture absent")
    _capi.set_default_options(None)
    calls = {"cycle": 0, "packed": 0}

    class Counting(RpContext):
        def plan_cycle(self, *a, **k):
            calls["cycle"] += 1
            return super().plan_cycle(*a, **k)

        def plan_packed(self, *a, **k):
            calls["packed"] += 1
            return super().plan_packed(*a, **k)

    class NoCycle(RpContext):
        plan_cycle = None   # the path of rounds 1-4: separate calls

    rp_a, z = build_planner_from_plan_golden(name, Counting)
    rp_b, _ = build_planner_from_plan_golden(name, NoCycle)
    rp_a.always_plan_cycle = True   # (by default only cycles that start from a Cartesian state take it)
    try:
        res_a, res_b = rp_a.plan(), rp_b.plan()
        compare_plan_result(res_a, rp_a, z, 1e-6)
        compare_plan_result(res_b, rp_b, z, 1e-6)
        fast = rp_a._fast_path_ok()
        assert (calls["cycle"] >= 1 and calls["packed"] == 0) or not fast
        assert rp_a.infeasible_count_kinematics == rp_b.infeasible_count_kinematics and rp_a.infeasible_count_collision == rp_b.infeasible_count_collision
        # the same cycle from the Cartesian state alone: the curvilinear state comes out of the call, as _compute_initial_states has it
        x0 = rp_a.x_0
        want = rp_b._compute_initial_states(x0)
        for rp in (rp_a, rp_b):
            rp.reset(initial_state_cart=x0, initial_state_curv=None, collision_checker=rp.collision_checker, coordinate_system=rp.coordinate_system)
        r2a, r2b = rp_a.plan(), rp_b.plan()
        np.testing.assert_allclose(np.concatenate(rp_a.x_0_cl), np.concatenate(want), rtol=0, atol=1e-12)
        assert (r2a is None) == (r2b is None)
        if r2a is not None:
            for sa, sb in zip(r2a[0].state_list, r2b[0].state_list):
                np.testing.assert_allclose(sa.position, sb.position, rtol=0, atol=1e-9)
                assert sa.time_step == sb.time_step
            assert r2a[2][1] == r2b[2][1] and r2a[3][1] == r2b[3][1]
    finally:
        rp_a.close(); rp_b.close()
