"""Parity of the HIP path (librp_amd.so through the C ABI) against the CPU oracle and against the
golden vectors of the reference, on the same inputs.  Needs a real MI355X: ``pytest -m gpu``.

Tolerances (BASELINE.json north_star: 1e-6 on trajectory states, same optimal index):
  states      |gpu - oracle| <= 1e-6 absolute (observed ~1e-12; the bound is the contract)
  costs       1e-9 relative
  labels, first-failure reasons, counters, winner index: exact
"""
import numpy as np
import pytest

from _golden import Golden, case_names
from commonroad_rp_amd import _capi
from commonroad_rp_amd._capi import (RpContext, FLAG_MATERIALIZE_ALL, FLAG_DRAW_ALL, FLAG_SKIP_COLLISION, copy_params,
                                     PlanInputs)

pytestmark = pytest.mark.gpu

STATE_ATOL = 1e-6
COST_RTOL = 1e-9
NAMES = case_names()


from _lazy import lazy_relaxed as _lazy_relaxed
from _paths import LAUNCH_PATHS, launch_path_env


@pytest.fixture(scope="module", params=list(LAUNCH_PATHS))
def ctx(request):
    with launch_path_env(request.param):
        c = RpContext(0)
        yield c
        c.close()


def _with_flags(inp: PlanInputs, extra: int) -> PlanInputs:
    p = copy_params(inp.params)
    p.flags = p.flags | extra
    return PlanInputs(p, inp.cost, inp.T, inp.traj_len, inp.L, inp.D)


def _compare_status(status, cost, orun, ctx=None, out=None, lo=0):
    status, _ = _lazy_relaxed(status, cost, orun, ctx, out, lo)
    np.testing.assert_array_equal(status & 3, orun.status & 3)                  # labels
    np.testing.assert_array_equal((status >> 4) & 7, (orun.status >> 4) & 7)    # first-failure reasons
    kin = (orun.status & 3) == 2
    np.testing.assert_array_equal((status >> 8)[kin], (orun.status >> 8)[kin])  # first failing step
    has = ~np.isnan(orun.cost)
    assert np.all(np.isnan(cost[~has]))
    np.testing.assert_allclose(cost[has], orun.cost[has], rtol=COST_RTOL)


def _compare_collision_counts(out, oout, ctx=None, unlabelled=None):
    """infeasible_count_collision is the same on every path; the number of colliding candidates FOUND is all of them after an
    eager plan and at least the ones before the winner after a cost-ordered one."""
    assert out.n_collision_before_best == oout.n_collision_before_best
    if ctx is not None and ctx.last_path() in (1, 3):
        assert oout.n_collision_before_best <= out.n_collision <= oout.n_collision
        if ctx.last_path() == 3:   # bounded sweep: the count in front of the winner (what it looked at behind it is not a property of the plan)
            assert out.n_collision == oout.n_collision_before_best
        elif unlabelled is not None:
            assert out.n_collision == oout.n_collision - unlabelled
    else:
        assert out.n_collision == oout.n_collision


def _compare_out(out, oout, ctx=None):
    assert out.best_index == oout.best_index
    assert out.n_candidates == oout.n_candidates
    assert out.n_feasible == oout.n_feasible
    _compare_collision_counts(out, oout, ctx)
    np.testing.assert_array_equal(out.reason_counts[1:7], oout.reason_counts[1:7])
    if oout.best_index >= 0:
        np.testing.assert_allclose(out.best_cost, oout.best_cost, rtol=COST_RTOL)
        np.testing.assert_allclose(out.best_states, oout.best_states, rtol=0, atol=STATE_ATOL)
        np.testing.assert_allclose(out.best_lon_coeffs, oout.best_lon_coeffs, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out.best_lat_coeffs, oout.best_lat_coeffs, rtol=1e-9, atol=1e-12)
    else:
        assert np.isnan(out.best_cost) and out.best_states is None


@pytest.mark.parametrize("name", NAMES)
def test_fused_mode_matches_oracle_and_reference(ctx, name):
    from oracle import oracle
    g = Golden(name)
    g.setup_context(ctx)
    orun = oracle.plan(g.inputs, g.oracle_tables())
    out = ctx.plan(g.inputs)
    status, cost = ctx.fetch_status()
    _compare_status(status, cost, orun, ctx, out)
    _compare_out(out, orun.out, ctx)
    # and directly against what the reference's own Python produced
    assert out.best_index == int(g["winner"])
    assert out.n_collision_before_best == int(g["n_infeasible_collision"])
    assert out.n_infeasible_kinematics == int(g["n_infeasible_kinematics"])
    np.testing.assert_array_equal(out.reason_counts[1:6], g["reason_counts"])
    np.testing.assert_array_equal((status >> 4) & 7, g["reason"])
    has = ~np.isnan(g["cost"])
    np.testing.assert_allclose(cost[has], g["cost"][has], rtol=COST_RTOL)
    if out.best_index >= 0:
        k = list(g["state_index"]).index(out.best_index)
        np.testing.assert_allclose(out.best_states, g["states"][k], rtol=0, atol=STATE_ATOL)


@pytest.mark.parametrize("name", NAMES)
def test_materialize_mode_states(ctx, name):
    from oracle import oracle
    g = Golden(name)
    g.setup_context(ctx)
    inp = _with_flags(g.inputs, FLAG_MATERIALIZE_ALL)
    orun = oracle.plan(inp, g.oracle_tables())
    out = ctx.plan(inp)
    status, cost = ctx.fetch_status()
    _compare_status(status, cost, orun, ctx, out)
    _compare_out(out, orun.out, ctx)
    states = ctx.fetch_states()
    lab = orun.status & 3
    # state blocks are defined for candidates that have states in the reference: feasible /
    # colliding ones, and every candidate that passed the pre-filter in draw mode
    defined = (lab == 1) | (lab == 3)
    if g.draw:
        defined = np.ones_like(defined)
    np.testing.assert_allclose(states[defined], orun.states[defined], rtol=0, atol=STATE_ATOL)
    # golden states straight from the reference
    idx = g["state_index"]
    np.testing.assert_allclose(states[idx], g["states"], rtol=0, atol=STATE_ATOL)
    # re-evaluation of single candidates agrees with the batch
    for i in list(idx[:3]):
        st, s1, c1 = ctx.eval_one(int(i))
        np.testing.assert_array_equal(st, states[i])
        assert s1 == status[i]
        assert (np.isnan(c1) and np.isnan(cost[i])) or c1 == cost[i]


@pytest.mark.parametrize("name", ["arc_hv_l2_obs", "scurve_hv_l3", "arc_n70_factor2", "straight_hv_l2_ties"])
def test_sharded_ranges_compose(ctx, name):
    """Multi-GPU contract on one device: evaluating disjoint candidate ranges and combining the
    (cost, index) minima + counters reproduces the unsharded result."""
    g = Golden(name)
    g.setup_context(ctx)
    full = ctx.plan(g.inputs)
    C = g.inputs.n_candidates
    cuts = [0, C // 3 + 1, C // 2 + 3, C]
    parts, stat = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        parts.append(ctx.plan(g.inputs, lo, hi))
        stat.append(ctx.fetch_status())
    best = min(((p.best_cost, p.best_index) for p in parts if p.best_index >= 0), default=(np.nan, -1))
    assert best[1] == full.best_index
    assert sum(p.n_feasible for p in parts) == full.n_feasible
    if ctx.last_path() not in (1, 3):   # (the cost-ordered stage reports the colliding candidates it looked at, which depends on the range)
        assert sum(p.n_collision for p in parts) == full.n_collision
    np.testing.assert_array_equal(sum(p.reason_counts for p in parts), full.reason_counts)
    # second pass: colliding samples that precede the global winner, per shard
    n_before = 0
    for (lo, hi), p in zip(zip(cuts[:-1], cuts[1:]), parts):
        ctx.plan(g.inputs, lo, hi, want_best_states=False)
        n_before += ctx.count_collisions_before(best[0], best[1])
    assert n_before == full.n_collision_before_best


@pytest.mark.parametrize("name", ["arc_hv_l1", "scurve_lv_l1", "arc_stop_l1", "arc_n70_factor2"])
def test_plan_coeffs_entry(ctx, name):
    """Foreign SamplingSpace path: polynomials handed in explicitly (taken from the reference's
    own TrajectorySample objects in the fixture)."""
    from oracle import oracle
    g = Golden(name)
    g.setup_context(ctx)
    C = g.inputs.n_candidates
    nLD = len(g.inputs.L) * len(g.inputs.D)
    tl = np.repeat(g.inputs.traj_len, nLD)
    lonT = np.repeat(g.inputs.T, nLD)
    orun = oracle.plan_coeffs(g.inputs.params, g.inputs.cost, g.oracle_tables(), g["lon_coeffs"], g["lat_coeffs"], tl)
    out = ctx.plan_coeffs(g.inputs.params, g.inputs.cost, g["lon_coeffs"], g["lat_coeffs"], lonT, tl)
    status, cost = ctx.fetch_status()
    assert len(status) == C
    _compare_status(status, cost, orun, ctx, out)
    _compare_out(out, orun.out, ctx) if False else None
    assert out.best_index == orun.out.best_index == int(g["winner"])
    assert out.n_collision_before_best == orun.out.n_collision_before_best
    if out.best_index >= 0:
        np.testing.assert_allclose(out.best_states, orun.out.best_states, rtol=0, atol=STATE_ATOL)
        np.testing.assert_array_equal(out.best_lon_coeffs, g["lon_coeffs"][out.best_index])


def test_select_with_external_costs(ctx):
    """Plug-in CostFunction path: states materialised, costs computed elsewhere, argmin on device."""
    g = Golden("arc_hv_l2_obs")
    g.setup_context(ctx)
    inp = _with_flags(g.inputs, FLAG_MATERIALIZE_ALL)
    ctx.plan(inp)
    status, cost = ctx.fetch_status()
    lab = status & 3
    rng = np.random.default_rng(0)
    user = np.where((lab == 1) | (lab == 3), rng.uniform(1.0, 2.0, size=len(lab)), np.nan)
    user[np.flatnonzero(lab == 1)[5]] = 0.5     # plant a unique winner
    user[np.flatnonzero(lab == 3)[:4]] = 0.25   # four colliding samples sort before it
    out = ctx.select(user)
    assert out.best_index == int(np.flatnonzero(lab == 1)[5])
    assert out.best_cost == 0.5
    assert out.n_collision_before_best == 4
    st, _, _ = ctx.eval_one(out.best_index)
    np.testing.assert_array_equal(out.best_states, st)


@pytest.mark.parametrize("name", ["arc_hv_l2_obs", "arc_n70_factor2", "scurve_hv_l3"])
def test_select_after_plan_without_states(ctx, name):
    """rp_plan with best_states = NULL (nothing materialised; small batches take the single-launch path, which keeps
    its profile rows in LDS) followed by rp_select WITH states: the winner is re-evaluated on the launch path of the
    plan the costs belong to (this sequence dereferenced a null profile pointer)."""
    g = Golden(name)
    g.setup_context(ctx)
    ctx.plan(g.inputs, want_best_states=False)
    status, cost = ctx.fetch_status()
    lab = status & 3
    feas = np.flatnonzero(lab == 1)
    if len(feas) == 0:
        pytest.skip("no feasible candidate")
    user = np.where((lab == 1) | (lab == 3), 10.0 + np.arange(len(lab)) * 1e-3, np.nan)
    pick = int(feas[len(feas) // 2])
    user[pick] = 1.0
    if ctx.last_path() in (1, 3):
        # the plan answered the collision query in cost order: candidates behind its winner were never looked at, a selection with
        # other costs could crown a colliding one -- refused; plans for plug-in costs materialise and run eager
        with pytest.raises(_capi.RpError, match="cost order"):
            ctx.select(user)
        ctx.plan(_with_flags(g.inputs, FLAG_MATERIALIZE_ALL))
        assert ctx.last_path() == 0
        out = ctx.select(user)
        assert out.best_index == pick and out.best_cost == 1.0
        return
    out = ctx.select(user)
    assert out.best_index == pick and out.best_cost == 1.0
    assert out.n_collision_before_best == 0
    st, s1, _ = ctx.eval_one(pick)
    np.testing.assert_array_equal(out.best_states, st)
    assert s1 & 3 == 1
    # the winner's rows against the materialised run of the same plan
    ctx.plan(_with_flags(g.inputs, FLAG_MATERIALIZE_ALL))
    np.testing.assert_allclose(out.best_states, ctx.fetch_states(pick, 1)[0], rtol=0, atol=1e-12)
    # and a shard: plan without states, select without states, then with
    C = g.inputs.n_candidates
    lo, hi = C // 4, C // 4 + max(1, C // 2)
    ctx.plan(g.inputs, lo, hi, want_best_states=False)
    s2, c2 = ctx.fetch_status()
    u2 = np.where(((s2 & 3) == 1) | ((s2 & 3) == 3), 5.0, np.nan)
    o2 = ctx.select(u2, want_best_states=False)
    f2 = np.flatnonzero((s2 & 3) == 1)
    assert o2.best_index == (lo + int(f2[0]) if len(f2) else -1)     # ties: smallest index
    o3 = ctx.select(u2)
    assert o3.best_index == o2.best_index
    if o3.best_index >= 0:
        np.testing.assert_array_equal(o3.best_states, ctx.eval_one(o3.best_index)[0])


def test_edge_cases(ctx):
    g = Golden("arc_hv_l1")
    g.setup_context(ctx)
    # empty range
    out = ctx.plan(g.inputs, 5, 5)
    assert out.best_index == -1 and out.n_candidates == 0 and out.n_feasible == 0
    # single candidate ranges reproduce the batch statuses
    full = ctx.plan(g.inputs)
    status, cost = ctx.fetch_status()
    for i in (0, 7, full.best_index, g.inputs.n_candidates - 1):
        o = ctx.plan(g.inputs, i, i + 1)
        s1, c1 = ctx.fetch_status()
        assert s1[0] == status[i]
        assert (np.isnan(c1[0]) and np.isnan(cost[i])) or c1[0] == cost[i]
        assert o.best_index == (i if (status[i] & 3) == 1 else -1)
    # skipping the collision tables turns colliding samples back into feasible ones
    g2 = Golden("arc_all_collide")
    g2.setup_context(ctx)
    assert ctx.plan(g2.inputs).best_index == -1
    # ... and "no winner" hands the completion ticket over like a winner does: the result block in pinned memory carries the ticket of
    # THIS plan when the call returns (a missing ticket would leave the host in its 200-ms fall-back -- a stream synchronisation --
    # and the block without it)
    fallbacks = ctx.get_option("wait_fallbacks")
    for _ in range(3):
        assert ctx.plan(g2.inputs).best_index == -1
    assert ctx.get_option("wait_fallbacks") == fallbacks
    out = ctx.plan(_with_flags(g2.inputs, FLAG_SKIP_COLLISION))
    assert out.best_index >= 0 and out.n_collision == 0
    # error paths: bad range, plan before reference
    with pytest.raises(_capi.RpError):
        ctx.plan(g2.inputs, 10, 5)
    c2 = RpContext(0)
    with pytest.raises(_capi.RpError, match="rp_set_reference"):
        c2.plan(g2.inputs)
    c2.close()


def _synthetic_case(ref_path, N=30, seed=0, low=False):
    """Inputs on an arbitrary polyline: oracle tables + PlanInputs (no fixture behind it)."""
    from commonroad_rp_amd.coordinate_system import CoordinateSystem
    from commonroad_rp_amd.workloads import VEHICLE2, traj_len_of
    from commonroad_rp_amd._capi import make_params, make_cost
    co = CoordinateSystem(ref_path)
    dt = 0.1
    T = np.array([dt * k for k in range(8, N + 1, 3)])
    L = np.linspace(4.0, 14.0, 7)
    D = np.append(np.linspace(-2.5, 2.5, 6), 0.3)
    s0 = co.ref_pos[len(co.ref_pos) // 4]
    params = make_params(dt=dt, N=N, x0_lon=[s0, 1.5 if low else 9.0, 0.2], x0_lat=[0.3, 0.05, 0.0],
                         x0_orientation=float(co.ref_theta[len(co.ref_pos) // 4]), low_vel_mode=low, **VEHICLE2)
    return co, PlanInputs(params, make_cost(desired_speed=8.0), T, traj_len_of(T, dt), L, D)


@pytest.mark.parametrize("kind", ["long_table_global_memory", "irregular_spacing_binary_search"])
def test_table_fallback_paths(ctx, kind):
    """Reference tables too large for LDS (read through L1/L2 instead) and spacing too irregular for
    the bucket lookup (binary search instead): same results as the oracle."""
    from oracle import oracle
    from commonroad_rp_amd.collision import ObstacleTables
    if kind == "long_table_global_memory":
        s = np.arange(0.0, 1400.0, 0.5)                     # 2800 vertices -> 9 rows x 2800 x 8 B = 197 KB > 64 KB
        path = np.stack((s, 30.0 * np.sin(s / 90.0)), axis=1)
    else:
        rng = np.random.default_rng(4)
        seg = np.where(rng.random(400) < 0.1, 1e-4, rng.uniform(0.5, 2.0, 400))   # 1e-4 m segments -> > 8192 buckets
        s = np.concatenate(([0.0], np.cumsum(seg)))
        path = np.stack((s, 10.0 * np.sin(s / 60.0)), axis=1)
    co, inp = _synthetic_case(path)
    ctx.set_coordinate_system(co)
    ctx.set_obstacles(ObstacleTables())
    tb = oracle.OracleTables.from_coordinate_system(co)
    for flags in (0, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL):
        i2 = _with_flags(inp, flags)
        orun = oracle.plan(i2, tb)
        out = ctx.plan(i2)
        status, cost = ctx.fetch_status()
        _compare_status(status, cost, orun, ctx, out)
        _compare_out(out, orun.out, ctx)
        if flags:
            np.testing.assert_allclose(ctx.fetch_states(), orun.states, rtol=0, atol=STATE_ATOL)


def test_many_static_shapes_and_triangles(ctx):
    """Road-boundary style input: hundreds of triangles, circles and boxes."""
    from oracle import oracle
    from commonroad_rp_amd.collision import ObstacleTables
    g = Golden("arc_hv_l2_obs")
    rng = np.random.default_rng(7)
    co_xy = g["ref_path"]
    tris, circs, boxes = [], [], []
    for k in range(300):
        p = co_xy[rng.integers(0, len(co_xy))] + rng.normal(size=2) * 1.0 + np.array([0.0, rng.choice([-1, 1]) * 4.5])
        tris.append([p[0], p[1], p[0] + rng.uniform(0.5, 2), p[1] + rng.uniform(-1, 1), p[0] + rng.uniform(-1, 1), p[1] + rng.uniform(0.5, 2)])
    for k in range(40):
        p = co_xy[rng.integers(0, len(co_xy))] + np.array([0.0, rng.choice([-1, 1]) * rng.uniform(2.0, 5.0)])
        circs.append([p[0], p[1], rng.uniform(0.2, 0.8)])
        boxes.append([p[0] + 1.0, p[1] + rng.choice([-1, 1]) * 2.0, rng.uniform(-3, 3), rng.uniform(0.5, 2.5), rng.uniform(0.3, 1.0)])
    tables = ObstacleTables(static_tri=tris, static_circ=circs, static_obb=boxes)
    z = g.z
    ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]))
    ctx.set_obstacles(tables)
    tb = oracle.OracleTables(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"],
                             float(z["proj_d_limit"]), tables)
    orun = oracle.plan(g.inputs, tb)
    out = ctx.plan(g.inputs)
    status, cost = ctx.fetch_status()
    _compare_status(status, cost, orun, ctx, out)
    _compare_out(out, orun.out, ctx)
    assert orun.out.n_collision > 0 and out.n_feasible > out.n_collision


@pytest.mark.parametrize("N", [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129])
def test_horizon_lengths_around_lane_group_boundaries(ctx, N):
    """N + 1 below, at and above 16 / 32 / 64 / 128 steps, in low- and high-velocity mode."""
    from oracle import oracle
    from commonroad_rp_amd.collision import ObstacleTables
    from commonroad_rp_amd.workloads import traj_len_of
    s = np.arange(0.0, 400.0, 1.0)
    path = np.stack((s, 25.0 * np.sin(s / 70.0)), axis=1)
    for low in (False, True):
        co, inp = _synthetic_case(path, N=N, low=low)
        if N < 8:   # the T grid of _synthetic_case starts at 8 dt
            T = np.array([0.1 * k for k in range(1, N + 1)])
            inp = PlanInputs(inp.params, inp.cost, T, traj_len_of(T, 0.1), inp.L, inp.D)
        ctx.set_coordinate_system(co)
        ctx.set_obstacles(ObstacleTables())
        tb = oracle.OracleTables.from_coordinate_system(co)
        i2 = _with_flags(inp, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)
        orun = oracle.plan(i2, tb)
        out = ctx.plan(i2)
        status, cost = ctx.fetch_status()
        _compare_status(status, cost, orun, ctx, out)
        _compare_out(out, orun.out, ctx)
        np.testing.assert_allclose(ctx.fetch_states(), orun.states, rtol=0, atol=STATE_ATOL)


@pytest.mark.parametrize("name,lo,count,low_vel", [("cfg2", 0, 7440, False), ("cfg3", 0, 4000, False), ("cfg3", 30000, 4000, False),
                                                   ("cfg3", 58000, 2543, False), ("cfg4", 250000, 3000, False),
                                                   ("cfg5obs", 500000, 3000, False), ("cfg3", 12000, 3000, True),
                                                   ("cfg2rb", 0, 7440, False), ("cfg3rb", 20000, 3000, False),
                                                   ("cfg4rb", 100000, 3000, False), ("cfg2rb", 0, 7440, True),
                                                   ("ramp", 0, 120, False), ("ramprb", 0, 120, False)])
def test_collision_broad_phase_at_workload_scale(ctx, name, lo, count, low_vel):
    """The benchmark workloads (many dynamic obstacles, extended horizons, curved routes): labels -- in particular
    INFEASIBLE_COLLISION -- exact against the oracle's brute-force query, in production and in draw mode.  This is
    what pins the (pair, step) broad phase of the collision query (csrc/rp_kernels.h: near_mask_step)."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    if name == "cfg5obs":
        w = W.cfg5(obstacles=50)
    elif name.endswith("rb"):   # with the road boundary: 85 / 8 thin rectangles (collision.road_boundary_obb)
        w = getattr(W, name[:-2])(road_boundary=True)
    else:
        w = W.WORKLOADS[name]()
    w.setup(ctx)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    for extra in (0, FLAG_DRAW_ALL):
        inp = _with_flags(w.inputs, extra)
        if low_vel:   # lateral motion sampled over arc length (reactive_planner.py:756-772)
            inp.params.low_vel_mode = 1
        hi = min(lo + count, inp.n_candidates)
        orun = oracle.plan(inp, tb, lo, hi, want_states=False, nthreads=8)
        out = ctx.plan(inp, lo, hi)
        status, cost = ctx.fetch_status()
        _compare_status(status, cost, orun, ctx, out)
        assert out.best_index == orun.out.best_index
        _compare_collision_counts(out, orun.out, ctx)


_MID_ORACLE = {}


@pytest.mark.parametrize("lo,count", [(20000, 3000), (9000, 12000), (40000, 16384), (1000, 2560)])
def test_state_rows_of_mid_sized_batches_at_n60(ctx, lo, count):
    """Batches of 2 560 .. 16 384 candidates at the reference's default horizon (N = 60) take two step blocks of 32 lanes on the
    default path (rp_host.hip: lanes_per_candidate; below, one wavefront per candidate; above, four blocks of 16): EVERY state row of
    such a batch against the oracle's -- labels and costs alone (test_collision_broad_phase_at_workload_scale) would not see a
    row stored to the wrong place.  cfg3 (DEU_Test, 51 obstacles), ranges of its grid, draw mode."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    w = W.cfg3()
    w.setup(ctx)
    inp = _with_flags(w.inputs, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)
    if (lo, count) not in _MID_ORACLE:   # (the same for every launch path: computed once)
        tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
        _MID_ORACLE[(lo, count)] = oracle.plan(inp, tb, lo, lo + count, want_states=True, nthreads=8)
    orun = _MID_ORACLE[(lo, count)]
    out = ctx.plan(inp, lo, lo + count)
    status, cost = ctx.fetch_status()
    _compare_status(status, cost, orun, ctx, out)
    _compare_out(out, orun.out, ctx)
    np.testing.assert_allclose(ctx.fetch_states(), orun.states, rtol=0, atol=STATE_ATOL)
    k = int(np.argmax((orun.status & 3) == 1)) if np.any((orun.status & 3) == 1) else 0
    st, s1, c1 = ctx.eval_one(lo + k)   # the re-evaluation of one candidate runs on the launch path of the batch: same bits
    np.testing.assert_array_equal(st, ctx.fetch_states(k, 1)[0])


def _dense_traffic(co, n_dyn, n_steps, seed=3, dt=0.1):
    """n_dyn constant-velocity boxes on and beside the route (some cross the ego's corridor)."""
    rng = np.random.default_rng(seed)
    dyn = np.full((n_dyn, n_steps, 5), np.nan)
    s_max = co.ref_pos[-1] - 2.0
    for j in range(n_dyn):
        s0, vel, off = rng.uniform(5.0, s_max), rng.uniform(-4.0, 12.0), rng.uniform(-6.0, 6.0)
        k0, k1 = (0, n_steps) if j % 5 else (int(rng.integers(0, 20)), int(rng.integers(25, n_steps)))   # some appear / vanish
        for k in range(k0, k1):
            s = s0 + vel * dt * k
            if not (1.0 < s < s_max):
                continue
            x, y = co.convert_to_cartesian_coords(s, off)
            kk = min(int(np.searchsorted(co.ref_pos, s, side="right")) - 1, len(co.ref_pos) - 2)
            dyn[j, k] = (x, y, co.ref_theta[kk] + rng.uniform(-0.3, 0.3), rng.uniform(1.0, 3.0), rng.uniform(0.5, 1.2))
    return dyn


@pytest.mark.parametrize("kind", ["more_than_63_dynamic", "stopping_mode", "lateral_start_velocity", "explicit_polynomials",
                                  "many_static_clusters", "explicit_polynomials_static", "static_far_apart", "static_nan_shape"])
def test_collision_mask_edge_paths(ctx, kind):
    """Branches of the (pair, step) broad phase that the benchmark workloads do not reach: the overflow bit for
    dynamic obstacles 63, 64, ...; stopping trajectories (standstill at the last valid step: extended steps are not
    bounded, every bit is set); a lateral start velocity (Hermite overshoot bound); explicit polynomials (no masks at
    all); more static shapes than cluster bits.  And of the grid over the static shapes (csrc/rp_host.hip: ensure_static_grid): explicit
    polynomials with static shapes (the grid serves every kind of plan), shapes kilometres apart (the cell size grows with
    the map; poses outside the grid), a shape without finite extent (no grid: every cluster is walked)."""
    from oracle import oracle
    from commonroad_rp_amd.collision import ObstacleTables
    from commonroad_rp_amd._capi import LON_STOPPING
    s = np.arange(0.0, 300.0, 1.0)
    path = np.stack((s, 18.0 * np.sin(s / 45.0)), axis=1)
    co, inp = _synthetic_case(path, N=40)
    n_dyn = {"more_than_63_dynamic": 90, "stopping_mode": 40}.get(kind, 12)
    tables = ObstacleTables(dyn_obb=_dense_traffic(co, n_dyn, 60), dyn_t0=0)
    if kind in ("explicit_polynomials_static", "static_far_apart", "static_nan_shape"):
        rng = np.random.default_rng(13)
        boxes = []
        for k in range(70):
            sx = rng.uniform(5.0, 290.0)
            x, y = co.convert_to_cartesian_coords(sx, rng.choice([-1, 1]) * rng.uniform(1.8, 5.0))
            boxes.append([x, y, rng.uniform(-3, 3), rng.uniform(0.3, 5.0), rng.uniform(0.05, 0.5)])
        if kind == "static_far_apart":
            boxes += [[9000.0, -7000.0, 0.3, 2.0, 1.0], [-8000.0, 6000.0, 1.0, 400.0, 0.5]]
        if kind == "static_nan_shape":
            boxes = boxes[:10] + [[float("nan"), 0.0, 0.0, 1.0, 1.0]]
        tables = ObstacleTables(static_obb=boxes, static_circ=[[40.0, 3.0, 0.5]], dyn_obb=tables.dyn_obb[:4], dyn_t0=0)
    if kind == "many_static_clusters":
        rng = np.random.default_rng(11)
        boxes, tris, circs = [], [], []
        for k in range(150):   # 150 + 60 + 40 shapes -> 5 per cluster
            sx = rng.uniform(5.0, 290.0)
            x, y = co.convert_to_cartesian_coords(sx, rng.choice([-1, 1]) * rng.uniform(1.5, 6.0))
            boxes.append([x, y, rng.uniform(-3, 3), rng.uniform(0.3, 6.0), rng.uniform(0.05, 0.6)])
            if k < 60:
                tris.append([x + 1, y + 1, x + 2.5, y + 1.2, x + 1.5, y + 2.6])
            if k < 40:
                circs.append([x - 2.0, y + 1.0, rng.uniform(0.2, 0.7)])
        tables = ObstacleTables(static_obb=boxes, static_tri=tris, static_circ=circs, dyn_obb=tables.dyn_obb, dyn_t0=0)
    p = copy_params(inp.params)
    if kind == "stopping_mode":
        p.lon_mode = LON_STOPPING
        inp = PlanInputs(p, inp.cost, inp.T, inp.traj_len, np.linspace(p.x0_lon[0] + 10.0, p.x0_lon[0] + 40.0, 7), inp.D)
    if kind == "lateral_start_velocity":
        p.x0_lat[1], p.x0_lat[2] = 0.9, -0.4
        inp = PlanInputs(p, inp.cost, inp.T, inp.traj_len, inp.L, inp.D)
    ctx.set_coordinate_system(co)
    ctx.set_obstacles(tables)
    tb = oracle.OracleTables.from_coordinate_system(co, tables)
    for extra in (0, FLAG_DRAW_ALL):
        i2 = _with_flags(inp, extra)
        if kind.startswith("explicit_polynomials"):
            oc = oracle.plan(i2, tb)   # coefficients of every candidate from the oracle, then the coefficient entry
            lon_T = np.repeat(i2.T, len(i2.L) * len(i2.D))
            tl = np.repeat(i2.traj_len, len(i2.L) * len(i2.D))
            orun = oracle.plan_coeffs(i2.params, i2.cost, tb, oc.coeffs[:, 0:6], oc.coeffs[:, 6:12], tl)
            out = ctx.plan_coeffs(i2.params, i2.cost, oc.coeffs[:, 0:6], oc.coeffs[:, 6:12], lon_T, tl)
        else:
            orun = oracle.plan(i2, tb)
            out = ctx.plan(i2)
        status, cost = ctx.fetch_status()
        _compare_status(status, cost, orun, ctx, out)
        assert out.best_index == orun.out.best_index
        _compare_collision_counts(out, orun.out, ctx)
    assert orun.out.n_collision > 0


@pytest.mark.parametrize("level", [1, 2, 3])
def test_cfg1_reference_example_all_levels(ctx, level):
    """BASELINE.json configs[0]: ZAM_Over-1_1 with the reference's own sampling levels (120 / 630 / 3 060 candidates,
    N = 20, one static obstacle), production and draw mode, with and without the road boundary: every label, reason, cost,
    the counters, the winner and every state row against the oracle."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    for rb in (False, True):
        w = W.cfg1(level=level, road_boundary=rb)
        assert w.n_candidates == {1: 120, 2: 630, 3: 3060}[level]
        w.setup(ctx)
        tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
        for extra in (0, FLAG_MATERIALIZE_ALL, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL):
            inp = _with_flags(w.inputs, extra)
            orun = oracle.plan(inp, tb, want_states=True, nthreads=4)
            out = ctx.plan(inp)
            status, cost = ctx.fetch_status()
            _compare_status(status, cost, orun, ctx, out)
            _compare_out(out, orun.out, ctx)
            if extra:
                lab = orun.status & 3
                defined = np.ones_like(lab, dtype=bool) if extra & FLAG_DRAW_ALL else ((lab == 1) | (lab == 3))
                np.testing.assert_allclose(ctx.fetch_states()[defined], orun.states[defined], rtol=0, atol=STATE_ATOL)
        assert orun.out.n_feasible > 0


@pytest.mark.parametrize("name", ["cfg3", "cfg4", "cfg5", "cfg5obs"])
def test_full_size_workloads(ctx, name):
    """BASELINE.json's cfg3 / cfg4 / cfg5 at full size (62 496, 512 064 and 1 009 899 candidates; cfg5 without and with 50
    synthetic obstacles): every label, first-failure reason and cost, the counters and the winner against the oracle
    (OpenMP over candidates)."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    w = W.cfg5(obstacles=50) if name == "cfg5obs" else W.WORKLOADS[name]()
    w.setup(ctx)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    orun = oracle.plan(w.inputs, tb, want_states=False, nthreads=8)
    out = ctx.plan(w.inputs)
    status, cost = ctx.fetch_status()
    _compare_status(status, cost, orun, ctx, out)
    assert out.best_index == orun.out.best_index and out.n_feasible == orun.out.n_feasible
    _compare_collision_counts(out, orun.out, ctx, _lazy_relaxed(status, cost, orun, ctx, out)[1])
    np.testing.assert_array_equal(out.reason_counts[1:7], orun.out.reason_counts[1:7])
    if out.best_index >= 0:   # the winner's states, re-evaluated by the oracle
        one = oracle.plan(w.inputs, tb, out.best_index, out.best_index + 1)
        np.testing.assert_allclose(out.best_states, one.states[0], rtol=0, atol=STATE_ATOL)


@pytest.mark.parametrize("N", [1000, 4094])
def test_longest_horizons(ctx, N):
    """The interface's maximum horizon (N = 4094, rp_plan rejects more) and a long one in between: many step blocks per
    candidate with carried state, dynamic obstacle rows as long as the horizon."""
    import math
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import make_params, make_cost, COST_DEFAULT, LON_VELOCITY_KEEPING
    from commonroad_rp_amd.collision import ObstacleTables
    from commonroad_rp_amd.coordinate_system import CoordinateSystem
    from oracle import oracle
    s = np.arange(0.0, 3000.0, 1.0)
    co = CoordinateSystem(np.stack((s, 30.0 * np.sin(s / 200.0)), 1))
    dt = 0.05
    T = np.array([dt * k for k in (N // 3, N // 2, N)])
    p = make_params(dt=dt, N=N, factor=1, time_step0=0, low_vel_mode=False, lon_mode=LON_VELOCITY_KEEPING, constraint_mask=31,
                    flags=FLAG_MATERIALIZE_ALL, x0_lon=[20.0, 12.0, 0.0], x0_lat=[0.3, 0.0, 0.0], x0_orientation=0.15, **W.VEHICLE2)
    cost = make_cost(COST_DEFAULT, w_a=1.0, desired_speed=12.0, desired_d=0.0, desired_s=None)
    inp = PlanInputs(p, cost, T, W.traj_len_of(T, dt), np.linspace(8, 14, 5), np.linspace(-2, 2, 5))
    dyn = np.full((1, N + 1, 5), np.nan)
    for q in range(N + 1):
        sq = 60 + 10.0 * q * dt
        dyn[0, q] = (sq, 3.0 + 30.0 * math.sin(sq / 200.0), 0.1, 2.2, 0.9)
    obs = ObstacleTables(dyn_obb=dyn, dyn_t0=0)
    orun = oracle.plan(inp, oracle.OracleTables.from_coordinate_system(co, obs), want_states=True)
    ctx.set_coordinate_system(co)
    ctx.set_obstacles(obs)
    out = ctx.plan(inp)
    status, cost_g = ctx.fetch_status()
    _compare_status(status, cost_g, orun, ctx, out)
    assert out.best_index == orun.out.best_index >= 0
    have = ((orun.status & 3) == 1) | ((orun.status & 3) == 3)
    assert have.sum() > 20 and ((orun.status & 3) == 3).any()
    np.testing.assert_allclose(ctx.fetch_states()[have], orun.states[have], rtol=0, atol=STATE_ATOL)
    with pytest.raises(Exception):
        p2 = copy_params(p)
        p2.N = 4095
        ctx.plan(PlanInputs(p2, cost, T, W.traj_len_of(T, dt), inp.L, inp.D))


@pytest.mark.parametrize("name", ["arc_hv_l2_obs", "arc_all_collide", "rand_07_stop", "rand_10_slow_hv", "scurve_hv_l3"])
def test_cost_range_entry(ctx, name):
    """rp_cost_range (TrajectoryBundle.min_costs / max_costs of the standstill branch, reactive_planner.py:650-651): range over
    the candidates that have a cost -- feasible and colliding ones -- without fetching the arrays; also for a shard and after
    plug-in costs were applied."""
    g = Golden(name)
    g.setup_context(ctx)
    C = g.inputs.n_candidates
    for lo, hi in ((0, C), (C // 3, (2 * C) // 3)):
        ctx.plan(g.inputs, lo, hi)
        status, cost = ctx.fetch_status()
        have = (((status & 3) == 1) | ((status & 3) == 3)) & ~np.isnan(cost)
        cmin, cmax, n = ctx.cost_range()
        assert n == int(have.sum())
        if n:
            assert cmin == cost[have].min() and cmax == cost[have].max()
        else:
            assert np.isnan(cmin) and np.isnan(cmax)
    ctx.plan(_with_flags(g.inputs, FLAG_MATERIALIZE_ALL))
    status, cost = ctx.fetch_status()
    have = ((status & 3) == 1) | ((status & 3) == 3)
    user = np.where(have, -3.0 + np.arange(C) * 0.5, np.nan)    # negative values too: the keys order them
    ctx.select(user)
    cmin, cmax, n = ctx.cost_range()
    if have.any():
        assert (cmin, cmax, n) == (user[have].min(), user[have].max(), int(have.sum()))


def test_fetch_states_in_pieces(ctx):
    """rp_fetch_states of padded rows compacts on the device in pieces of at most 256 MB: every block of cfg3 in draw mode
    (427 MB: two pieces; the boundary lies at candidate 39 291), compared with the oracle around the boundary and at both ends."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    w = W.cfg3()
    w.setup(ctx)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    inp = _with_flags(w.inputs, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)
    ctx.plan(inp)
    states = ctx.fetch_states()
    C = inp.n_candidates
    assert states.shape == (C, 14, inp.params.N + 1)
    piece = (256 << 20) // (14 * (inp.params.N + 1) * 8)
    compared = 0
    for lo, hi in ((0, 64), (piece - 64, piece + 64), (C - 64, C)):
        ref = oracle.plan(inp, tb, lo, hi, want_states=True)
        keep = ((ref.status & 3) == 1) | ((ref.status & 3) == 3)      # blocks the reference keeps whole
        np.testing.assert_allclose(states[lo:hi][keep], ref.states[keep], rtol=0, atol=STATE_ATOL)
        compared += int(keep.sum())
    assert compared > 0
    part = ctx.fetch_states(piece - 3, 6)                             # a range that starts inside one piece of the whole
    np.testing.assert_array_equal(part, states[piece - 3:piece + 3])


def test_static_grid_follows_vehicle_and_tables(ctx):
    """The grid over the static shapes is built for the ego rectangle of the plan and kept while the static shapes stay
    (csrc/rp_host.hip: ensure_static_grid): a larger vehicle after a smaller one rebuilds it, a smaller one after a larger one
    reuses it (conservative), new dynamic obstacles with the same static shapes keep it, new static shapes replace it."""
    from oracle import oracle
    from commonroad_rp_amd.collision import ObstacleTables
    s = np.arange(0.0, 300.0, 1.0)
    path = np.stack((s, 18.0 * np.sin(s / 45.0)), axis=1)
    co, inp = _synthetic_case(path, N=30)
    rng = np.random.default_rng(21)

    def boxes(n):
        out = []
        for _ in range(n):
            x, y = co.convert_to_cartesian_coords(rng.uniform(60.0, 200.0), rng.choice([-1, 1]) * rng.uniform(1.8, 4.5))
            out.append([x, y, rng.uniform(-3, 3), rng.uniform(0.3, 1.5), rng.uniform(0.05, 0.6)])
        return out
    static_a, static_b = boxes(40), boxes(26)
    dyn1, dyn2 = _dense_traffic(co, 1, 40, seed=5), _dense_traffic(co, 2, 40, seed=6)
    ctx.set_coordinate_system(co)
    hits = []
    for length, width, static, dyn in ((2.0, 1.0, static_a, dyn1), (6.0, 2.5, static_a, dyn1), (3.0, 1.5, static_a, dyn2),
                                       (4.5, 1.8, static_b, dyn2), (4.5, 1.8, static_a, dyn1)):
        tables = ObstacleTables(static_obb=static, dyn_obb=dyn, dyn_t0=0)
        ctx.set_obstacles(tables)
        tb = oracle.OracleTables.from_coordinate_system(co, tables)
        p = copy_params(inp.params)
        p.length, p.width = length, width
        i2 = PlanInputs(p, inp.cost, inp.T, inp.traj_len, inp.L, inp.D)
        orun = oracle.plan(i2, tb)
        out = ctx.plan(i2)
        status, cost = ctx.fetch_status()
        _compare_status(status, cost, orun, ctx, out)
        assert out.best_index == orun.out.best_index
        _compare_collision_counts(out, orun.out, ctx)
        hits.append(orun.out.n_collision)
    assert max(hits) > 0 and len(set(hits)) > 1
