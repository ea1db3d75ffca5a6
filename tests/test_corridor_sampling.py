"""Data-dependent sampling space on the explicit-polynomial entry (rp_plan_coeffs): ``CorridorSampling``
(reference: commonroad_rp/sampling.py:273-397) -- a different number of lateral samples per (T, v), none where the
corridor is not reached -- against fixtures produced by the reference's OWN CorridorSampling and _get_optimal_trajectory
running on this build's duck-typed corridor (tests/golden/make_golden.py::run_corridor_case; CommonRoad-Reach itself is
not available).  CPU: candidate list, oracle labels / costs / winner.  GPU: every launch path against oracle and fixture."""
import glob
import os

import numpy as np
import pytest

from _golden import GOLDEN_DIR, build_planner_from_plan_golden
from commonroad_rp_amd._capi import FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL, make_cost, make_params
from commonroad_rp_amd.collision import ObstacleTables
from commonroad_rp_amd.config import ReactivePlannerConfiguration
from commonroad_rp_amd.corridor import ReachBox
from commonroad_rp_amd.sampling import CorridorSampling

CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "corridor_*.npz")))


def _load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def _space(z):
    cfg = ReactivePlannerConfiguration.from_dict(dict(planning=dict(dt=float(z["dt"]), time_steps_computation=int(z["N"])),
                                                      sampling=dict(t_min=float(z["t_min"]), sampling_method=2)))
    sp = CorridorSampling(cfg)
    cor = {}
    for row in z["corridor"]:
        cor.setdefault(int(row[0]), []).append(ReachBox(*row[1:]))
    sp.driving_corridor = cor
    return sp


def _params(z, flags=0):
    veh = z["vehicle"]
    p = make_params(dt=float(z["dt"]), N=int(z["N"]), factor=int(z["factor"]), time_step0=int(z["time_step0"]), low_vel_mode=False,
                    lon_mode=0, constraint_mask=31, flags=flags | (FLAG_DRAW_ALL if int(z["draw"]) else 0), x0_lon=z["x0_lon"],
                    x0_lat=z["x0_lat"], x0_orientation=float(z["x0_orientation"]), wheelbase=veh[0], wb_rear_axle=veh[1], length=veh[2],
                    width=veh[3], a_max=veh[4], v_switch=veh[5], delta_max=veh[6], v_delta_max=veh[7])
    cost = make_cost(w_a=float(z["w_a"]), desired_speed=float(z["desired_speed"]), desired_d=float(z["desired_d"]))
    return p, cost


def _tables(z):
    return ObstacleTables(static_obb=z["static_obb"], static_tri=z["static_tri"], static_circ=z["static_circ"], dyn_obb=z["dyn_obb"],
                          dyn_t0=int(z["dyn_t0"]))


def test_fixture_inventory():
    assert len(CASES) >= 3


@pytest.mark.parametrize("name", CASES)
def test_candidate_list_follows_the_reference(name):
    """same candidates, same order (sets of floats for T, v, d; parts of the corridor in the order of the reach operations),
    in the object view and in the batch view"""
    z = _load(name)
    sp = _space(z)
    lvl = int(z["level"])
    trajs = sp.generate_trajectories_at_level(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    lon, lat, T, tl, v_end, d_end = sp.coeffs_at_level(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    assert len(trajs) == len(T) == len(z["lon_T"])
    np.testing.assert_array_equal(T, z["lon_T"])
    np.testing.assert_array_equal(v_end, z["lon_end"])
    np.testing.assert_array_equal(d_end, z["lat_end"])
    np.testing.assert_array_equal(np.array([t.trajectory_long.delta_tau for t in trajs]), z["lon_T"])
    np.testing.assert_allclose(np.array([t.trajectory_long.coeffs for t in trajs]), z["lon_coeffs"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.array([t.trajectory_lat.coeffs for t in trajs]), z["lat_coeffs"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lon, z["lon_coeffs"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(lat, z["lat_coeffs"], rtol=1e-9, atol=1e-10)
    np.testing.assert_array_equal(tl, np.round(T / float(z["dt"])).astype(int) + 1)
    # irregular: the number of lateral samples differs between longitudinal samples
    counts = {}
    for t, v in zip(T.tolist(), v_end.tolist()):
        counts[(t, v)] = counts.get((t, v), 0) + 1
    assert len(set(counts.values())) >= 2


def _check_against_fixture(status, cost, out, z):
    mine, ref = status & 3, z["label"]
    # the reference checks collisions lazily, in cost order, up to the winner: a colliding sample that sorts after the winner
    # keeps the label FEASIBLE there and carries INFEASIBLE_COLLISION here (eager check); everything else is label for label
    # (draw mode stores out-of-domain samples as kinematically infeasible here, without a label there)
    np.testing.assert_array_equal(mine == 2, (ref == 2) | ((ref == 0) & (mine == 2) & bool(int(z["draw"]))))
    np.testing.assert_array_equal((mine == 1) | (mine == 3), (ref == 1) | (ref == 3))
    assert np.all(mine[ref == 3] == 3)
    has = ~np.isnan(z["cost"])
    np.testing.assert_allclose(cost[has], z["cost"][has], rtol=1e-9)
    assert out.best_index == int(z["winner"])
    assert out.n_collision_before_best == int(z["n_infeasible_collision"])
    assert out.n_candidates - out.n_feasible == int(z["n_infeasible_kinematics"])
    np.testing.assert_array_equal(out.reason_counts[1:6], z["reason_counts"])
    if out.best_index >= 0:
        np.testing.assert_allclose(out.best_cost, float(z["winner_cost"]), rtol=1e-9)


@pytest.mark.parametrize("name", CASES)
def test_oracle_on_the_reference_candidates(name):
    from oracle import oracle
    z = _load(name)
    p, cost = _params(z)
    tb = oracle.OracleTables(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]), _tables(z))
    tl = np.round(z["lon_T"] / float(z["dt"])).astype(np.int32) + 1
    run = oracle.plan_coeffs(p, cost, tb, z["lon_coeffs"], z["lat_coeffs"], tl)
    _check_against_fixture(run.status, run.cost, run.out, z)


@pytest.mark.parametrize("name", CASES)
def test_planner_with_corridor_sampling_cpu_glue(name):
    """the planner's per-level hot path with this sampling space (batch view -> rp_plan_coeffs), oracle-backed context"""
    from _oracle_ctx import OracleContext
    z = _load(name)
    z.setdefault("low_vel_mode_threshold", np.float64(4.0))
    rp, _ = build_planner_from_plan_golden(dict(z, continuous=0, lon_mode=0, v_range=np.array([np.nan, np.nan])), OracleContext)
    rp._draw_traj_set = bool(int(z["draw"]))
    rp.set_sampling_space(_space(z))
    bundle = rp._create_trajectory_bundle(rp.x_0_cl[0], rp.x_0_cl[1], samp_level=int(z["level"]))
    assert bundle.coeffs is not None and bundle.n_candidates == len(z["lon_T"])
    opt = rp._get_optimal_trajectory(bundle)
    assert (opt is not None) == (int(z["winner"]) >= 0)
    assert rp.infeasible_count_collision == int(z["n_infeasible_collision"])
    assert rp.infeasible_count_kinematics == int(z["n_infeasible_kinematics"])
    if opt is not None:
        w = int(z["winner"])
        np.testing.assert_allclose(opt.cost, float(z["winner_cost"]), rtol=1e-9)
        np.testing.assert_allclose(opt.trajectory_lat.coeffs, z["lat_coeffs"][w], rtol=1e-9, atol=1e-10)
        assert opt.trajectory_long.delta_tau == z["lon_T"][w] and opt.trajectory_lat.x_d[0] == z["lat_end"][w]
    if rp._draw_traj_set:
        assert len(rp.stored_trajectories) == len(z["lon_T"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_corridor_candidates_on_every_launch_path(name):
    from _paths import LAUNCH_PATHS, launch_path_env
    from commonroad_rp_amd._capi import RpContext
    from oracle import oracle
    z = _load(name)
    sp = _space(z)
    lon, lat, T, tl, _, _ = sp.coeffs_at_level(int(z["level"]), z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    tabs = _tables(z)
    tb = oracle.OracleTables(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]), tabs)
    for path in LAUNCH_PATHS:
        with launch_path_env(path):
            ctx = RpContext(0)
            ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]))
            ctx.set_obstacles(tabs)
            # every case twice: plain arrays (one longitudinal profile per candidate), and written into the context's arena with the
            # groups the batch view reports (rp_plan_coeffs_grouped: one profile per (T, v) sample) -- where this interpreter's sets
            # allow the native batch view
            spa = _space(z)
            grouped = spa._native_ok()
            if grouped:
                spa.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
                alon, alat, aT, atl, _, _ = spa.coeffs_at_level(int(z["level"]), z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
                groups = spa._last_groups[1]
                assert groups is not None and groups[0] > 0
            for extra, use_groups in ((0, False), (FLAG_MATERIALIZE_ALL, False)) + (((0, True), (FLAG_MATERIALIZE_ALL, True)) if grouped else ()):
                p, cost = _params(z, extra)
                orun = oracle.plan_coeffs(p, cost, tb, lon, lat, tl)
                out = ctx.plan_coeffs(p, cost, alon, alat, aT, atl, groups=groups) if use_groups else ctx.plan_coeffs(p, cost, lon, lat, T, tl)
                status, c = ctx.fetch_status()
                from _lazy import lazy_relaxed
                _check_against_fixture(status, c, out, z)             # (the reference's own labels are lazy: raw labels against them)
                status, _ = lazy_relaxed(status, c, orun, ctx, out)   # (cost-ordered stage: labels of the candidates it looked at)
                np.testing.assert_array_equal(status & 0x7F, orun.status & 0x7F)
                if out.best_index >= 0:
                    np.testing.assert_allclose(out.best_states, orun.out.best_states, rtol=0, atol=1e-6)
                if extra:
                    have = ((orun.status & 3) == 1) | ((orun.status & 3) == 3) | bool(int(z["draw"]))
                    np.testing.assert_allclose(ctx.fetch_states()[have], orun.states[have], rtol=0, atol=1e-6)
            ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES[:2])
def test_candidates_written_into_the_contexts_arena(name):
    """rp_coeffs_arena: the library's batch view writes into pinned arrays of the context, rp_plan_coeffs takes them as they are --
    same result as with ordinary arrays, also after the arena had to grow and on a second plan that reuses it"""
    from commonroad_rp_amd._capi import RpContext
    z = _load(name)
    sp = _space(z)
    level, args = int(z["level"]), (z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    lon, lat, T, tl, _, _ = (np.array(a) for a in sp.coeffs_at_level(level, *args))
    tabs = _tables(z)
    ctx = RpContext(0)
    ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]))
    ctx.set_obstacles(tabs)
    p, cost = _params(z, FLAG_MATERIALIZE_ALL)
    want = ctx.plan_coeffs(p, cost, lon, lat, T, tl)
    want_status, want_cost = ctx.fetch_status()
    sp2 = _space(z)
    if not sp2._native_ok():
        pytest.skip("this interpreter's set order is not the library's: no native batch view")
    calls = []

    def alloc(cap):
        calls.append(cap)
        return ctx.coeffs_arena(cap)
    sp2.__dict__["_native_buffers"] = {"alloc": alloc, "cap": 64}    # (too small on purpose: the first call has to grow)
    for rep in range(3):
        alon, alat, aT, atl, _, _ = sp2.coeffs_at_level(level, *args)
        base = ctx.coeffs_arena(calls[-1])
        groups = sp2._last_groups[1]
        # candidates of a group are adjacent and share their longitudinal polynomial and traj_len; group_first points at their first
        ng, gof, gfirst = groups
        assert ng == len(gfirst) and len(gof) == len(aT) and gof[0] == 0 and gof[-1] == ng - 1 and np.all(np.diff(gof) >= 0) and np.all(np.diff(gof) <= 1)
        np.testing.assert_array_equal(alon, alon[gfirst[gof]])
        np.testing.assert_array_equal(atl, atl[gfirst[gof]])
        assert ng < len(aT)
        assert alon.ctypes.data == base[0].ctypes.data and alat.ctypes.data == base[1].ctypes.data and atl.ctypes.data == base[2].ctypes.data
        np.testing.assert_array_equal(alon, lon)
        # rep 0 / 2: one longitudinal profile per group (rp_plan_coeffs_grouped); rep 1: one per candidate -- same results
        out = ctx.plan_coeffs(p, cost, alon, alat, aT, atl, groups=groups if rep != 1 else None)
        status, c = ctx.fetch_status()
        assert out.best_index == want.best_index and out.n_feasible == want.n_feasible
        np.testing.assert_array_equal(status, want_status)
        if rep == 1:   # the same launch path as `want` (a batch this small takes the single-launch kernel): bit for bit
            np.testing.assert_array_equal(c, want_cost)
        else:          # grouped plans take the two-kernel path: its sums round differently in the last bit (tolerances of test_gpu_parity)
            np.testing.assert_allclose(c, want_cost, rtol=1e-12, atol=1e-9)
        if want.best_index >= 0:
            if rep == 1:
                np.testing.assert_array_equal(out.best_states, want.best_states)
            else:
                np.testing.assert_allclose(out.best_states, want.best_states, rtol=0, atol=1e-9)
            np.testing.assert_array_equal(out.best_lon_coeffs, lon[want.best_index])
            np.testing.assert_array_equal(out.best_lat_coeffs, lat[want.best_index])
    assert len(calls) >= 2 and calls[-1] >= len(T)
    # (the last plan was a grouped one) a single candidate re-evaluated on the device: its rows are the batch's
    states = ctx.fetch_states()
    have = np.flatnonzero((want_status & 3) == 1)    # (rows of infeasible candidates are not materialised outside draw mode)
    for idx in (have[0], have[len(have) // 2], have[-1]):
        np.testing.assert_array_equal(ctx.eval_one(int(idx))[0], states[idx])
    ctx.close()


def test_library_restates_the_interpreters_set_order():
    """The reference's candidate order is the iteration order of Python sets of floats (sampling.py:367,384-386); the library's
    batch view restates CPython's set (csrc/rp_corridor.h).  Against this interpreter's own sets: plain sets of np.linspace values
    of every size the levels use, with duplicates (a degenerate interval), and ``.union({0})`` -- a copy (different table size!)
    plus the reference path."""
    from commonroad_rp_amd import _capi
    rng = np.random.default_rng(7)
    for trial in range(1500):
        n = int(rng.choice([3, 5, 9, 17, 21, 33, 65, 129, 257]))
        lo = float(rng.uniform(-50.0, 50.0))
        hi = lo if trial % 40 == 0 else lo + float(rng.uniform(0.0, 60.0))
        v = np.linspace(lo, hi, n)
        np.testing.assert_array_equal(_capi.pyset_order(v), np.array(list(set(v)), dtype=float))
        w = np.linspace(-abs(lo) - 0.25, abs(hi) + 0.25, n)
        np.testing.assert_array_equal(_capi.pyset_order(w, True), np.array([float(x) for x in set(w).union({0})]))
    assert CorridorSampling._native_state in (None, True)


@pytest.mark.parametrize("name", CASES)
def test_library_batch_view_equals_the_numpy_one(name):
    """rp_corridor_coeffs against the NumPy batch view: same candidates in the same order (T, end velocity, end offset bit for
    bit), coefficients to 1e-12 (closed-form quartic against the LAPACK solve the reference uses)."""
    z = _load(name)
    sp = _space(z)
    lvl = int(z["level"])
    assert sp._native_ok()
    a = sp.coeffs_at_level(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    b = sp._coeffs_at_level_py(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    for k in (2, 3, 4, 5):
        np.testing.assert_array_equal(a[k], b[k])
    np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-12)
    # a larger synthetic corridor (three lanes that merge and split over time), several levels of sample counts
    s0, v0 = float(z["x0_lon"][0]), max(float(z["x0_lon"][1]), 1.0)
    cor = {}
    for q in range(int(z["N"]) + 2):
        t = q * float(z["dt"])
        cor[int(z["time_step0"]) + q] = [ReachBox(s0 - 1.0, s0 + 1.6 * v0 * t + 4.0, -2.8, -0.3, 0.5 * v0, 1.3 * v0 + 1.0),
                                        ReachBox(s0 + 0.3 * v0 * t, s0 + 1.8 * v0 * t + 6.0, -0.5, 1.4, 0.6 * v0, 1.4 * v0 + 1.0),
                                        ReachBox(s0 + 0.8 * v0 * t + 2.0, s0 + 2.0 * v0 * t + 8.0, 2.0, 3.0, 0.9 * v0, 1.5 * v0 + 1.0)]
    sp.driving_corridor = cor
    for n_samples in (3, 9, 21):
        sp._dict_level_to_num_samples = {k: n_samples for k in range(sp.num_sampling_levels)}
        a = sp.coeffs_at_level(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
        b = sp._coeffs_at_level_py(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
        assert len(a[2]) == len(b[2]) > 0
        for k in (2, 3, 4, 5):
            np.testing.assert_array_equal(a[k], b[k])
        np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-12)


def test_library_batch_view_from_concurrent_callers():
    """rp_corridor_coeffs runs its time samples on the library's pool of host threads (csrc/rp_pool.h), one job at a time: calls
    from several Python threads at once (ctypes releases the GIL) must each get the complete list in the reference's order"""
    import threading
    z = _load(CASES[0])
    sp0 = _space(z)
    if not sp0._native_ok():
        pytest.skip("this interpreter's set order is not the library's: no native batch view")
    lvl = int(z["level"])
    want = [np.array(a) for a in sp0._coeffs_at_level_py(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)]
    errors = []

    def caller():
        try:
            sp = _space(z)     # (its own output buffers)
            for _ in range(40):
                got = sp.coeffs_at_level(lvl, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
                for k in (2, 3, 4, 5):
                    np.testing.assert_array_equal(got[k], want[k])
                np.testing.assert_allclose(got[0], want[0], rtol=1e-12, atol=1e-12)
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=caller) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


@pytest.mark.gpu
def test_grouped_plan_equals_ungrouped_on_a_large_corridor():
    """25 536 candidates in 651 (T, v) groups on cfg3's route and obstacles (the level bench.py's corridor record times): one
    longitudinal profile per group against one per candidate -- labels, costs, counters, winner; production and materialising mode"""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import copy_params
    base = W.WORKLOADS["cfg3"]()
    rp = W.make_planner(base, device=0)
    p = base.inputs.params
    sp = CorridorSampling(rp.config)
    sp.samples_t._dict_level_to_sample_set[1] = set(float(t) for t in base.inputs.T)
    sp._dict_level_to_num_samples = {k: 21 for k in range(rp.config.sampling.num_sampling_levels)}
    s0, v0 = p.x0_lon[0], max(p.x0_lon[1], 1.0)
    cor = {}
    for q in range(p.N + 2):
        t = q * p.dt
        cor[p.time_step0 + q] = [ReachBox(s0 - 1.0, s0 + 1.6 * v0 * t + 4.0, -2.8, -0.3, 0.5 * v0, 1.3 * v0 + 1.0),
                                 ReachBox(s0 + 0.3 * v0 * t, s0 + 1.8 * v0 * t + 6.0, -0.5, 1.4, 0.6 * v0, 1.4 * v0 + 1.0),
                                 ReachBox(s0 + 0.8 * v0 * t + 2.0, s0 + 2.0 * v0 * t + 8.0, 2.0, 3.0, 0.9 * v0, 1.5 * v0 + 1.0)]
    sp.driving_corridor = cor
    if not sp._native_ok():
        pytest.skip("this interpreter's set order is not the library's: no native batch view")
    ctx = rp._gpu_ctx()
    sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
    x0_lon, x0_lat = rp.x_0_cl
    lon, lat, T, tl, _, _ = sp.coeffs_at_level(1, x0_lon, x0_lat, rp.config.sampling.longitudinal_mode, False)
    groups = sp._last_groups[1]
    assert len(T) > 20000 and 0 < groups[0] < len(T) // 10
    cost = rp._gpu_cost()
    for flags in (0, FLAG_MATERIALIZE_ALL):
        prm = copy_params(rp._gpu_params(x0_lon, x0_lat, flags))
        res = {}
        for key, g in (("plain", None), ("grouped", groups)):
            ctx.set_option("lazy", 0)      # (labels of both runs from the eager query: comparable one by one)
            try:
                out = ctx.plan_coeffs(prm, cost, lon, lat, T, tl, groups=g)
            finally:
                ctx.set_option("lazy", -1)
            status, c = ctx.fetch_status()
            res[key] = (out, status.copy(), c.copy(), ctx.fetch_states().copy() if flags else None)
        a, b = res["plain"], res["grouped"]
        assert a[0].best_index == b[0].best_index >= 0 and a[0].n_feasible == b[0].n_feasible
        assert a[0].n_collision == b[0].n_collision and a[0].n_collision_before_best == b[0].n_collision_before_best
        np.testing.assert_array_equal(a[0].reason_counts, b[0].reason_counts)
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])            # (the same launch path, the same profile values: bit for bit)
        np.testing.assert_array_equal(a[0].best_states, b[0].best_states)
        if flags:
            have = (a[1] & 3) == 1
            np.testing.assert_array_equal(a[3][have], b[3][have])
    rp.close()


@pytest.mark.gpu
def test_grouped_plan_refuses_malformed_group_tables():
    """the kernels index through the group tables: rp_plan_coeffs_grouped looks at every entry before it launches anything"""
    from commonroad_rp_amd._capi import RpContext, RpError
    z = _load(CASES[0])
    sp = _space(z)
    if not sp._native_ok():
        pytest.skip("no native batch view with this interpreter's set order")
    ctx = RpContext(0)
    ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]))
    ctx.set_obstacles(_tables(z))
    sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
    lon, lat, T, tl, _, _ = sp.coeffs_at_level(int(z["level"]), z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
    ng, gof, gfirst = sp._last_groups[1]
    p, cost = _params(z, 0)
    want = ctx.plan_coeffs(p, cost, lon, lat, T, tl, groups=(ng, gof, gfirst)).best_index
    for spoil in ("gap", "order", "first", "range"):
        keep_of, keep_first = gof.copy(), gfirst.copy()
        if spoil == "gap":
            gof[len(gof) // 2:] += 1
        elif spoil == "order":
            gof[1], gof[-1] = gof[-1], gof[1]
        elif spoil == "first":
            gfirst[1] += 1
        else:
            gof[-1] = ng + 5
        with pytest.raises(RpError):
            ctx.plan_coeffs(p, cost, lon, lat, T, tl, groups=(ng, gof, gfirst))
        gof[:], gfirst[:] = keep_of, keep_first
    assert ctx.plan_coeffs(p, cost, lon, lat, T, tl, groups=(ng, gof, gfirst)).best_index == want
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_corridors_grouped_against_the_oracle(seed):
    """random corridors (one to four reach boxes per time step, overlapping or apart; 3 .. 11 samples per
    interval) on a fixture's route and obstacles: the grouped device plan out of the arena against the oracle on the same candidates --
    on the two-kernel paths (eager and cost-ordered), where the groups are used whatever the batch size, and on the default one"""
    from _paths import launch_path_env
    from commonroad_rp_amd._capi import RpContext
    from oracle import oracle
    rng = np.random.default_rng(1000 + seed)
    z = _load(CASES[seed % len(CASES)])
    sp = _space(z)
    if not sp._native_ok():
        pytest.skip("no native batch view with this interpreter's set order")
    s0, v0 = float(z["x0_lon"][0]), max(float(z["x0_lon"][1]), 1.0)
    dt, N, t0 = float(z["dt"]), int(z["N"]), int(z["time_step0"])
    cor = {}
    for q in range(N + 2):
        t = q * dt
        boxes = []
        for _ in range(int(rng.integers(1, 5))):
            lo = s0 + rng.uniform(-1.0, 1.2) * v0 * t + rng.uniform(-2.0, 2.0)
            dlo = rng.uniform(-3.5, 2.5)
            vlo = rng.uniform(0.3, 1.0) * v0
            boxes.append(ReachBox(lo, lo + rng.uniform(0.5, 1.5) * v0 * max(t, 0.5) + 2.0, dlo, dlo + rng.uniform(0.3, 3.0), vlo,
                                  vlo + rng.uniform(0.2, 1.0) * v0 + 0.5))
        if boxes:
            cor[t0 + q] = boxes
    sp.driving_corridor = cor
    n_samples = int(rng.choice([3, 5, 9, 11]))
    sp._dict_level_to_num_samples = {k: n_samples for k in range(sp.num_sampling_levels)}
    level = int(z["level"])
    tabs = _tables(z)
    tb = oracle.OracleTables(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]), tabs)
    for path in ("single_launch", "two_kernel", "lazy"):
        with launch_path_env(path):
            ctx = RpContext(0)
            ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"], float(z["proj_d_limit"]))
            ctx.set_obstacles(tabs)
            sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
            try:
                lon, lat, T, tl, _, _ = sp.coeffs_at_level(level, z["x0_lon"], z["x0_lat"], "velocity_keeping", False)
            except KeyError:      # (a time sample of the level without a corridor entry: the reference raises there as well)
                ctx.close()
                pytest.skip("the random corridor misses a time sample of the level")
            groups = sp._last_groups[1]
            if len(T) == 0:
                ctx.close()
                continue
            assert groups is not None and 0 < groups[0] <= len(T)
            for extra in (0, FLAG_MATERIALIZE_ALL):
                p, cost = _params(z, extra)
                orun = oracle.plan_coeffs(p, cost, tb, np.array(lon), np.array(lat), np.array(tl))
                out = ctx.plan_coeffs(p, cost, lon, lat, T, tl, groups=groups)
                status, c = ctx.fetch_status()
                from _lazy import lazy_relaxed
                status, _ = lazy_relaxed(status, c, orun, ctx, out)
                np.testing.assert_array_equal(status & 0x7F, orun.status & 0x7F)
                has = ~np.isnan(orun.cost)
                np.testing.assert_allclose(c[has], orun.cost[has], rtol=1e-9)
                assert out.best_index == orun.out.best_index and out.n_feasible == orun.out.n_feasible
                assert out.n_collision_before_best == orun.out.n_collision_before_best
                if out.best_index >= 0:
                    np.testing.assert_allclose(out.best_states, orun.out.best_states, rtol=0, atol=1e-6)
            ctx.close()
