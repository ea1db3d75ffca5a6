"""The host-side mirror of the reference planner interface (commonroad_rp_amd.reactive_planner)
reproduces the reference's plan() output -- levels loop, standstill branch, output packing -- on the
golden plan_* fixtures.  CPU run: oracle-backed context (test infrastructure); GPU run: the HIP library."""
import numpy as np
import pytest

from _golden import build_planner_from_plan_golden, compare_plan_result, plan_case_names

NAMES = plan_case_names()


def test_plan_fixture_inventory():
    assert len(NAMES) >= 5


@pytest.mark.parametrize("name", NAMES)
def test_plan_matches_reference_cpu_glue(name):
    from _oracle_ctx import OracleContext
    rp, z = build_planner_from_plan_golden(name, OracleContext)
    res = rp.plan()
    compare_plan_result(res, rp, z, atol=1e-9)
    assert len(rp.planning_times) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_plan_matches_reference_gpu(name):
    from commonroad_rp_amd._capi import RpContext
    rp, z = build_planner_from_plan_golden(name, RpContext)
    res = rp.plan()
    compare_plan_result(res, rp, z, atol=1e-6)
    rp.close()


def test_sampling_grids_follow_reference_set_order():
    """grids_at_level reproduces the candidate order stored in the fixtures (Python set iteration order)."""
    from _golden import Golden
    from commonroad_rp_amd.config import ReactivePlannerConfiguration
    from commonroad_rp_amd.sampling import FixedIntervalSampling, VelocitySampling
    # (the *_d0_on_sample fixtures: the vehicle exactly on a lateral sample -- the reference iterates a COPY of the sample set then,
    #  whose order at 17 samples is not the set's own)
    for name, level, t_min in (("arc_hv_l1", 1, 0.4), ("arc_hv_l2_obs", 2, 0.4), ("scurve_hv_l3", 3, 0.2), ("arc_hv_l1_d0_on_sample", 1, 0.4),
                               ("arc_hv_l3_d0_on_sample", 3, 0.4)):
        g = Golden(name)
        cfg = ReactivePlannerConfiguration.from_dict(dict(planning=dict(dt=0.1, time_steps_computation=20),
                                                          sampling=dict(t_min=t_min)))
        sp = FixedIntervalSampling(cfg)
        h = 2.0
        v0, vd = float(g["x0_velocity"]), float(g["desired_speed"])
        min_v = max(0, v0 - (0.125 * h * 11.5))
        sp.samples_v = VelocitySampling(min_v, max(min_v + 5.0, v0 + 2), 4)
        T, tl, L, D = sp.grids_at_level(level, g["x0_lon"], g["x0_lat"], "velocity_keeping")
        np.testing.assert_array_equal(T, g["T"])
        np.testing.assert_array_equal(tl, g["traj_len"])
        np.testing.assert_array_equal(L, g["L"])
        np.testing.assert_array_equal(D, g["D"])
        # a replacement of a 1-D sample set goes through samples_at_level(), as generate_trajectories_at_level does

        class Halved(VelocitySampling):
            def samples_at_level(self, sampling_level=0):
                return set(sorted(super().samples_at_level(sampling_level))[::2])
        sp.samples_v = Halved(min_v, max(min_v + 5.0, v0 + 2), 4)
        T2, tl2, L2, D2 = sp.grids_at_level(level, g["x0_lon"], g["x0_lat"], "velocity_keeping")
        np.testing.assert_array_equal(T2, T)
        np.testing.assert_array_equal(tl2, tl)
        np.testing.assert_array_equal(D2, D)
        assert list(L2) == list(sp.samples_v.samples_at_level(level)) and len(L2) == (len(L) + 1) // 2
        ref = sp.generate_trajectories_at_level(level, g["x0_lon"], g["x0_lat"], "velocity_keeping", False)
        assert len(ref) == len(T2) * len(L2) * len(D2)


def test_output_packing_paths_agree():
    """_compute_trajectory_pair takes positions and the lon / lat lists from slices of the winner's [14, N + 1] block when the
    sample's arrays are rows of one block, and column by column otherwise (reactive_planner.py:514-568): same values."""
    from commonroad_rp_amd import workloads
    from _oracle_ctx import OracleContext
    rp = workloads.make_planner(workloads.cfg2(), backend_factory=OracleContext)
    seen = {}
    orig = rp._compute_trajectory_pair

    def spy(trajectory):
        seen["t"] = trajectory
        return orig(trajectory)
    rp._compute_trajectory_pair = spy
    fast = rp.plan()
    t = seen["t"]
    assert t._state_block is not None and t._samples is None                     # the fast path was taken: no container was built
    blk = t._state_block
    assert t.cartesian.x.base is blk and t.curvilinear.d_ddot.base is blk        # containers on access: views of the block
    t._state_block = None                                                        # force the column-by-column path
    slow = orig(t)
    assert fast[2] == slow[2] and fast[3] == slow[3]
    for a, b in zip(fast[0].state_list, slow[0].state_list):
        assert np.array_equal(a.position, b.position) and a.orientation == b.orientation and a.velocity == b.velocity
    for a, b in zip(fast[1].state_list, slow[1].state_list):
        assert np.array_equal(a.position, b.position) and a.yaw_rate == b.yaw_rate


def test_state_list_is_a_list_and_keeps_yaw_rate_none():
    """The reference's ``state_list`` is a plain list and state 0 carries ``x_0.yaw_rate`` as it is -- ``None`` for a hand-built
    state (state.py:20) (reactive_planner.py:536-539).  The lazily built list behaves the same."""
    import pickle
    from commonroad_rp_amd import workloads
    from commonroad_rp_amd.state import LazyStateList
    from _oracle_ctx import OracleContext
    rp = workloads.make_planner(workloads.cfg2(), backend_factory=OracleContext)
    rp.x_0.yaw_rate = None
    res = rp.plan()
    sl = res[0].state_list
    assert isinstance(sl, list) and isinstance(sl, LazyStateList)
    assert sl[0].yaw_rate is None and isinstance(sl[1].yaw_rate, float)
    n = len(sl)
    assert sl[1] is sl[1] and sl[-1] is sl[n - 1]
    both = sl + res[1].state_list                       # list concatenation builds what is missing
    assert type(both) is list and len(both) == 2 * n and both[2] is sl[2]
    again = pickle.loads(pickle.dumps(sl))
    assert type(again) is list and len(again) == n and again[3].velocity == sl[3].velocity
    assert [s.time_step for s in sl] == [rp.x_0.time_step + i for i in range(n)]


def _plan_summary(res, rp):
    head = (rp.infeasible_count_kinematics, rp.infeasible_count_collision, dict(rp.infeasible_reason_dict))
    if res is None:
        return head, None
    cart, cvln, lon, lat = res
    c = np.array([[st.time_step, st.position[0], st.position[1], st.orientation, st.velocity, st.acceleration, st.steering_angle]
                  + ([] if st.yaw_rate is None else [st.yaw_rate]) for st in cart.state_list], dtype=float)
    v = np.array([[st.time_step, st.position[0], st.position[1], st.orientation, st.velocity, st.acceleration, st.yaw_rate]
                  for st in cvln.state_list], dtype=float)
    return head, (c, v, np.array(lon), np.array(lat))


@pytest.mark.parametrize("name", NAMES)
def test_fast_cycle_equals_general_cycle(name):
    """``plan()`` of the stand-alone planner takes a cycle without bundle / sample / container objects when nothing needs them
    (``_plan_fast``: level loop on the arrays, output packing in ``rp_pack_trajectory``).  Same result as the general loop:
    counters and reasons exact, positions / lon / lat samples exact (copies of the same block), angles to 1e-13 (libm's atan2
    against NumPy's)."""
    from _oracle_ctx import OracleContext
    fast, _ = build_planner_from_plan_golden(name, OracleContext)
    slow, _ = build_planner_from_plan_golden(name, OracleContext)
    slow._get_optimal_trajectory = slow._get_optimal_trajectory        # (an instance attribute: the general loop)
    assert not slow._fast_path_ok()
    a, b = _plan_summary(fast.plan(), fast), _plan_summary(slow.plan(), slow)
    assert a[0] == b[0]
    assert (a[1] is None) == (b[1] is None)
    if a[1] is not None:
        for x, y in zip(a[1], b[1]):
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-13)
        np.testing.assert_array_equal(a[1][2], b[1][2])
        np.testing.assert_array_equal(a[1][3], b[1][3])


def test_level_loop_goes_to_the_device_in_one_call():
    """A cycle whose first level has no winner costs the reference a whole pass per level (reactive_planner.py:616-636).  The
    stand-alone planner hands the grids of every level the loop could visit to the context in ONE call (rp_plan_levels: one device
    round trip whatever level delivers); with an ``on_device_launched`` hook the same in two halves, the hook in between; a context
    without the levels call is asked level by level.  Results as the general loop's."""
    from _oracle_ctx import OracleContext
    log = []

    class Logged(OracleContext):
        def plan_levels_packed(self, params, cost, levels):
            log.append(("levels_packed", [len(T) * len(L) * len(D) for (T, tl, L, D) in levels]))
            return super().plan_levels_packed(params, cost, levels)

        def plan_levels_begin(self, params, cost, levels, want_best_states=True):
            log.append(("levels_begin", len(levels)))
            return super().plan_levels_begin(params, cost, levels, want_best_states)

        def plan_wait(self):
            log.append(("wait",))
            return super().plan_wait()

    class PerLevel(OracleContext):
        plan_levels_packed = plan_levels_begin = plan_packed = None

        def plan(self, inp, *a, **k):
            log.append(("plan", inp.n_candidates))
            return super().plan(inp, *a, **k)

    name = "plan_all_collide"          # (no level has a winner: plan() visits all three)
    ref, z = build_planner_from_plan_golden(name, OracleContext)
    ref._get_optimal_trajectory = ref._get_optimal_trajectory        # (an instance attribute: the general loop)
    want = _plan_summary(ref.plan(), ref)
    rp, _ = build_planner_from_plan_golden(name, Logged)
    assert rp._fast_path_ok()
    rp.level_policy = "chain"                      # every level in one call, every cycle
    assert _plan_summary(rp.plan(), rp) == want
    assert log == [("levels_packed", [120, 540, 2754])]
    del log[:]
    rp.on_device_launched = lambda: log.append(("hook",))
    assert _plan_summary(rp.plan(), rp) == want
    assert log == [("levels_begin", 3), ("hook",), ("wait",)]
    del log[:]
    assert _plan_summary(rp.plan(current_sampling_level=2), rp)[0] == _plan_summary(ref.plan(current_sampling_level=2), ref)[0]
    assert log[0] == ("levels_begin", 1)
    rp.close()
    del log[:]
    rp2, _ = build_planner_from_plan_golden(name, PerLevel)
    assert _plan_summary(rp2.plan(), rp2) == want
    assert log == [("plan", 120), ("plan", 540), ("plan", 2754)]
    # the default policy: the first level alone; it fails -> the REMAINING levels in one call; the next cycles go over whole
    del log[:]
    rp3, _ = build_planner_from_plan_golden(name, Logged)
    assert rp3.level_policy == "adaptive"
    packed1 = []
    inner = rp3._gpu_ctx().plan_packed
    rp3._gpu_ctx().plan_packed = lambda *a: (packed1.append(len(a[2]) * len(a[4]) * len(a[5])), inner(*a))[1]
    assert _plan_summary(rp3.plan(), rp3) == want
    # (the oracle-backed double serves its levels call through plan_packed: only the FIRST entry is the planner's own call)
    assert packed1[0] == 120 and log == [("levels_packed", [540, 2754])]
    del log[:], packed1[:]
    assert _plan_summary(rp3.plan(), rp3) == want
    assert packed1[0] == 120 and log == [("levels_packed", [120, 540, 2754])]
    # ... and a scene whose first level delivers never sends more than that level
    del log[:]
    rp4, _ = build_planner_from_plan_golden("plan_arc_hv_obs", Logged)
    for _ in range(3):
        assert rp4.plan() is not None
    assert log == []


def test_closed_planner_plans_again_with_fresh_arena_buffers():
    """the sampling space caches NumPy views of the context's pinned arena (rp_coeffs_arena).  ``close()`` destroys the
    context and with it the arena: the views go with it, and a planner that plans again hands out the NEW context's arrays."""
    from _oracle_ctx import OracleContext

    class Arena(OracleContext):
        n_made = 0

        def __init__(self, device=0):
            super().__init__(device)
            Arena.n_made += 1
            self.tag = Arena.n_made

        def coeffs_arena(self, cap):
            return (np.empty((cap, 6)), np.empty((cap, 6)), np.empty(cap, dtype=np.int32))

    rp, _ = build_planner_from_plan_golden("plan_arc_hv_obs", Arena)

    class Space:   # a batch-view sampling space: what _create_trajectory_bundle asks for the arena on behalf of
        def coeffs_at_level(self, *a):
            return (np.zeros((0, 6)), np.zeros((0, 6)), np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0), np.zeros(0))
    rp.sampling_space = Space()
    rp._create_trajectory_bundle(rp.x_0_cl[0], rp.x_0_cl[1], 1)
    first = rp.sampling_space.__dict__["_arena_owner"]
    assert first is rp._rp_ctx and rp.sampling_space.__dict__["_native_buffers"]["alloc"].__self__ is first
    rp.close()
    assert "_arena_owner" not in rp.sampling_space.__dict__ and "_native_buffers" not in rp.sampling_space.__dict__
    rp._create_trajectory_bundle(rp.x_0_cl[0], rp.x_0_cl[1], 1)
    second = rp.sampling_space.__dict__["_arena_owner"]
    assert second is rp._rp_ctx and second is not first and rp.sampling_space.__dict__["_native_buffers"]["alloc"].__self__ is second
    rp.close()
