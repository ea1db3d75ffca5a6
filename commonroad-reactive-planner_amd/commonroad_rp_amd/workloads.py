"""The five BASELINE.json configurations as concrete ``PlanInputs`` + tables.

Scenario geometry (route centre line, obstacle boxes per time step, initial state) comes from the
small ``tests/golden/scenario_*.npz`` fixtures extracted from the reference's example XML files by
``tests/golden/make_scenarios.py``; grids, horizons and synthetic obstacles follow SURVEY 8d.
Everything here is deterministic (seeded) synthetic input of the configured shape.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Optional

import numpy as np

from ._capi import PlanInputs, make_cost, make_params
from .collision import ObstacleTables
from .coordinate_system import CoordinateSystem

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCENARIO_DIR = os.path.join(_REPO, "tests", "golden")

# commonroad-vehicle-models 3.0.2, vehicle 2 (BMW 320i) as used by all shipped configurations
# (configurations/*.yaml: id_type_vehicle 2; commonroad_rp/utility/config.py:198-222)
VEHICLE2 = dict(length=4.508, width=1.610, wb_rear_axle=1.4227, wheelbase=1.1562 + 1.4227, a_max=11.5,
                v_switch=7.319, delta_max=1.066, v_delta_max=0.4)


@dataclasses.dataclass
class Workload:
    name: str
    description: str
    inputs: PlanInputs
    coordinate_system: CoordinateSystem
    obstacles: ObstacleTables

    @property
    def n_candidates(self) -> int:
        return self.inputs.n_candidates

    def setup(self, ctx):
        ctx.set_coordinate_system(self.coordinate_system)
        ctx.set_obstacles(self.obstacles)


def traj_len_of(T, dt: float) -> np.ndarray:
    """``len(np.arange(0, np.round(T + dt, 5), dt))`` (reactive_planner.py:733,748)."""
    return np.array([len(np.arange(0, np.round(t + dt, 5), dt)) for t in np.atleast_1d(T)], dtype=np.int32)


def velocity_range(v0: float, horizon: float, a_max: float = VEHICLE2["a_max"]):
    """``set_desired_velocity`` sampling interval (reactive_planner.py:332-335)."""
    min_v = max(0, v0 - (0.125 * horizon * a_max))
    max_v = max(min_v + 5.0, v0 + 2)
    return min_v, max_v


def _load_scenario(name: str):
    return dict(np.load(os.path.join(SCENARIO_DIR, f"scenario_{name}.npz")))


def _with_d0(D: np.ndarray, d0: float) -> np.ndarray:
    """``samples_d.union({x_0_lat[0]})`` (sampling.py:226): the current offset is appended unless present."""
    return D if np.any(D == d0) else np.append(D, d0)


def _scenario_workload(name, scen_name, N, T, nL, nD, low_vel_threshold, desired_speed=None, extra_obstacles=0,
                       flags=0, description="", extra_jitter=1.5, extra_lane=4.0, road_boundary=False, extra_sides=(-1.0, 1.0),
                       scenario_obstacles=True) -> Workload:
    sc = _load_scenario(scen_name)
    dt = float(sc["dt"])
    co = CoordinateSystem(sc["centre"], smooth_reference=True)   # (the reference smooths a route by default, utils_coordinate_system.py:88,98-100)
    x, y, th, v0 = (float(v) for v in sc["init"])
    # the planner state sits on the rear axle (state.py:52-55)
    xr, yr = x - VEHICLE2["wb_rear_axle"] * math.cos(th), y - VEHICLE2["wb_rear_axle"] * math.sin(th)
    low = v0 < low_vel_threshold
    x0_lon, x0_lat = co.initial_state(xr, yr, th, v0, wheelbase=VEHICLE2["wheelbase"], low_vel_mode=low)
    vmin, vmax = velocity_range(v0, N * dt)
    L = np.linspace(vmin, vmax, nL)
    D = _with_d0(np.linspace(-3.0, 3.0, nD), x0_lat[0])
    T = np.asarray(T, dtype=np.float64)
    dyn = sc["dyn_obb"] if scenario_obstacles else np.zeros((0,) + tuple(sc["dyn_obb"].shape[1:]))
    if extra_obstacles:
        rng = np.random.default_rng(0)
        n_steps = max(dyn.shape[1], N + 1)
        ext = np.full((dyn.shape[0] + extra_obstacles, n_steps, 5), np.nan)
        ext[:dyn.shape[0], :dyn.shape[1]] = dyn
        s_max = co.ref_pos[-1]
        for j in range(extra_obstacles):   # constant-velocity 4.5 x 2.0 m boxes along the route
            s0, vel, off = rng.uniform(0.0, s_max), rng.uniform(5.0, 15.0), rng.uniform(-extra_jitter, extra_jitter)
            lane = rng.choice(list(extra_sides)) * extra_lane      # neighbouring lanes
            for k in range(n_steps):
                s = s0 + vel * dt * k
                if s >= s_max - 1.0:
                    break
                p = co.convert_to_cartesian_coords(s, off + lane)
                kk = min(int(np.searchsorted(co.ref_pos, s, side="right")) - 1, len(co.ref_pos) - 2)
                ext[dyn.shape[0] + j, k] = (p[0], p[1], co.ref_theta[kk], 2.25, 1.0)
        dyn = ext
    static_obb = sc["static_obb"] if scenario_obstacles else np.zeros((0, 5))
    if road_boundary:   # thin rectangles along the outer border of the lanelet network (collision.road_boundary_obb)
        from .collision import road_boundary_obb, lanelets_from_arrays
        lls = lanelets_from_arrays(sc["ll_ids"], sc["ll_left"], sc["ll_right"], sc["ll_offsets"], sc["ll_flags"])
        static_obb = np.concatenate((np.asarray(static_obb, dtype=float).reshape(-1, 5), road_boundary_obb(lls)))
    obstacles = ObstacleTables(static_obb=static_obb, dyn_obb=dyn, dyn_t0=int(sc["dyn_t0"]))
    params = make_params(dt=dt, N=N, x0_lon=x0_lon, x0_lat=x0_lat, x0_orientation=th, low_vel_mode=low,
                         time_step0=int(sc["init_time_step"]), flags=flags, **VEHICLE2)
    cost = make_cost(desired_speed=v0 if desired_speed is None else desired_speed)
    inp = PlanInputs(params, cost, T, traj_len_of(T, dt), L, D)
    return Workload(name, description, inp, co, obstacles)


def cfg1(level: int = 3, flags: int = 0, road_boundary: bool = False) -> Workload:
    """ZAM_Over-1_1, reference sampling levels (N = 20, t_min 0.2, configurations/ZAM_Over-1_1.yaml)."""
    dt, N, t_min = 0.1, 20, 0.2
    step = int((1 / (level + 1)) / dt)
    T = sorted(set(np.arange(t_min, round(N * dt + dt, 2), step * dt)) - {round(N * dt + dt, 2)})
    n = 2 ** (level + 1) + 1
    return _scenario_workload("cfg1", "ZAM_Over-1_1", N, T, n, n, low_vel_threshold=4.0, flags=flags, road_boundary=road_boundary,
                              description=f"ZAM_Over-1_1, sampling level {level}, N=20" + (", road boundary" if road_boundary else ""))


def ramp(level: int = 1, flags: int = 0, road_boundary: bool = False) -> Workload:
    """ZAM-Ramp-1_1-T-1 at its initial state: standstill on the middle lane (v = 0: standstill branch and low-velocity mode), the
    reference's sampling levels (N = 20, t_min 0.4), three dynamic obstacles on the neighbouring lanes.  The whole scenario in
    closed loop: tests/test_ramp_loop.py."""
    dt, N, t_min = 0.1, 20, 0.4
    step = int((1 / (level + 1)) / dt)
    T = sorted(set(np.arange(t_min, round(N * dt + dt, 2), step * dt)) - {round(N * dt + dt, 2)})
    n = 2 ** (level + 1) + 1
    return _scenario_workload("ramp", "ZAM-Ramp-1_1-T-1", N, T, n, n, low_vel_threshold=4.0, flags=flags, road_boundary=road_boundary,
                              desired_speed=8.0, description=f"ZAM-Ramp-1_1-T-1 at standstill, sampling level {level}, N=20")


def cfg2(flags: int = 0, road_boundary: bool = False) -> Workload:
    """ZAM_Tjunction-1_42_T-1, 15 x 15 x 31 grid, N = 30, 5 dynamic obstacles."""
    dt, N = 0.1, 30
    T = [dt * (16 + k) for k in range(15)]
    return _scenario_workload("cfg2", "ZAM_Tjunction-1_42_T-1", N, T, 31, 15, low_vel_threshold=2.0, flags=flags,
                              road_boundary=road_boundary,
                              description="ZAM_Tjunction-1_42_T-1, 15(d)x15(T)x31(v) grid, N=30, 5 dynamic obstacles"
                                          + (", road boundary (85 rectangles)" if road_boundary else ""))


def cfg3(flags: int = 0, road_boundary: bool = False) -> Workload:
    """DEU_Test-1_1_T-1, 31 x 31 x 63 grid, N = 60, 1 static + 1 dynamic + 49 synthetic obstacles."""
    dt, N = 0.1, 60
    T = [dt * (30 + k) for k in range(31)]
    # synthetic boxes 5 m either side of the route (+- 1.5 m): at 4 m every candidate collides with something
    # within the 6 s horizon and no replanning step has a winner; at 5 m ~1.5 % of the feasible candidates survive
    return _scenario_workload("cfg3", "DEU_Test-1_1_T-1", N, T, 63, 31, low_vel_threshold=4.0, extra_obstacles=49,
                              flags=flags, extra_lane=5.0, road_boundary=road_boundary,
                              description="DEU_Test-1_1_T-1, 31x31x63 grid, N=60, 51 obstacles (49 synthetic, seed 0)")


def cfg3f(flags: int = 0, road_boundary: bool = False) -> Workload:
    """cfg3's grid in mostly-free traffic: 51 synthetic obstacles on ONE side of the route, 5 m out (the opposite lane), and
    without the scenario's own two obstacles -- those, not the synthetic ones, are why 95 % of cfg3's candidates collide (a
    static obstacle in the ego lane that every candidate driving far enough runs into).  Here 20 % of the feasible candidates
    collide and the cheapest one is free: the collision query cannot bail out early on most candidates (the eager kernel
    walks all 61 poses of four in five), and the cost-ordered stage has what it is for."""
    dt, N = 0.1, 60
    T = [dt * (30 + k) for k in range(31)]
    return _scenario_workload("cfg3f", "DEU_Test-1_1_T-1", N, T, 63, 31, low_vel_threshold=4.0, extra_obstacles=51,
                              flags=flags, extra_lane=5.0, extra_sides=(1.0,), road_boundary=road_boundary, scenario_obstacles=False,
                              description="DEU_Test-1_1_T-1 route, 31x31x63 grid, N=60, 51 synthetic obstacles on the opposite lane "
                                          "(seed 0; 20 % of the feasible candidates collide)")


def cfg4(flags: int = 0, road_boundary: bool = False) -> Workload:
    """ZAM_Tjunction-1_42_T-1, 63 x 63 x 127 grid, N = 100."""
    dt, N = 0.1, 100
    T = [dt * (38 + k) for k in range(63)]
    return _scenario_workload("cfg4", "ZAM_Tjunction-1_42_T-1", N, T, 127, 63, low_vel_threshold=2.0, flags=flags,
                              road_boundary=road_boundary,
                              description="ZAM_Tjunction-1_42_T-1, 63x63x127 grid, N=100, 5 dynamic obstacles")


def cfg5(flags: int = 0, obstacles: int = 0, scale: int = 1) -> Workload:
    """Synthetic straight lane, 101(d) x 99(T) x 101(v) = 1 009 899 candidates, N = 100."""
    dt, N = 0.1, 100
    ref = np.stack((np.arange(0.0, 401.0, 1.0), np.zeros(401)), axis=1)
    co = CoordinateSystem(ref)
    T = np.array([dt * k for k in range(2, 101)])
    L = np.linspace(12.0, 18.0, 101 * scale)
    D = np.linspace(-3.0, 3.0, 101)
    tables = ObstacleTables()
    if obstacles:
        rng = np.random.default_rng(0)
        dyn = np.full((obstacles, N + 1, 5), np.nan)
        for j in range(obstacles):
            s0, vel, lane = rng.uniform(30.0, 380.0), rng.uniform(5.0, 15.0), rng.choice([-4.5, 4.5])
            for k in range(N + 1):
                dyn[j, k] = (s0 + vel * dt * k, lane + rng.uniform(-0.2, 0.2), 0.0, 2.25, 1.0)
        tables = ObstacleTables(dyn_obb=dyn, dyn_t0=0)
    params = make_params(dt=dt, N=N, x0_lon=[20.0, 15.0, 0.0], x0_lat=[0.0, 0.0, 0.0], x0_orientation=0.0,
                         low_vel_mode=False, flags=flags, **VEHICLE2)
    inp = PlanInputs(params, make_cost(desired_speed=15.0), T, traj_len_of(T, dt), L, _with_d0(D, 0.0))
    return Workload("cfg5", "synthetic straight lane, 101x99x101 grid, N=100", inp, co, tables)


def replicate_for_ranks(w: Workload, world_size: int) -> Workload:
    """Weak-scaling variant: the longitudinal grid is densified ``world_size`` times so that every
    rank evaluates a shard of the original size (BASELINE north star: shard the candidate batch)."""
    if world_size == 1:
        return w
    L = w.inputs.L
    dense = np.linspace(L.min(), L.max(), len(L) * world_size)
    inp = PlanInputs(w.inputs.params, w.inputs.cost, w.inputs.T, w.inputs.traj_len, dense, w.inputs.D)
    return Workload(w.name, w.description + f", L densified x{world_size}", inp, w.coordinate_system, w.obstacles)


def make_planner(w: Workload, backend_factory=None, device: int = 0, draw: bool = False, nL: Optional[int] = None):
    """A ``ReactivePlanner`` set up for a workload: same reference path, obstacles, initial state, cost
    parameters and -- through explicit sample sets at level 1 -- the same (T, L, D) grids, so that
    ``planner.plan()`` evaluates exactly the workload's candidate batch (used for the p50 plan() latency).
    The grid density survives ``set_desired_velocity`` (which the replanning loop calls every cycle and which rebuilds
    the velocity samples from the current speed, reactive_planner.py:332-335): the level-1 sample set is rebuilt with
    the workload's number of samples (``nL`` overrides it) over the new interval."""
    from .config import ReactivePlannerConfiguration
    from .reactive_planner import ReactivePlanner as _Base
    from .state import ReactivePlannerState
    n_lon = int(nL or len(w.inputs.L))

    class ReactivePlanner(_Base):
        def set_v_sampling_parameters(self, v_min, v_max):
            super().set_v_sampling_parameters(v_min, v_max)
            self.sampling_space.samples_v._dict_level_to_sample_set[1] = set(np.linspace(v_min, v_max, n_lon).tolist())
    p = w.inputs.params
    cfg = ReactivePlannerConfiguration.from_dict(dict(
        planning=dict(dt=p.dt, time_steps_computation=p.N, factor=p.factor,
                      low_vel_mode_threshold=1e9 if p.low_vel_mode else 0.0),
        sampling=dict(longitudinal_mode="stopping" if p.lon_mode else "velocity_keeping", num_sampling_levels=2,
                      t_min=float(max(2 * p.dt, min(w.inputs.T)))),
        debug=dict(draw_traj_set=draw, show_plots=draw)))
    rp = ReactivePlanner(cfg, backend_factory=backend_factory, device=device)
    rp.set_reference_path(coordinate_system=w.coordinate_system)
    rp.set_collision_checker(collision_checker=w.obstacles)
    co = w.coordinate_system
    pos = co.convert_to_cartesian_coords(p.x0_lon[0], p.x0_lat[0])
    x0 = ReactivePlannerState(time_step=p.time_step0, position=pos, orientation=p.x0_orientation,
                              velocity=max(p.x0_lon[1], 0.1), steering_angle=0.0, acceleration=0.0, yaw_rate=0.0)
    rp.reset(initial_state_cart=x0, initial_state_curv=(list(p.x0_lon), list(p.x0_lat)),
             collision_checker=rp.collision_checker, coordinate_system=co)
    c = w.inputs.cost
    rp.set_desired_velocity(desired_velocity=None if math.isnan(c.desired_speed) else c.desired_speed,
                            current_speed=p.x0_lon[1])
    sp = rp.sampling_space
    sp.samples_t._dict_level_to_sample_set[1] = set(float(t) for t in w.inputs.T)
    L = w.inputs.L if n_lon == len(w.inputs.L) else np.linspace(w.inputs.L.min(), w.inputs.L.max(), n_lon)
    sp.samples_v._dict_level_to_sample_set[1] = set(float(v) for v in L)
    sp.samples_s._dict_level_to_sample_set[1] = set(float(v) for v in L)
    D = w.inputs.D   # (_with_d0 appended the current offset: the planner's own grid is the one without it, sampling.py:226)
    if len(D) > 1 and D[-1] == p.x0_lat[0] and not np.any(D[:-1] == p.x0_lat[0]):
        D = D[:-1]
    sp.samples_d._dict_level_to_sample_set[1] = set(float(d) for d in D)
    return rp


def make_level_planner(w: Workload, t_min: float, backend_factory=None, device: int = 0, low_vel_mode_threshold: float = 4.0,
                       draw: bool = False):
    """A ``ReactivePlanner`` on a workload's scenario (reference path, obstacles, initial state) with the REFERENCE's OWN sampling
    scheme -- levels 1 .. num_sampling_levels - 1 of ``FixedIntervalSampling``, velocity range from ``set_desired_velocity`` -- and
    the settings of the shipped YAML files (configurations/*.yaml: N = 20, dt 0.1, t_min per scenario): what run_planner.py:28-107
    drives.  The workload's own grids are not used."""
    from .config import ReactivePlannerConfiguration
    from .reactive_planner import ReactivePlanner
    from .state import ReactivePlannerState
    p = w.inputs.params
    cfg = ReactivePlannerConfiguration.from_dict(dict(
        planning=dict(dt=p.dt, time_steps_computation=20, factor=p.factor, low_vel_mode_threshold=low_vel_mode_threshold),
        sampling=dict(longitudinal_mode="velocity_keeping", t_min=t_min),
        debug=dict(draw_traj_set=draw, show_plots=draw)))
    rp = ReactivePlanner(cfg, backend_factory=backend_factory, device=device)
    co = w.coordinate_system
    rp.set_reference_path(coordinate_system=co)
    rp.set_collision_checker(collision_checker=w.obstacles)
    pos = co.convert_to_cartesian_coords(p.x0_lon[0], p.x0_lat[0])
    v0 = float(w.inputs.cost.desired_speed)
    x0 = ReactivePlannerState(time_step=p.time_step0, position=pos, orientation=p.x0_orientation, velocity=v0, steering_angle=0.0,
                              acceleration=0.0, yaw_rate=0.0)
    rp.reset(initial_state_cart=x0, initial_state_curv=(list(p.x0_lon), list(p.x0_lat)), collision_checker=rp.collision_checker,
             coordinate_system=co)
    rp.set_desired_velocity(desired_velocity=v0, current_speed=v0)
    return rp


def replan_sequence(w: Workload, n_states: int = 32, device: int = 0, backend_factory=None, nL: Optional[int] = None):
    """The inputs of ``n_states`` consecutive replanning cycles of a workload ("replan sequence = closed
    loop over the scenario (cfg1-4) or a fixed list of initial states drawn with seed 1 (cfg5)"), as a list of
    ``PlanInputs``: what the benchmark steps through instead of repeating one input.

    Scenario workloads: the planner drives the scenario in closed loop, replanning every time step
    (``harness.run_closed_loop``, the loop of run_planner.py:61-107): the winner's next state becomes the initial state,
    the velocity grid is re-centred on the current speed (reactive_planner.py:332-335) at the workload's density, the current
    offset d0 joins the lateral grid (sampling.py:226), time advances against the obstacles' trajectories.  The inputs
    of every ``rp_plan`` call are recorded.  The loop ends early when no candidate survives (end of the route, blocked
    road); the recorded states are then what the sequence holds.  ``nL``: longitudinal samples per cycle (default: the
    workload's; weak-scaling runs densify it).
    Needs the device: the loop runs on the product path (no CPU fallback)."""
    from ._capi import RpContext, copy_params
    from .harness import run_closed_loop
    inp0 = w.inputs
    nL = int(nL or len(inp0.L))
    if w.name == "cfg5":
        rng = np.random.default_rng(1)
        seq = []
        for _ in range(n_states):
            p = copy_params(inp0.params)
            v0 = 15.0 + rng.uniform(-1.0, 1.0)
            p.x0_lon[:] = [20.0 + rng.uniform(-5.0, 5.0), v0, rng.uniform(-0.3, 0.3)]
            p.x0_lat[:] = [rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), 0.0]
            D = _with_d0(np.linspace(-3.0, 3.0, len(inp0.D) - (0 if np.any(np.linspace(-3.0, 3.0, len(inp0.D)) == 0.0) else 1)), p.x0_lat[0])
            seq.append(PlanInputs(p, inp0.cost, inp0.T, inp0.traj_len, np.linspace(v0 - 3.0, v0 + 3.0, nL), D))
        return seq
    recorded = []
    factory = backend_factory or RpContext

    class Recording(factory):
        # (the planner's one-call cycle bypasses plan(): off, so that every level's inputs pass through here)
        plan_packed = plan_levels_packed = plan_levels_begin = None

        def plan(self, inp, *a, **k):
            recorded.append(PlanInputs(copy_params(inp.params), inp.cost, inp.T.copy(), inp.traj_len.copy(), inp.L.copy(),
                                       inp.D.copy()))
            return super().plan(inp, *a, **k)

    rp = make_planner(w, backend_factory=Recording, device=device, nL=nL)
    c = inp0.cost
    run_closed_loop(rp, max_steps=n_states, replanning_frequency=1,
                    desired_velocity=None if math.isnan(c.desired_speed) else float(c.desired_speed))
    rp.close()
    if not recorded:
        raise RuntimeError(f"{w.name}: the closed loop produced no replanning step")
    return recorded[:n_states]


WORKLOADS = {"cfg1": cfg1, "cfg2": cfg2, "cfg3": cfg3, "cfg3f": cfg3f, "cfg4": cfg4, "cfg5": cfg5, "ramp": ramp}
