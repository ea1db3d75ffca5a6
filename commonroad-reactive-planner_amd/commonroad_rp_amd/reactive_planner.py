"""The planner behind the reference's ``ReactivePlanner`` interface, with the hot path on the GPU.

Two pieces:

``GpuBackendMixin``
    Overrides exactly the two methods ``plan()`` calls per sampling level
    (reference: commonroad_rp/reactive_planner.py:620 ``_create_trajectory_bundle`` and :624
    ``_get_optimal_trajectory``) and routes them through ``librp_amd.so``.  It reads only attributes
    the reference planner already has (``config, dt, N, horizon, vehicle_params, x_0, _co,
    sampling_space, cost_function, _low_vel_mode, _draw_traj_set, ...``), so it can be mixed into the
    reference class itself: ``class Planner(GpuBackendMixin, commonroad_rp.reactive_planner.ReactivePlanner)``
    (exercised against the real reference class in tests/test_planner_dropin.py).

``ReactivePlanner``
    A stand-alone planner with the reference's public surface (``reset, set_reference_path,
    set_collision_checker, set_desired_velocity, set_desired_lon_position, set_cost_function,
    set_sampling_space, set_*_sampling_parameters, plan, record_state_and_input, ...``) that does not
    need commonroad-io / commonroad-drivability-checker.  Everything outside the two overridden methods
    follows the reference's control flow (levels loop :616-636, standstill :638-653, output packing
    :514-568).

There is no CPU fallback: without the HIP library ``plan()`` raises ``RpLibraryMissing``.
"""
from __future__ import annotations

import logging
import math
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi
from ._capi import (CHECK_BITS, COST_DEFAULT, COST_EXTERNAL, COST_FAILSAFE, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL,
                    LON_STOPPING, LON_VELOCITY_KEEPING, PlanInputs, PlanOutput, RpContext, make_cost, make_params)
from .collision import ObstacleTables
from .config import ReactivePlannerConfiguration
from .coordinate_system import CoordinateSystem, interpolate_angle
from .cost_function import CostFunction, DefaultCostFunction, DefaultCostFunctionFailSafe
from .polynomial_trajectory import QuarticTrajectory, QuinticTrajectory
from .sampling import (FixedIntervalSampling, PositionSampling, SamplingSpace, TimeSampling, VelocitySampling,
                       sampling_space_factory)
from .state import CustomState, LazyStateList, ReactivePlannerState, Trajectory
from .trajectories import (CartesianSample, CurviLinearSample, FeasibilityStatus, TrajectorySample, bind_states,
                           label_from_status)

logger = logging.getLogger("RP_LOGGER")
_ZERO3 = (0.0, 0.0, 0.0)


class _CostView:
    def __init__(self, cost):
        self.cost = cost


class GpuTrajectoryBundle:
    """What ``_create_trajectory_bundle`` hands to ``_get_optimal_trajectory`` on the GPU path: the
    sample grids (or, for foreign sampling spaces, explicit polynomials) instead of one Python object
    per candidate.  ``trajectories`` materialises the objects on demand for code that wants them."""

    def __init__(self, planner, level: int, x_0_lon, x_0_lat, grids=None, samples: Optional[List[TrajectorySample]] = None,
                 coeffs=None):
        self._planner = planner
        self.level = level
        self.x_0_lon, self.x_0_lat = list(x_0_lon), list(x_0_lat)
        self.grids = grids            # (T, traj_len, L, D) or None
        self._samples = samples       # foreign sampling space: list of TrajectorySample
        self.coeffs = coeffs          # sampling space with a batch view but no product grid (CorridorSampling):
                                      # (lon_coeffs [C, 6], lat_coeffs [C, 6], lon_T [C], traj_len [C], lon_end [C], lat_end [C])
        self._is_sorted = False
        self.costs: Optional[np.ndarray] = None   # per-candidate costs after _get_optimal_trajectory (when they were fetched)
        self._range_fn = None         # () -> (min, max, n) of the costs on the device: min_costs() / max_costs() without a fetch
        self._range = None

    @property
    def n_candidates(self) -> int:
        if self.grids is not None:
            T, _, L, D = self.grids
            return len(T) * len(L) * len(D)
        if self.coeffs is not None:
            return len(self.coeffs[2])
        return len(self._samples)

    @property
    def trajectories(self) -> List[TrajectorySample]:
        if self._samples is None and self.coeffs is not None:
            self._samples = [self._planner._gpu_candidate_sample(self, i) for i in range(len(self.coeffs[2]))]
        if self._samples is None:
            p = self._planner
            self._samples = p.sampling_space.generate_trajectories_at_level(
                self.level, self.x_0_lon, self.x_0_lat, p.config.sampling.longitudinal_mode, p._low_vel_mode)
            if p.config.sampling.longitudinal_mode == "stopping":
                self._samples = [t for t in self._samples if t.trajectory_long.x_0[0] < t.trajectory_long.x_d[0]]
        return self._samples

    @property
    def empty(self) -> bool:
        return self.n_candidates == 0

    def _cost_range(self):
        """(min, max) over the candidates that have a cost, or None: from the fetched costs if there are any, else from the
        device (``rp_cost_range``, asked once, while the context still holds this bundle's plan)."""
        if self._range is None:
            c = self.costs
            if c is not None:
                self._range = (float(np.nanmin(c)), float(np.nanmax(c))) if np.any(~np.isnan(c)) else ()
            elif self._range_fn is not None:
                lo, hi, n = self._range_fn()
                self._range = (lo, hi) if n > 0 else ()
            else:
                self._range = ()
        return self._range or None

    def min_costs(self):
        r = self._cost_range()
        return _CostView(np.float64(r[0])) if r else None

    def max_costs(self):
        r = self._cost_range()
        return _CostView(np.float64(r[1])) if r else None


def _tables_from_scenario(scenario, road_boundary_obstacle=None, continuous: bool = False) -> ObstacleTables:
    """What ``set_collision_checker(scenario=...)`` puts into the checker (reactive_planner.py:234-251): the
    obstacles -- dynamic ones as OBB sums of consecutive steps when ``planning.continuous_collision_check`` is on
    (:238-245) -- and the road boundary: the caller's ((n, 6) triangles), or generated from the lanelet network when
    none is given, as the reference calls ``create_road_boundary_obstacle(scenario)`` (:246-248).
    The reference's ``road_boundary_obstacle`` is an opaque ``pycrcc.CollisionObject``; its content cannot be read
    from here, so anything but triangles is refused instead of being dropped silently."""
    tri = None
    if road_boundary_obstacle is not None:
        if not isinstance(road_boundary_obstacle, (np.ndarray, list, tuple)):
            raise TypeError(
                "GPU backend: road_boundary_obstacle must be an (n, 6) array of triangles (or None: the boundary is "
                f"generated from the lanelet network); an opaque {type(road_boundary_obstacle).__name__} cannot be "
                "turned into obstacle tables (INTEGRATION.md, Obstacles)")
        tri = road_boundary_obstacle
    net = getattr(scenario, "lanelet_network", None)
    generate = tri is None and net is not None and len(getattr(net, "lanelets", [])) > 0
    return ObstacleTables.from_scenario(scenario, tri, road_boundary=generate, continuous=continuous)


class _WinnerSample(TrajectorySample):
    """The winner as a ``TrajectorySample`` whose two polynomial objects are built when somebody reads them: ``plan()`` itself
    only reads the states and the cost, and two polynomial objects per replanning cycle were a tenth of its Python time."""

    def __init__(self, horizon, dt, make_polys):
        self._make_polys = make_polys
        self._polys = None
        self._samples = None       # (CartesianSample, CurviLinearSample) once somebody reads them, or what a caller assigned
        self._state_block = None   # the [14, N + 1] block the two samples are views of (bind_block)
        super().__init__(horizon, dt, None, None)

    def _get(self, k):
        if self._polys is None:
            self._polys = self._make_polys()
        return self._polys[k]

    trajectory_long = property(lambda self: self._get(0), lambda self, v: None)
    trajectory_lat = property(lambda self: self._get(1), lambda self, v: None)

    # the two sample containers are built from the state block when somebody reads them: plan() packs its output straight from
    # the block (_compute_trajectory_pair), and two container objects per replanning cycle were ~4 us of its Python time
    def bind_block(self, block, cost):
        self._state_block, self._samples = block, None
        self.feasibility_label = FeasibilityStatus.FEASIBLE
        self._cost = cost

    def _sample(self, k):
        if self._samples is None:
            if self._state_block is None:
                return None
            blk = self._state_block
            n = blk.shape[1]
            x, y, theta, v, a, kappa, kappa_dot, s, d, theta_cl, s_dot, s_ddot, d_dot, d_ddot = blk
            self._samples = [CartesianSample(x, y, theta, v, a, kappa, kappa_dot, current_time_step=n),
                             CurviLinearSample(s, d, theta_cl, dd=d_dot, ddd=d_ddot, ss=s_dot, sss=s_ddot, current_time_step=n)]
        return self._samples[k]

    def _set_sample(self, k, value):
        if value is None and self._samples is None:
            return                       # (TrajectorySample.__init__ assigns None)
        if self._samples is None:
            self._samples = [self._sample(0), self._sample(1)] if self._state_block is not None else [None, None]
        self._samples[k] = value
        self._state_block = None         # a caller replaced a container: the block no longer describes the sample

    cartesian = property(lambda self: self._sample(0), lambda self, v: self._set_sample(0, v))
    curvilinear = property(lambda self: self._sample(1), lambda self, v: self._set_sample(1, v))


_REASON_ITEMS = tuple(enumerate(_capi.REASON_NAMES[1:6], start=1))   # (reason code, name in infeasible_reason_dict)
_LON_ROWS, _LAT_ROWS = [7, 10, 11], [8, 12, 13]   # rows of a state block (include/rp_amd.h: RP_S, RP_S_DOT, RP_S_DDOT | RP_D, ...)


#: Candidates per sampling level from which a plan is sharded over the GPUs of the planner's process group (SURVEY 8e:
#: "only when the sample grid exceeds one GPU's saturation point").  From the one-GPU step times of this build (MI355X,
#: production mode): 7 440 candidates 31 us, 62 496: 163 us, 512 064: 521 us -- beyond the single-launch
#: regime a step grows by ~2.4 us per 1 000 candidates, and the winner exchange costs 6-9 us (shared-memory mailbox of one node; a
#: device collective ~40 us).  Halving a batch pays once C / 2 x 2.4 us / 1 000 exceeds the exchange with a margin of two:
#: C > 15 000.  Below that every shard pays the whole launch chain for nothing (2 ranks on a 7 440-candidate grid: 35 -> 43 us).
SHARD_MIN_CANDIDATES = 16384


class GpuBackendMixin:
    """Routes the per-level hot path of ``plan()`` through the HIP library."""

    #: factory ``device -> context`` with the interface of ``_capi.RpContext``
    backend_factory = RpContext
    gpu_device = 0
    # multi-GPU: the process group (``torch.distributed`` or an object with its interface) this planner's ranks share, and from
    # how many candidates on a level is sharded over it (set_process_group)
    shard_dist = None
    shard_device = None
    shard_transport = "auto"
    shard_min_candidates = SHARD_MIN_CANDIDATES
    shard_single_rank = False   # True: a group of ONE rank still takes the sharded path (the whole exchange with itself: tests of the
                                # RCCL transport on a one-GPU box)
    #: called once per ``plan()`` of the stand-alone planner, between the launch of the cycle's kernels and the wait for their
    #: result (``rp_plan_begin`` / ``rp_plan_wait``): the caller's own work overlaps the device (harness.run_closed_loop)
    on_device_launched = None

    def set_process_group(self, dist, device=None, transport: str = "auto", min_candidates: Optional[int] = None):
        """One planner per rank, the same inputs on every rank (the reference's only parallel backend also lives inside
        ``_get_optimal_trajectory``: the ``multiprocessing`` fan-out over ``config.debug.num_workers`` workers,
        reactive_planner.py:1084-1111).  With a group of more than one rank, a sampling level of at least ``min_candidates``
        candidates (default ``SHARD_MIN_CANDIDATES``) is evaluated in contiguous index ranges, one per rank
        (``distributed.shard_range``), and the ranks exchange their winners (``distributed.exchange_winner``): every rank
        returns the same optimal trajectory, counters and costs as an unsharded call.  Smaller levels, draw mode, plug-in cost
        functions and sampling spaces without a product grid run whole on every rank.  ``dist=None`` switches sharding off.
        Collective: call it on every rank with the same arguments."""
        self.shard_dist = dist
        if dist is not None and device is None:
            import torch
            device = torch.device("cuda", self.gpu_device) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
        self.shard_device = device
        self.shard_transport = transport
        if min_candidates is not None:
            self.shard_min_candidates = int(min_candidates)
        if dist is not None and dist.get_world_size() > 1:
            # (ranks of one node share its CPUs: distributed.wait_mode_for_group -- yield between polls when there are fewer than two per rank)
            from .distributed import wait_mode_for_group, local_world_size
            ctx = self._gpu_ctx() if self._co is not None else None
            if ctx is not None and hasattr(ctx, "set_wait_mode"):
                ctx.set_wait_mode(wait_mode_for_group(local_world_size(dist)))

    def _shard_world(self, n_candidates: int, grid_plan: bool, external: bool) -> int:
        """Ranks this level is sharded over (1: not sharded).  The same decision on every rank: it depends on the inputs only."""
        dist = self.shard_dist
        if dist is None or not grid_plan or external or self._draw_traj_set:
            return 1
        world = dist.get_world_size()
        if world == 1 and self.shard_single_rank and n_candidates >= self.shard_min_candidates:
            return -1   # (a world of one, sharded all the same: see shard_single_rank)
        return world if (world > 1 and n_candidates >= self.shard_min_candidates) else 1

    # ---- context and tables -----------------------------------------------------------------------
    def _gpu_ctx(self):
        ctx = getattr(self, "_rp_ctx", None)
        if ctx is None:
            ctx = self._rp_ctx = self.backend_factory(self.gpu_device)
            self._rp_ref_id = None
            self._rp_obs_id = None
        co = self._co
        if self._rp_ref_id is not co:
            ctx.set_coordinate_system(co)
            self._rp_ref_id = co
        tables = getattr(self, "_obstacle_tables", None)
        if tables is None and getattr(self, "_rp_tables_refused", False):
            raise TypeError("GPU backend: the planner holds a collision checker whose content could not be read "
                            "(set_collision_checker refused it); refusing to plan without a collision check")
        if self._rp_obs_id is not tables or tables is None:
            if tables is None:
                tables = self._obstacle_tables = ObstacleTables()
            ctx.set_obstacles(tables)
            self._rp_obs_id = tables
        return ctx

    def set_collision_checker(self, scenario=None, collision_checker=None, road_boundary_obstacle=None):
        """When mixed into the reference class: keep its pycrcc checker (other code may use it) and, in
        addition, keep the obstacle tables the GPU check needs in step with it:
          * ``scenario=...``: tables extracted from the same scenario (what the reference feeds to pycrcc, :234-251);
          * ``collision_checker=`` an ``ObstacleTables``: taken as they are;
          * ``collision_checker=`` the very checker object the current tables were extracted for -- what ``reset()``
            hands back every replanning cycle (reactive_planner.py:190-195, run_planner.py:99-107) --: tables kept;
          * any other (opaque pycrcc) checker: its content cannot be read, and planning against stale or empty
            tables would silently skip the collision check -- refused with a ``TypeError``.  Attach the matching
            tables as ``checker.obstacle_tables`` or call ``set_obstacle_tables`` to use such a checker."""
        if road_boundary_obstacle is not None and not isinstance(road_boundary_obstacle, (np.ndarray, list, tuple)):
            _tables_from_scenario(None, road_boundary_obstacle)   # raises: an opaque pycrcc object cannot become tables
        parent = getattr(super(), "set_collision_checker", None)
        if parent is not None:
            parent(scenario=scenario, collision_checker=collision_checker, road_boundary_obstacle=road_boundary_obstacle)
        if collision_checker is None:
            if scenario is not None:
                cont = bool(getattr(getattr(self.config, "planning", None), "continuous_collision_check", False))
                self.set_obstacle_tables(_tables_from_scenario(scenario, road_boundary_obstacle, continuous=cont))
                self._rp_tables_checker = getattr(self, "_cc", None)
            return
        if isinstance(collision_checker, ObstacleTables):
            self.set_obstacle_tables(collision_checker)
        elif getattr(self, "_obstacle_tables", None) is not None and \
                collision_checker is getattr(self, "_rp_tables_checker", None):
            pass   # the checker the tables mirror, handed back by reset(): nothing changed
        elif isinstance(getattr(collision_checker, "obstacle_tables", None), ObstacleTables):
            self.set_obstacle_tables(collision_checker.obstacle_tables)
        else:
            self._obstacle_tables = None
            self._rp_tables_checker = None
            self._rp_tables_refused = True
            raise TypeError(
                f"GPU backend: cannot read the content of {type(collision_checker).__name__}; pass the scenario, an "
                "ObstacleTables object, or attach the matching tables as `collision_checker.obstacle_tables` "
                "(INTEGRATION.md, Obstacles)")
        self._rp_tables_checker = collision_checker

    def set_obstacle_tables(self, tables: Optional[ObstacleTables]):
        """Obstacle content for the GPU collision check (what the reference keeps inside the opaque
        pycrcc checker, reactive_planner.py:234-251)."""
        self._obstacle_tables = tables if tables is not None else ObstacleTables()
        self._rp_tables_refused = False
        self._rp_tables_checker = getattr(self, "_cc", None)   # (drop-in mode: these tables stand for the planner's checker)

    def close(self):
        ctx = getattr(self, "_rp_ctx", None)
        if ctx is not None:
            ctx.close()
            self._rp_ctx = None
        # a sampling space that wrote its candidates into pinned arrays of the context just destroyed (rp_coeffs_arena) must not
        # keep views of them: the next context hands out its own
        sp = getattr(self, "sampling_space", None)
        if sp is not None:
            for key in ("_native_buffers", "_arena_owner", "_last_groups"):
                sp.__dict__.pop(key, None)

    # ---- per-call inputs ---------------------------------------------------------------------------
    def _gpu_params_struct(self):
        """The ``rp_params`` struct of this call with everything that does NOT change from one replanning cycle to the next filled
        in.  The struct is kept between calls (filling all 25 fields through ctypes costs more than the rest of the call's set-up);
        the cycle's own fields are written by ``_gpu_params`` -- or by the call itself (``RpContext.plan_packed_fast``)."""
        cfg, vp = self.config, self.vehicle_params
        key = (id(cfg), id(vp), self.dt, self.N, cfg.planning.factor, cfg.sampling.longitudinal_mode,
               tuple(cfg.planning.constraints_to_check), vp.wheelbase, vp.wb_rear_axle, vp.length, vp.width, vp.a_max,
               vp.v_switch, vp.delta_max, vp.v_delta_max)
        cached = getattr(self, "_rp_params_cache", None)
        if cached is None or cached[0] != key:
            mask = 0
            for c in cfg.planning.constraints_to_check:
                mask |= CHECK_BITS[c]
            p = make_params(dt=self.dt, N=self.N, factor=cfg.planning.factor, time_step0=0, low_vel_mode=False,
                            lon_mode=LON_STOPPING if cfg.sampling.longitudinal_mode == "stopping" else LON_VELOCITY_KEEPING,
                            constraint_mask=mask, flags=0, x0_lon=(0.0, 0.0, 0.0), x0_lat=(0.0, 0.0, 0.0), x0_orientation=0.0,
                            wheelbase=vp.wheelbase, wb_rear_axle=vp.wb_rear_axle, length=vp.length, width=vp.width,
                            a_max=vp.a_max, v_switch=vp.v_switch, delta_max=vp.delta_max, v_delta_max=vp.v_delta_max)
            cached = self._rp_params_cache = (key, p, [_capi.copy_params(p), _capi.copy_params(p)], [0])
        # two structs used in turn (a copy per call costs ~1.5 us): the inputs of the previous plan stay intact while this one
        # runs; whoever keeps inputs longer copies them (workloads.replan_sequence does)
        cached[3][0] ^= 1
        return cached[2][cached[3][0]]

    def _gpu_params(self, x_0_lon, x_0_lat, flags: int):
        """The ``rp_params`` of this call: ``_gpu_params_struct`` with the cycle's fields written (reactive_planner.py:586-594)."""
        p = self._gpu_params_struct()
        p.time_step0 = int(self.x_0.time_step)
        p.low_vel_mode = 1 if self._low_vel_mode else 0
        p.flags = flags
        p.x0_lon[0], p.x0_lon[1], p.x0_lon[2] = x_0_lon
        p.x0_lat[0], p.x0_lat[1], p.x0_lat[2] = x_0_lat
        p.x0_orientation = self.x_0.orientation
        return p

    def _gpu_cost(self):
        """Cost parameters for the fused path, or ``None`` for a plug-in cost function."""
        cf = self.cost_function
        if type(cf) is DefaultCostFunction or type(cf).__name__ == "DefaultCostFunction" and \
                type(cf).evaluate.__qualname__ == "DefaultCostFunction.evaluate":
            key = (cf.w_a, cf.desired_speed, cf.desired_d, cf.desired_s)   # (the struct is kept while the weights stay)
            hit = getattr(self, "_rp_cost_cache", None)
            if hit is None or hit[0] != key:
                hit = self._rp_cost_cache = (key, make_cost(COST_DEFAULT, w_a=cf.w_a, desired_speed=cf.desired_speed,
                                                            desired_d=cf.desired_d, desired_s=cf.desired_s))
            return hit[1]
        if type(cf) is DefaultCostFunctionFailSafe or type(cf).__name__ == "DefaultCostFunctionFailSafe" and \
                type(cf).evaluate.__qualname__ == "DefaultCostFunctionFailSafe.evaluate":
            return make_cost(COST_FAILSAFE)
        return None

    # ---- the two overridden methods ----------------------------------------------------------------
    def _create_trajectory_bundle(self, x_0_lon, x_0_lat, samp_level: int) -> GpuTrajectoryBundle:
        logger.info("===== Sampling trajectories ... =====")
        logger.info("Sampling density %d of %d", samp_level + 1, self.sampling_level)   # (lazy formatting: hot path)
        sp = self.sampling_space
        if hasattr(sp, "grids_at_level"):
            grids = sp.grids_at_level(samp_level, x_0_lon, x_0_lat, self.config.sampling.longitudinal_mode)
            bundle = GpuTrajectoryBundle(self, samp_level, x_0_lon, x_0_lat, grids=grids)
        elif hasattr(sp, "coeffs_at_level"):   # data-dependent candidates (CorridorSampling), batch view: coefficient arrays
            ctx = self._gpu_ctx()
            if sp.__dict__.get("_arena_owner") is not ctx:
                # the library writes the candidates straight into pinned arrays of this planner's context (rp_coeffs_arena): rp_plan_coeffs
                # then skips its copy into the staging buffer -- a quarter of a millisecond at 25 000 candidates.  The arrays belong
                # to the CONTEXT (rp_destroy frees them): a planner that was closed and plans again gets the new context's
                sp.__dict__.pop("_last_groups", None)
                if hasattr(ctx, "coeffs_arena"):
                    sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
                else:
                    sp.__dict__.pop("_native_buffers", None)
                sp.__dict__["_arena_owner"] = ctx
            coeffs = sp.coeffs_at_level(samp_level, x_0_lon, x_0_lat, self.config.sampling.longitudinal_mode, self._low_vel_mode)
            bundle = GpuTrajectoryBundle(self, samp_level, x_0_lon, x_0_lat, coeffs=coeffs)
            lg = sp.__dict__.get("_last_groups")   # (groups of exactly these arrays, if the batch view reported any)
            bundle.coeff_groups = lg[1] if (lg is not None and lg[0] is coeffs[2]) else None
        else:   # foreign sampling space: its own objects, polynomials handed to the device explicitly
            samples = sp.generate_trajectories_at_level(samp_level, x_0_lon, x_0_lat,
                                                        self.config.sampling.longitudinal_mode, self._low_vel_mode)
            if self.config.sampling.longitudinal_mode == "stopping":
                samples = [t for t in samples if t.trajectory_long.x_0[0] < t.trajectory_long.x_d[0]]
            bundle = GpuTrajectoryBundle(self, samp_level, x_0_lon, x_0_lat, samples=samples)
        logger.info("Number of trajectory samples: %d", bundle.n_candidates)
        return bundle

    def _get_optimal_trajectory(self, bundle: GpuTrajectoryBundle) -> Optional[TrajectorySample]:
        logger.info("===== Checking trajectories ... =====")
        self._reset_statistics()
        ctx = self._gpu_ctx()
        t0 = time.time()
        cost = self._gpu_cost()
        external = cost is None
        flags = 0
        if self._draw_traj_set:
            flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
        if external:
            flags |= FLAG_MATERIALIZE_ALL
            cost = make_cost(COST_EXTERNAL)
        params = self._gpu_params(bundle.x_0_lon, bundle.x_0_lat, flags)

        sharded = False
        if bundle.grids is not None:
            T, traj_len, L, D = bundle.grids
            C = len(T) * len(L) * len(D)
            if C == 0:
                self._infeasible_count_kinematics = 0
                return None
            world = self._shard_world(C, True, external)
            if world == -1:
                world = 1
                force_exchange = True
            else:
                force_exchange = False
            if world > 1 or force_exchange:   # this rank's contiguous range of the reference list index, then the winner exchange
                from .distributed import exchange_winner, shard_range
                lo, hi = shard_range(C, self.shard_dist.get_rank(), world)
                out = ctx.plan(PlanInputs(params, cost, T, traj_len, L, D), lo, hi)
                out = exchange_winner(ctx, out, self.shard_dist, self.shard_device, self.shard_transport)
                sharded = True
            else:
                out = ctx.plan(PlanInputs(params, cost, T, traj_len, L, D))
        elif bundle.coeffs is not None:
            lon, lat, lon_T, tl = bundle.coeffs[:4]
            if len(lon_T) == 0:
                self._infeasible_count_kinematics = 0
                return None
            groups = getattr(bundle, "coeff_groups", None)
            out = ctx.plan_coeffs(params, cost, lon, lat, lon_T, tl, groups=groups) if groups else ctx.plan_coeffs(params, cost, lon, lat, lon_T, tl)
        else:
            samples = bundle.trajectories
            if not samples:
                self._infeasible_count_kinematics = 0
                return None
            lon = np.array([t.trajectory_long.coeffs for t in samples])
            lat = np.array([t.trajectory_lat.coeffs for t in samples])
            lon_T = np.array([t.trajectory_long.delta_tau for t in samples], dtype=float)
            tl = np.array([len(np.arange(0, np.round(tt + self.dt, 5), self.dt)) for tt in lon_T], dtype=np.int32)
            out = ctx.plan_coeffs(params, cost, lon, lat, lon_T, tl)
        logger.info("Kinematic checks took:  \t%.7fs", time.time() - t0)

        if external:   # plug-in CostFunction: states from the device, evaluate() in Python, argmin on the device
            out = self._gpu_external_costs(ctx, bundle, out)

        # statistics the reference keeps (reactive_planner.py:127-145, 1119)
        self._infeasible_count_kinematics = out.n_infeasible_kinematics
        self._infeasible_count_collision = out.n_collision_before_best
        for i, name in enumerate(_capi.REASON_NAMES[1:6], start=1):
            if name in self._infeasible_reason_dict:
                self._infeasible_reason_dict[name] = int(out.reason_counts[i])
        status, costs = ctx.fetch_status() if (self._draw_traj_set or (bundle.grids is None and bundle.coeffs is None)) else (None, None)
        bundle.costs = costs
        # plan()'s standstill branch reads bundle.min_costs() / max_costs() (reactive_planner.py:650-651): two numbers, worked out
        # on the device when somebody asks (the reference's bundle then holds the kinematically feasible samples, :1128)
        bundle._range_fn, bundle._range = getattr(ctx, "cost_range", None), None
        if sharded and bundle._range_fn is not None:   # (every rank takes the standstill branch or none: same inputs, same winner)
            bundle._range_fn = self._gpu_sharded_cost_range(ctx)
        if self._draw_traj_set:
            self.stored_trajectories = self._gpu_stored_trajectories(ctx, bundle, status, costs)
        if out.best_index < 0:
            return None
        if self.config.planning.continuous_collision_check:
            # second test of _check_collisions (reactive_planner.py:1049-1058), applied to the first sample of the
            # sorted list that passed the per-pose test.  A hit there `break`s out of the loop over the sorted list
            # (:1058): the level ends without a result, the counter goes up by one.
            st = out.best_states
            if ctx.check_swept(params, st[0], st[1], st[2]) >= 0:
                self._infeasible_count_collision += 1
                return None
        return self._gpu_winner_sample(bundle, out)

    # ---- helpers ----------------------------------------------------------------------------------
    def _gpu_sharded_cost_range(self, ctx):
        """min / max / number of the costs over ALL ranks' shards (``TrajectoryBundle.min_costs() / max_costs()``,
        reactive_planner.py:650-651): each rank's range from its device, combined over the group."""
        dist = self.shard_dist

        def fn():
            lo, hi, n = ctx.cost_range()
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, (lo, hi, n))
            parts = [q for q in parts if q[2] > 0]
            if not parts:
                return float("nan"), float("nan"), 0
            return min(q[0] for q in parts), max(q[1] for q in parts), sum(q[2] for q in parts)
        return fn

    def _gpu_candidate_sample(self, bundle: GpuTrajectoryBundle, index: int, lon_coeffs=None, lat_coeffs=None,
                              lat_T=None) -> TrajectorySample:
        if bundle.coeffs is not None:   # polynomials straight from the coefficient arrays of the batch view
            lon_c, lat_c, lon_T, _, lon_end, lat_end = bundle.coeffs
            x0_lon, x0_lat = np.asarray(bundle.x_0_lon, dtype=float), np.asarray(bundle.x_0_lat, dtype=float)
            if self.config.sampling.longitudinal_mode == "stopping":
                lon = QuinticTrajectory(tau_0=0, delta_tau=lon_T[index], x_0=x0_lon, x_d=np.array([lon_end[index], 0.0, 0.0]), coeffs=lon_c[index])
            else:
                lon = QuarticTrajectory(tau_0=0, delta_tau=lon_T[index], x_0=x0_lon, x_d=np.array([lon_end[index], 0.0]), coeffs=lon_c[index])
            lat = QuinticTrajectory(tau_0=0, delta_tau=lon_T[index], x_0=x0_lat, x_d=np.array([lat_end[index], 0.0, 0.0]), coeffs=lat_c[index])
            return TrajectorySample(self.horizon, self.dt, lon, lat)
        if bundle.grids is None:
            return bundle.trajectories[index]
        T, _, L, D = bundle.grids
        nLD = len(L) * len(D)
        iT, r = divmod(index, nLD)
        iL, iD = divmod(r, len(D))
        x0_lon, x0_lat = np.asarray(bundle.x_0_lon, dtype=float), np.asarray(bundle.x_0_lat, dtype=float)
        if self.config.sampling.longitudinal_mode == "stopping":
            lon = QuinticTrajectory(tau_0=0, delta_tau=T[iT], x_0=x0_lon, x_d=np.array([L[iL], 0.0, 0.0]), coeffs=lon_coeffs)
        else:
            lon = QuarticTrajectory(tau_0=0, delta_tau=T[iT], x_0=x0_lon, x_d=np.array([L[iL], 0.0]), coeffs=lon_coeffs)
        tau = T[iT]
        if lat_T is not None and not np.isnan(lat_T):
            tau = lat_T
        elif self._low_vel_mode:
            s_goal = lon.evaluate_state_at_tau(T[iT])[0] - x0_lon[0]
            tau = T[iT] if s_goal <= 0 else s_goal
        lat = QuinticTrajectory(tau_0=0, delta_tau=tau, x_0=x0_lat, x_d=np.array([D[iD], 0.0, 0.0]), coeffs=lat_coeffs)
        return TrajectorySample(self.horizon, self.dt, lon, lat)

    def _gpu_winner_sample(self, bundle, out: PlanOutput) -> TrajectorySample:
        if bundle.coeffs is not None:
            s = self._gpu_candidate_sample(bundle, out.best_index)
        elif bundle.grids is None:
            s = bundle.trajectories[out.best_index]      # foreign sampling space: its own object
        else:
            def polys(b=bundle, o=out):
                t = self._gpu_candidate_sample(b, o.best_index, o.best_lon_coeffs, o.best_lat_coeffs, o.best_lat_T)
                return t.trajectory_long, t.trajectory_lat
            s = _WinnerSample(self.horizon, self.dt, polys)
            s.bind_block(out.best_states, out.best_cost)
            s._cost_function = self.cost_function
            return s
        bind_states(s, out.best_states, 1, out.best_cost)
        s._cost_function = self.cost_function
        return s

    def _gpu_external_costs(self, ctx, bundle, out: PlanOutput) -> PlanOutput:
        status, _ = ctx.fetch_status()
        states = ctx.fetch_states()
        costs = np.full(len(status), np.nan)
        for i in np.flatnonzero(((status & 3) == 1) | ((status & 3) == 3)):
            s = self._gpu_candidate_sample(bundle, int(i))
            bind_states(s, states[i], int(status[i]))
            costs[i] = float(self.cost_function.evaluate(s))
        return ctx.select(costs)

    def _gpu_stored_trajectories(self, ctx, bundle, status, costs) -> List[TrajectorySample]:
        """``stored_trajectories = feasible + infeasible`` of draw_traj_set (reactive_planner.py:1122-1123)."""
        states = ctx.fetch_states()
        feas, infeas = [], []
        for i in range(len(status)):
            lab = int(status[i]) & 3
            s = self._gpu_candidate_sample(bundle, i)
            c = None if np.isnan(costs[i]) else float(costs[i])
            bind_states(s, states[i], int(status[i]), c)
            (feas if lab in (1, 3) else infeas).append(s)
        return feas + infeas


class ReactivePlanner(GpuBackendMixin):
    """Stand-alone planner with the reference's public interface (reactive_planner.py:52-1159)."""

    def __init__(self, config: ReactivePlannerConfiguration, backend_factory=None, device: int = 0):
        if backend_factory is not None:
            self.backend_factory = backend_factory
        self.gpu_device = device
        self.dt: float = config.planning.dt
        self.N: int = config.planning.time_steps_computation
        self.horizon: float = config.planning.dt * config.planning.time_steps_computation
        self.vehicle_params = config.vehicle
        self.x_0: Optional[ReactivePlannerState] = None
        self.x_0_cl: Optional[Tuple[List, List]] = None
        self._co: Optional[CoordinateSystem] = None
        self._obstacle_tables: Optional[ObstacleTables] = None
        self._infeasible_count_collision = 0
        self._infeasible_count_kinematics = 0
        self._infeasible_reason_dict: Dict = dict()
        self._optimal_cost = 0.0
        self._planning_times_list: List = list()
        self._record_state_list: List[ReactivePlannerState] = list()
        self._record_input_list: List = list()
        self.stored_trajectories: Optional[List[TrajectorySample]] = None
        self._desired_speed: Optional[float] = None
        self._desired_lon_position: Optional[float] = None
        self._low_vel_mode = False
        self._draw_traj_set = config.debug.draw_traj_set and (config.debug.show_plots or config.debug.save_plots)
        self.config: Optional[ReactivePlannerConfiguration] = None
        self.reset(config)
        self.sampling_space: Optional[SamplingSpace] = None
        self.set_sampling_space()
        self.sampling_level = config.sampling.num_sampling_levels
        self.cost_function: Optional[CostFunction] = None
        self.set_cost_function()
        self._standstill_lookahead = config.planning.standstill_lookahead

    # ---- read-only views (reactive_planner.py:115-160) ---------------------------------------------
    coordinate_system = property(lambda self: self._co)
    reference_path = property(lambda self: self._co.reference)
    infeasible_count_collision = property(lambda self: self._infeasible_count_collision)
    infeasible_count_kinematics = property(lambda self: self._infeasible_count_kinematics)
    infeasible_reason_dict = property(lambda self: self._infeasible_reason_dict)
    optimal_cost = property(lambda self: self._optimal_cost)
    planning_times = property(lambda self: self._planning_times_list)
    record_state_list = property(lambda self: self._record_state_list)
    record_input_list = property(lambda self: self._record_input_list)
    collision_checker = property(lambda self: self._obstacle_tables)

    def goal_reached(self) -> bool:
        x0 = self.x_0.shift_positions_to_center(self.vehicle_params.wb_rear_axle)
        return bool(self.config.planning_problem.goal.is_reached(x0))

    # ---- set-up (reactive_planner.py:172-389) ------------------------------------------------------
    def reset(self, config=None, initial_state_cart=None, initial_state_curv=None, collision_checker=None,
              coordinate_system=None):
        if config is not None:
            self.config = config
        else:
            assert self.config is not None, "<ReactivePlanner.reset(). No Configuration object provided>"
        self._reset_statistics()
        if collision_checker is None:
            self.set_collision_checker(scenario=self.config.scenario)
        else:
            self.set_collision_checker(collision_checker=collision_checker)
        if coordinate_system is not None:
            self.set_reference_path(coordinate_system=coordinate_system)
        if self.x_0 is None and initial_state_cart is None:
            pp = self.config.planning_problem
            self.x_0 = ReactivePlannerState.create_from_initial_state(
                pp.initial_state, self.vehicle_params.wheelbase, self.vehicle_params.wb_rear_axle) if pp else None
        else:
            self.x_0 = initial_state_cart if initial_state_cart is not None else self.x_0
        self.x_0_cl = initial_state_curv if initial_state_curv is not None else self._compute_initial_states(self.x_0)

    def set_collision_checker(self, scenario=None, collision_checker=None, road_boundary_obstacle=None):
        """``collision_checker`` is an ``ObstacleTables`` here (the GPU path cannot read an opaque pycrcc
        checker); from a scenario the tables are extracted as ``set_collision_checker`` feeds pycrcc
        (reactive_planner.py:234-251); ``road_boundary_obstacle``: (n, 6) triangles."""
        if collision_checker is None:
            tables = _tables_from_scenario(scenario, road_boundary_obstacle,
                                           continuous=bool(self.config.planning.continuous_collision_check)) \
                if scenario is not None else ObstacleTables()
        else:
            assert isinstance(collision_checker, ObstacleTables), "pass commonroad_rp_amd.collision.ObstacleTables"
            tables = collision_checker
        self.set_obstacle_tables(tables)

    def set_reference_path(self, reference_path: np.ndarray = None, coordinate_system: CoordinateSystem = None):
        if coordinate_system is None:
            assert reference_path is not None, "<set reference path>: Please provide a reference path OR a CoordinateSystem object to the planner."
            self._co = CoordinateSystem(reference_path, smooth_reference=True)   # (utils_coordinate_system.py:88: smoothed by default)
        else:
            assert reference_path is None, "<set reference path>: Please provide a reference path OR a CoordinateSystem object to the planner."
            self._co = coordinate_system

    def set_t_sampling_parameters(self, t_min):
        self.sampling_space.samples_t = TimeSampling(t_min, self.horizon, self.sampling_level, self.dt)

    def set_d_sampling_parameters(self, delta_d_min, delta_d_max):
        self.sampling_space.samples_d = PositionSampling(delta_d_min, delta_d_max, self.sampling_level)

    def set_v_sampling_parameters(self, v_min, v_max):
        self.sampling_space.samples_v = VelocitySampling(v_min, v_max, self.sampling_level)

    def set_s_sampling_parameters(self, s_min, s_max):
        self.sampling_space.samples_s = PositionSampling(s_min, s_max, self.sampling_level)

    def set_desired_velocity(self, desired_velocity: float = None, current_speed: float = None, stopping: bool = False):
        self._desired_lon_position = None
        if desired_velocity is None and self._desired_speed is None:
            self._desired_speed = float(self.x_0.velocity)
        else:
            self._desired_speed = desired_velocity if desired_velocity is not None else self._desired_speed
        assert self._desired_speed >= 0.0, f"<ReactivePlanner.set_desired_velocity(): desired speed has to be positive. Provided speed{self._desired_speed}>"
        if not stopping:
            ref = current_speed if current_speed is not None else self._desired_speed
            min_v = max(0, ref - (0.125 * self.horizon * self.vehicle_params.a_max))
            max_v = max(min_v + 5.0, ref + 2)
            self.set_v_sampling_parameters(min_v, max_v)
        else:
            self.set_v_sampling_parameters(v_min=self._desired_speed, v_max=self._desired_speed)
        cf = self.cost_function
        if hasattr(cf, "desired_speed"):
            cf.desired_speed = self._desired_speed
        if hasattr(cf, "w_a"):
            cf.w_a = 5
        if hasattr(cf, "desired_s"):
            cf.desired_s = self._desired_lon_position

    def set_desired_lon_position(self, lon_position: float, delta_s_min: Optional[float] = None,
                                 delta_s_max: Optional[float] = None):
        self._desired_lon_position = lon_position
        self._desired_speed = 0.0
        if delta_s_min is None and delta_s_max is None:
            delta_s_min, delta_s_max = self.config.sampling.s_min, self.config.sampling.s_max
        self.set_s_sampling_parameters(s_min=lon_position + delta_s_min, s_max=lon_position + delta_s_max)
        cf = self.cost_function
        if hasattr(cf, "desired_s"):
            cf.desired_s = self._desired_lon_position
        if hasattr(cf, "desired_speed"):
            cf.desired_speed = self._desired_speed
        if hasattr(cf, "w_a"):
            cf.w_a = 1

    def set_cost_function(self, cost_function: CostFunction = None):
        self.cost_function = cost_function if cost_function else DefaultCostFunction(
            self._desired_speed, desired_d=0.0, desired_s=self._desired_lon_position)

    def set_sampling_space(self, sampling_space: SamplingSpace = None):
        self.sampling_space = sampling_space if sampling_space else sampling_space_factory(self.config)

    def record_state_and_input(self, state: ReactivePlannerState):
        self._record_state_list.append(state)
        rate = (state.steering_angle - self._record_state_list[-2].steering_angle) / self.dt \
            if len(self._record_state_list) > 1 else 0.0
        self._record_input_list.append(dict(time_step=state.time_step, acceleration=state.acceleration,
                                            steering_angle_speed=rate))

    def _reset_statistics(self):
        self._optimal_cost = 0
        self._infeasible_count_kinematics = 0
        self._infeasible_count_collision = 0
        for constraint in self.config.planning.constraints_to_check:
            self._infeasible_reason_dict[constraint] = 0

    # ---- initial state in the curvilinear frame (reactive_planner.py:446-512) ----------------------
    def _compute_initial_states(self, x_0: ReactivePlannerState):
        if not self._co or x_0 is None:
            return None
        try:
            return self._co.initial_state(x_0.position[0], x_0.position[1], x_0.orientation, x_0.velocity,
                                          acceleration=x_0.acceleration or 0.0, steering_angle=x_0.steering_angle or 0.0,
                                          wheelbase=self.vehicle_params.wheelbase, low_vel_mode=self._low_vel_mode)
        except ValueError:
            logger.critical("Initial state could not be transformed.")
            raise ValueError("Initial state could not be transformed.")

    # ---- one replanning cycle (reactive_planner.py:570-665) ----------------------------------------
    def plan(self, current_sampling_level: int = None) -> Optional[tuple]:
        start = time.time()
        assert self.x_0 is not None, "<ReactivePlanner.plan(): Planner Cartesian initial state is empty!>"
        assert self._co is not None, "<ReactivePlanner.plan(): No coordinate system given. Call set_reference_path()>"
        optimal, bundle = None, None
        i = 1 if current_sampling_level is None else current_sampling_level
        prev_low = self._low_vel_mode
        if self._fast_path_ok():
            # (one foreign call per cycle -- rp_plan_cycle -- where the context has it: a missing curvilinear state is then worked
            #  out inside that call, with the low-velocity flag of the previous cycle, as below)
            x_0_lon, x_0_lat = self.x_0_cl if self.x_0_cl else (None, None)
            result = self._plan_fast(x_0_lon, x_0_lat, i, current_sampling_level is not None, prev_low)
            if result is not NotImplemented:
                self._planning_times_list.append(time.time() - start)
                if result is None:
                    logger.warning("Planner failed to find an optimal trajectory with given sampling configuration!")
                return result
            self._low_vel_mode = prev_low
        if not self.x_0_cl:
            self.x_0_cl = self._compute_initial_states(self.x_0)
        assert self.x_0_cl is not None, "<ReactivePlanner.plan(): Planner curvilinear initial state is empty!>"
        x_0_lon, x_0_lat = self.x_0_cl
        self._low_vel_mode = bool(self.x_0.velocity < self.config.planning.low_vel_mode_threshold)
        while optimal is None and i < self.sampling_level:
            bundle = self._create_trajectory_bundle(x_0_lon, x_0_lat, samp_level=i)
            optimal = self._get_optimal_trajectory(bundle)
            if current_sampling_level is not None:
                break
            i += 1

        # (reactive_planner.py:641; the cheap operand first: the winner's containers are built on access)
        if self.x_0.velocity <= 0.05 and (optimal is None or optimal.cartesian.v[self._standstill_lookahead] <= 0.05):
            optimal = self._compute_standstill_trajectory()
            self._optimal_cost = optimal.cost

        result = self._compute_trajectory_pair(optimal) if optimal is not None else None
        self._planning_times_list.append(time.time() - start)
        if result is None:
            logger.warning("Planner failed to find an optimal trajectory with given sampling configuration!")
        return result

    # ---- the same cycle without the objects in between ---------------------------------------------
    def _fast_path_ok(self) -> bool:
        """The standard configuration of a moving vehicle -- fixed-interval sampling, a built-in cost function, no drawing, no
        continuous collision check, no sharding -- and nobody has replaced one of the three methods the cycle goes through
        (subclass or instance attribute): then ``plan()`` may skip the bundle, winner-sample and container objects, none of
        which it reads itself (the standstill branch, which does, is left to the general loop)."""
        cls, d = type(self), self.__dict__
        plain = cls.__dict__.get("_fast_cls_ok")   # (per class: none of the cycle's methods is overridden by it)
        if plain is None:
            plain = (cls._get_optimal_trajectory is GpuBackendMixin._get_optimal_trajectory
                     and cls._create_trajectory_bundle is GpuBackendMixin._create_trajectory_bundle
                     and cls._compute_trajectory_pair is ReactivePlanner._compute_trajectory_pair
                     and cls._compute_standstill_trajectory is ReactivePlanner._compute_standstill_trajectory)
            cls._fast_cls_ok = plain
        return (plain and type(self.sampling_space) is FixedIntervalSampling and not self._draw_traj_set and self.shard_dist is None
                and self.x_0.velocity > 0.05 and not self.config.planning.continuous_collision_check
                and "_compute_trajectory_pair" not in d and "_get_optimal_trajectory" not in d and "_create_trajectory_bundle" not in d)

    #: How ``plan()`` hands the sampling levels of a cycle to the device (reactive_planner.py:616-636 visits them one by one):
    #:   "chain"      every level the loop could visit in ONE call (``rp_plan_levels``: the kernels of all levels back to back on the
    #:                stream, the epilogue of a level that finds a winner stops the levels behind it) -- one device round trip whatever
    #:                level delivers, at the price of launching the later levels every cycle (ZAM_Over, levels 1-3: p50 54 us against
    #:                47 us for the first level alone; a cycle in which all three fail: 89 us instead of three round trips);
    #:   "adaptive"   (default) the first level alone -- in the reference's shipped scenarios it always delivers -- and, when it has
    #:                no winner, the REMAINING levels in one call; after such a cycle the next ``LEVEL_CHAIN_CYCLES`` cycles are
    #:                handed over whole (a scene that needed a finer level tends to need it again);
    #:   "sequential" one call per level.
    level_policy = "adaptive"
    LEVEL_CHAIN_CYCLES = 16
    #: ``rp_plan_cycle`` for every cycle of the fast path, not only for those that start from a Cartesian state (tests, measurements)
    always_plan_cycle = False

    def _plan_fast(self, x_0_lon, x_0_lat, level: int, single_level: bool, prev_low_vel_mode: bool = False):
        """The level loop of ``plan()`` (reactive_planner.py:616-636) on the arrays themselves: grids -> ``rp_plan_levels`` (or
        ``rp_plan_packed`` for one level) -> counters, the output packing of :514-568 done in the same call.  Which levels travel
        together: ``level_policy``.  Same results, same counters as the general loop (tests/test_planner_mirror.py runs both)."""
        cost = self._gpu_cost()
        if cost is None:
            return NotImplemented
        ctx = self._gpu_ctx()
        sp, mode = self.sampling_space, self.config.sampling.longitudinal_mode
        hook = self.on_device_launched
        x0 = self.x_0
        # rp_plan_cycle: initial state (if missing), low-velocity flag, the lateral samples' union with the current offset, the level
        # loop and the output packing in ONE foreign call
        # -- for the cycles that START FROM A CARTESIAN STATE: there the separate path pays a second foreign call with five table
        # arguments (rp_initial_state).  With the curvilinear state at hand (every cycle of a closed loop but the first: reset() gets
        # it from the last trajectory) the separate calls below are the faster Python (profiles/r05_plan_latency.txt: cfg2 p50 43.2
        # vs 47.3 us -- filling the cycle struct through ctypes costs more than the set union it saves).
        cyc = None
        plan_cycle = getattr(ctx, "plan_cycle", None) if (hook is None and (x_0_lon is None or self.always_plan_cycle)) else None
        if plan_cycle is not None and getattr(sp, "cycle_ok", None) is not None and sp.cycle_ok():
            cyc = self.__dict__.get("_rp_cycle")
            if cyc is None:
                cyc = self._rp_cycle = _capi.RpCycle()
            cyc.prev_low_vel_mode = 1 if prev_low_vel_mode else 0
            cyc.x, cyc.y = x0.position[0], x0.position[1]
            cyc.orientation, cyc.velocity = x0.orientation, x0.velocity
            cyc.acceleration, cyc.steering_angle = x0.acceleration or 0.0, x0.steering_angle or 0.0
            cyc.low_vel_mode_threshold = self.config.planning.low_vel_mode_threshold
            if x_0_lon is None:
                cyc.have_curvilinear = 0
            else:
                cyc.have_curvilinear = 1
                cyc.x0_lon[0], cyc.x0_lon[1], cyc.x0_lon[2] = x_0_lon
                cyc.x0_lat[0], cyc.x0_lat[1], cyc.x0_lat[2] = x_0_lat
        elif x_0_lon is None:
            self.x_0_cl = self._compute_initial_states(x0)
            assert self.x_0_cl is not None, "<ReactivePlanner.plan(): Planner curvilinear initial state is empty!>"
            x_0_lon, x_0_lat = self.x_0_cl
        self._low_vel_mode = bool(x0.velocity < self.config.planning.low_vel_mode_threshold)
        params = None   # (built where a call takes the struct filled in: _gpu_params; the single-level call fills it itself)
        stop = min(level + 1, self.sampling_level) if single_level else self.sampling_level
        self._reset_statistics()
        if level >= stop:
            return None
        policy = self.level_policy
        chain_left = self.__dict__.get("_chain_cycles", 0)
        whole = policy == "chain" or (policy == "adaptive" and chain_left > 0)
        res = blk = buf = None
        first, decided = level, level
        while level < stop and blk is None:
            # the levels of this call: all that are left, or the next one alone
            upto = stop if (whole or (policy == "adaptive" and level > first)) else level + 1
            many = upto - level > 1
            if cyc is not None:
                levels = [sp.cycle_level(k, mode) for k in range(level, upto)]
                if any(lv is None for lv in levels):
                    cyc = None   # (a lateral sample set that is not the level's linspace: the general path from here on)
                    if x_0_lon is None:
                        self.x_0_cl = self._compute_initial_states(x0)
                        x_0_lon, x_0_lat = self.x_0_cl
                        params = None
                else:
                    if params is None:
                        params = self._gpu_params(x_0_lon if x_0_lon is not None else _ZERO3, x_0_lat if x_0_lat is not None else _ZERO3, 0)
                    res, blk, buf = plan_cycle(cyc, params, cost, levels)
                    if type(res) is int:   # the initial state could not be worked out: as _compute_initial_states raises
                        if res == _capi.E_DOMAIN:
                            logger.critical("Initial state could not be transformed.")
                            raise ValueError("Initial state could not be transformed.")
                        raise Exception("Initial state or reference incorrect! The longitudinal velocity along the reference path is negative: "
                                        "the ego vehicle does not drive in the direction of the reference path")
                    if cyc.have_curvilinear == 0:
                        x_0_lon, x_0_lat = list(cyc.x0_lon), list(cyc.x0_lat)
                        self.x_0_cl = (x_0_lon, x_0_lat)
                        cyc.have_curvilinear = 1
                        if not self.always_plan_cycle:   # (levels still to come go through the separate calls: the state is at hand now)
                            cyc = None
                        params = None
                    decided = level + self._rp_cycle.level
                    level = upto
                    continue
            packed = getattr(ctx, "plan_levels_packed" if many else "plan_packed", None) if hook is None else None
            begin = getattr(ctx, "plan_levels_begin", None) if (hook is not None or (many and packed is None)) else None
            fastcall = getattr(ctx, "plan_packed_fast", None) if (packed is not None and not many) else None
            if fastcall is not None:
                # one level, no hook: the cycle's fields of rp_params, the grids and the call itself in the binding's extension module,
                # which reads the velocity / lateral sample SETS themselves (no arrays built to be copied)
                sets = getattr(sp, "grid_sets_at_level", None)
                g4 = sets(level, x_0_lon, mode) if sets is not None else None
                T, traj_len, L, D = g4 if g4 is not None else sp.grids_at_level(level, x_0_lon, x_0_lat, mode)
                if len(T) and len(L):   # (D holds the current offset at least)
                    res, blk, buf = fastcall(self._gpu_params_struct(), cost, T, traj_len, L, D, int(x0.time_step), self._low_vel_mode, 0,
                                             x_0_lon, x_0_lat, x0.orientation)
                else:
                    res = None   # (an empty bundle: the reference's loop visits it and leaves the counters at zero)
                decided = level
                level = upto
                continue
            levels = [sp.grids_at_level(k, x_0_lon, x_0_lat, mode) for k in range(level, upto)] if many else \
                (sp.grids_at_level(level, x_0_lon, x_0_lat, mode),)
            if params is None:
                params = self._gpu_params(x_0_lon, x_0_lat, 0)
            if packed is not None and many:
                res, k, blk, buf = packed(params, cost, levels)
                decided = level + k
            elif packed is not None:
                T, traj_len, L, D = levels[0]
                if len(T) * len(L) * len(D):
                    res, blk, buf = packed(params, cost, T, traj_len, L, D)
                else:
                    res = None   # (an empty bundle: the reference's loop visits it and leaves the counters at zero)
                decided = level
            elif begin is not None and not any(len(T) * len(L) * len(D) for (T, _tl, L, D) in levels):
                res = None       # (nothing to launch at these levels)
                decided = upto - 1
            elif begin is not None:
                # the same in two halves: the caller's own work (harness: the last cycle's bookkeeping) runs while the device is busy
                begin(params, cost, levels)
                if hook is not None and level == first:
                    hook()
                res = ctx.plan_wait()
                blk = res.best_states if res.best_index >= 0 else None
                decided = level + (ctx.last_level() if hasattr(ctx, "last_level") else 0)
            else:   # a context without these calls (test doubles, recorders): level by level
                upto = level + 1
                T, traj_len, L, D = levels[0]
                if len(T) * len(L) * len(D):
                    res = ctx.plan(PlanInputs.trusted(params, cost, T, traj_len, L, D))
                    blk = res.best_states if res.best_index >= 0 else None
                else:
                    res = None
                decided = level
            level = upto
        if not single_level and policy == "adaptive":
            self._chain_cycles = self.LEVEL_CHAIN_CYCLES if (blk is None or decided > first) else max(0, chain_left - 1)
        if res is not None:
            rd = self._infeasible_reason_dict
            self._infeasible_count_kinematics = res.n_candidates - res.n_feasible
            self._infeasible_count_collision = res.n_collision_before_best
            rc = res.reason_counts[:]   # (one conversion to a list: indexing a ctypes array costs a call per element)
            for k, name in _REASON_ITEMS:
                if name in rd:
                    rd[name] = rc[k]
        if blk is None:
            return None
        if buf is None:   # output packing: positions, lon / lat samples, shifted orientations, steering angles and yaw rates in one pass in C
            buf = _capi.pack_trajectory(blk, self.dt, self.vehicle_params.wheelbase, self.x_0.orientation)
        return self._fast_output(blk, buf)

    def _fast_output(self, blk, buf):
        """(Cartesian trajectory, curvilinear trajectory, lon list, lat list) of reactive_planner.py:514-568 from the winner's state
        block [14, N + 1] and its packed form [N + 1, 13] (``rp_pack_trajectory``); state objects are built on access."""
        t0, factor, n = self.x_0.time_step, self.config.planning.factor, blk.shape[1]
        yaw0 = self.x_0.yaw_rate
        new_state, RS, CS = object.__new__, ReactivePlannerState, CustomState
        # (rows of blk: x y theta v a kappa ...; columns of buf: x y | s d | s s' s'' | d d' d'' | theta shifted | steering angle | yaw
        #  rate.  Nothing is sliced here: a cycle reads a state or two of the lists, the views are made where a state is built)

        def cart_state(i):
            st = new_state(RS)
            row = buf[i]
            st.__dict__ = {"time_step": t0 + factor * i, "position": row[0:2], "orientation": float(row[10]), "velocity": float(blk[3, i]),
                           "steering_angle": float(row[11]), "acceleration": float(blk[4, i]), "yaw_rate": float(row[12]) if i else yaw0}
            return st

        def curv_state(i):
            sc = new_state(CS)
            sc.__dict__ = {"time_step": t0 + factor * i, "position": buf[i, 2:4], "orientation": float(blk[2, i]), "velocity": float(blk[3, i]),
                           "acceleration": float(blk[4, i]), "yaw_rate": float(blk[5, i])}
            return sc
        # (the lon / lat lists -- [s, s', s''] and [d, d', d''] per step, reactive_planner.py:552-553 -- are lists whose rows are built
        #  on access as well: the loop reads one row of each per cycle, run_planner.py:84-85)
        return (Trajectory(t0, LazyStateList(n, cart_state)), Trajectory(t0, LazyStateList(n, curv_state)),
                LazyStateList(n, lambda i: buf[i, 4:7].tolist()), LazyStateList(n, lambda i: buf[i, 7:10].tolist()))

    def _compute_standstill_trajectory(self) -> TrajectorySample:
        """reactive_planner.py:667-713 (arrays of length N, not N + 1, as in the reference)."""
        from .trajectories import CartesianSample, CurviLinearSample
        x_0 = self.x_0
        x_0_lon, x_0_lat = self.x_0_cl
        lon = QuarticTrajectory(tau_0=0, delta_tau=self.horizon, x_0=np.asarray(x_0_lon), x_d=np.array([0, 0]))
        lat = QuinticTrajectory(tau_0=0, delta_tau=self.horizon, x_0=np.asarray(x_0_lat), x_d=np.array([x_0_lat[0], 0, 0]))
        kappa_0 = np.tan(x_0.steering_angle) / self.vehicle_params.wheelbase
        p = TrajectorySample(self.horizon, self.dt, lon, lat)
        a = np.repeat(0.0, self.N)
        a[1] = -self.x_0.velocity / self.dt
        rep = lambda v: np.repeat(v, self.N)   # noqa: E731
        p.cartesian = CartesianSample(rep(x_0.position[0]), rep(x_0.position[1]), rep(x_0.orientation), rep(0.0), a,
                                      rep(kappa_0), rep(0.0), current_time_step=self.N)
        ref_pos = self._co.ref_pos
        s_idx = int(np.argmax(ref_pos > x_0_lon[0])) - 1
        ref_theta = np.unwrap(self._co.ref_theta)
        theta_cl = x_0.orientation - interpolate_angle(x_0_lon[0], ref_pos[s_idx], ref_pos[s_idx + 1], ref_theta[s_idx],
                                                       ref_theta[s_idx + 1])
        p.curvilinear = CurviLinearSample(rep(x_0_lon[0]), rep(x_0_lat[0]), rep(theta_cl), dd=rep(x_0_lat[1]),
                                          ddd=rep(x_0_lat[2]), ss=rep(x_0_lon[1]), sss=rep(x_0_lon[2]),
                                          current_time_step=self.N)
        return p

    def _compute_trajectory_pair(self, trajectory: TrajectorySample):
        """Output packing, reactive_planner.py:514-568: (Cartesian trajectory, curvilinear trajectory, lon list, lat list).
        Same values as the reference's; the two state lists build their state objects on access (``LazyStateList``)."""
        t0, factor, dt = self.x_0.time_step, self.config.planning.factor, self.dt
        blk = getattr(trajectory, "_state_block", None)
        whole = type(trajectory) is _WinnerSample and blk is not None and trajectory._samples is None
        if whole:   # the winner of this build's own path: everything comes from rows of its [14, N + 1] block
            n = blk.shape[1]
            theta, v, acc, kappa = blk[2], blk[3], blk[4], blk[5]
            ca = cu = None
        else:
            ca, cu = trajectory.cartesian, trajectory.curvilinear
            n = len(ca.x)
            theta = np.asarray(ca.theta, dtype=float)
            kappa = np.asarray(ca.kappa, dtype=float)
            v, acc = np.asarray(ca.v, dtype=float), np.asarray(ca.a, dtype=float)
        yaw = np.empty(n)
        yaw[0] = 0.0                                                              # (state 0 carries x_0.yaw_rate as it is, None included: cart_state)
        yaw[1:] = (theta[1:] - theta[:-1]) / dt                                   # :531-537
        yaw0 = self.x_0.yaw_rate
        steer = np.arctan2(self.vehicle_params.wheelbase * kappa, 1.0)            # :539
        # shift_orientation (utility/general.py:49-55) on the Cartesian trajectory only; the curvilinear states keep theta
        lo, hi = self.x_0.orientation - np.pi, self.x_0.orientation + np.pi
        th_c = theta
        if theta.min() < lo or theta.max() > hi:
            th_c = theta.copy()
            for i, o in enumerate(th_c):
                while o < lo:
                    o += 2 * np.pi
                while o > hi:
                    o -= 2 * np.pi
                th_c[i] = o
        # the winner's arrays are rows of one [14, N + 1] block (bind_states): positions and the lon / lat lists then come from
        # slices of it instead of column-by-column copies
        if not whole:
            blk = ca.x.base
            whole = (blk is not None and blk.shape == (14, n) and ca.y.base is blk and cu.s.base is blk and cu.d.base is blk
                     and cu.s_dot.base is blk and cu.s_ddot.base is blk and cu.d_dot.base is blk and cu.d_ddot.base is blk
                     and getattr(trajectory, "_state_block", None) is blk)
        if whole:
            pos, sd = blk[0:2].T.copy(), blk[7:9].T.copy()          # rows x, y | s, d
        else:
            pos = np.empty((n, 2))
            pos[:, 0], pos[:, 1] = ca.x, ca.y
            sd = np.empty((n, 2))
            sd[:, 0], sd[:, 1] = cu.s, cu.d
        # (states are filled through __dict__: a dataclass __init__ call costs three times as much)
        new_state, RS, CS = object.__new__, ReactivePlannerState, CustomState

        def cart_state(i):
            st = new_state(RS)
            st.__dict__ = {"time_step": t0 + factor * i, "position": pos[i], "orientation": float(th_c[i]), "velocity": float(v[i]),
                           "steering_angle": float(steer[i]), "acceleration": float(acc[i]), "yaw_rate": float(yaw[i]) if i else yaw0}
            return st

        def curv_state(i):
            sc = new_state(CS)
            sc.__dict__ = {"time_step": t0 + factor * i, "position": sd[i], "orientation": float(theta[i]), "velocity": float(v[i]),
                           "acceleration": float(acc[i]), "yaw_rate": float(kappa[i])}
            return sc
        if whole:
            lon, lat = blk[_LON_ROWS].T.tolist(), blk[_LAT_ROWS].T.tolist()   # rows s, s_dot, s_ddot | d, d_dot, d_ddot
        else:
            lo3 = np.empty((n, 3))
            lo3[:, 0], lo3[:, 1], lo3[:, 2] = cu.s, cu.s_dot, cu.s_ddot
            la3 = np.empty((n, 3))
            la3[:, 0], la3[:, 1], la3[:, 2] = cu.d, cu.d_dot, cu.d_ddot
            lon, lat = lo3.tolist(), la3.tolist()
        return (Trajectory(t0, LazyStateList(n, cart_state)), Trajectory(t0, LazyStateList(n, curv_state)), lon, lat)
