"""Closed-loop replanning harness: the driver loop of the reference (run_planner.py:61-107) around a
planner with the reference interface, without the scenario I/O / visualisation around it.

Per simulated time step (run_planner.py:62-107):
  * every ``replanning_frequency``-th step: ``set_desired_velocity(current_speed=x_0.velocity)``, ``plan()``,
    adopt state 1 of the new optimal trajectory;
  * in between: follow the last optimal trajectory (state ``1 + temp``);
  * ``reset(initial_state_cart=..., initial_state_curv=(lon[k], lat[k]), collision_checker=..., coordinate_system=...)``
    so that tables stay resident (the GPU context keeps its reference / obstacle tables as long as the same
    objects are handed back).
Works with ``commonroad_rp_amd.reactive_planner.ReactivePlanner`` and with the reference's class (or the
``GpuBackendMixin`` mixed into it): only the reference's public methods are used.

``overlap=True`` ("overlap next-cycle preparation with current-cycle output packing"): the bookkeeping of a cycle --
the ``on_step`` callback, e.g. recording / logging / visualising its optimal trajectory, whose state objects are built on access
-- runs while the kernels of the NEXT cycle are on the device, between ``rp_plan_begin`` and ``rp_plan_wait``
(``ReactivePlanner.on_device_launched``).  The planner's inputs never depend on it, so the loop's results are the same.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Callable, List, Optional


@dataclasses.dataclass
class LoopResult:
    states: List            # recorded planner states (one per simulated step, first = initial state)
    plan_times: List[float]  # wall time of every plan() call
    n_replans: int
    completed: bool         # False if plan() failed before max_steps / goal


def run_closed_loop(planner, max_steps: int, replanning_frequency: Optional[int] = None,
                    goal_reached: Optional[Callable[[], bool]] = None, desired_velocity: Optional[float] = None,
                    on_step: Optional[Callable] = None, overlap: bool = False) -> LoopResult:
    freq = replanning_frequency or planner.config.planning.replanning_frequency
    planner.record_state_and_input(planner.x_0)
    plan_times: List[float] = []
    optimal = None
    done = goal_reached or (lambda: False)
    deferred: List = []   # on_step calls of finished cycles, run while the next cycle's kernels are on the device
    overlap = overlap and on_step is not None and hasattr(planner, "on_device_launched")

    def flush():
        while deferred:
            on_step(*deferred.pop(0))
    if overlap:
        planner.on_device_launched = flush
    while not done() and len(planner.record_state_list) - 1 < max_steps:
        current_count = len(planner.record_state_list) - 1
        temp = current_count % freq
        if temp == 0:
            if callable(desired_velocity):   # a schedule: desired velocity of this replanning cycle (e.g. hold, then pull away)
                planner.set_desired_velocity(desired_velocity=desired_velocity(current_count), current_speed=planner.x_0.velocity)
            elif desired_velocity is not None:
                planner.set_desired_velocity(desired_velocity=desired_velocity, current_speed=planner.x_0.velocity)
            else:
                planner.set_desired_velocity(current_speed=planner.x_0.velocity)
            t0 = time.perf_counter()
            optimal = planner.plan()
            plan_times.append(time.perf_counter() - t0)
            if not optimal:
                flush()
                if overlap:
                    planner.on_device_launched = None
                return LoopResult(list(planner.record_state_list), plan_times, len(plan_times), False)
        k = 1 + temp
        planner.record_state_and_input(optimal[0].state_list[k])
        planner.reset(initial_state_cart=planner.record_state_list[-1],
                      initial_state_curv=(optimal[2][k], optimal[3][k]),
                      collision_checker=planner.collision_checker, coordinate_system=planner.coordinate_system)
        if on_step is not None:
            if overlap:
                deferred.append((current_count, planner, optimal))
            else:
                on_step(current_count, planner, optimal)
    flush()
    if overlap:
        planner.on_device_launched = None
    return LoopResult(list(planner.record_state_list), plan_times, len(plan_times), True)
