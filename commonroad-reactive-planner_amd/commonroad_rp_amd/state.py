"""Planner state (reference: commonroad_rp/state.py:7-67 ``ReactivePlannerState(KSState)``).
Position refers to the rear axle.  Stand-alone dataclass: commonroad-io is not required."""
from __future__ import annotations

import dataclasses
from typing import Any

import numpy as np


@dataclasses.dataclass(eq=False)
class ReactivePlannerState:
    time_step: Any = None
    position: Any = None
    orientation: Any = None
    velocity: Any = None
    steering_angle: Any = None
    acceleration: Any = None
    yaw_rate: Any = None

    def shift_positions_to_center(self, wb_rear_axle: float):
        """Rear axle -> vehicle centre (state.py:21-30)."""
        o = self.orientation
        new = dataclasses.replace(self)
        new.position = np.asarray(self.position, dtype=float) + np.array([wb_rear_axle * np.cos(o), wb_rear_axle * np.sin(o)])
        return new

    @classmethod
    def create_from_initial_state(cls, initial_state, wheelbase: float, wb_rear_axle: float):
        """Centre-referenced initial state -> rear-axle planner state (state.py:32-67)."""
        o = initial_state.orientation
        pos = np.asarray(initial_state.position, dtype=float) - np.array([wb_rear_axle * np.cos(o), wb_rear_axle * np.sin(o)])
        yaw_rate = getattr(initial_state, "yaw_rate", 0.0) or 0.0
        v = initial_state.velocity
        return cls(time_step=initial_state.time_step, position=pos, orientation=o, velocity=v,
                   acceleration=getattr(initial_state, "acceleration", 0.0) or 0.0, yaw_rate=yaw_rate,
                   steering_angle=np.arctan2(wheelbase * yaw_rate, v))


@dataclasses.dataclass(eq=False)
class CustomState:
    time_step: Any = None
    position: Any = None
    orientation: Any = None
    velocity: Any = None
    acceleration: Any = None
    yaw_rate: Any = None


class Trajectory:
    def __init__(self, initial_time_step, state_list):
        self.initial_time_step = initial_time_step
        self.state_list = state_list
