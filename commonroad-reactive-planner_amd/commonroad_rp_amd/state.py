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


class LazyStateList:
    """The ``state_list`` of a planned trajectory as a read-only sequence whose state objects are built when they are
    asked for.  The replanning loop reads one or two states of the optimal trajectory per cycle
    (run_planner.py:81-86,102-107: ``optimal[0].state_list[1 + temp]``); building all 2 (N + 1) objects of both output
    trajectories on every ``plan()`` was the largest single item of its Python time.  Indexing, slicing, ``len``,
    iteration, ``list(...)`` and comparison with lists work; a built state is kept, so repeated access returns the
    same object as a list would."""

    __slots__ = ("_n", "_make", "_built")

    def __init__(self, n: int, make):
        self._n, self._make, self._built = n, make, [None] * n

    def __len__(self):
        return self._n

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self._n))]
        if k < 0:
            k += self._n
        if not 0 <= k < self._n:
            raise IndexError("state index out of range")
        st = self._built[k]
        if st is None:
            st = self._built[k] = self._make(k)
        return st

    def __iter__(self):
        return (self[k] for k in range(self._n))

    def __eq__(self, other):
        try:
            return len(other) == self._n and all(a is b or a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return f"LazyStateList({self._n} states, {sum(s is not None for s in self._built)} built)"
