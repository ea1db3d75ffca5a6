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


class LazyStateList(list):
    """The ``state_list`` of a planned trajectory: a ``list`` whose state objects are built when they are asked for.  The
    replanning loop reads one or two states of the optimal trajectory per cycle (run_planner.py:81-86,102-107:
    ``optimal[0].state_list[1 + temp]``); building all 2 (N + 1) objects of both output trajectories on every ``plan()`` was the
    largest single item of its Python time.  It IS a list (``isinstance``, ``+``, ``*``, slicing, pickling -- as a plain list --,
    ``json``, mutation all work as on the reference's ``state_list``): indexing by an integer builds that one state, any other
    operation builds the states still missing first and then is the list operation."""

    __slots__ = ("_make", "_missing")

    def __init__(self, n: int = 0, make=None):
        list.__init__(self, [None] * n if make is not None else ())
        self._make, self._missing = make, (n if make is not None else 0)

    def _fill(self):
        if getattr(self, "_missing", 0):
            get, put, make = list.__getitem__, list.__setitem__, self._make
            for k in range(list.__len__(self)):
                if get(self, k) is None:
                    put(self, k, make(k))
            self._missing, self._make = 0, None
        return self

    def __getitem__(self, k):
        if getattr(self, "_missing", 0):
            if isinstance(k, slice):
                self._fill()
            else:
                st = list.__getitem__(self, k)       # (IndexError as a list raises it)
                if st is None:
                    st = self._make(k if k >= 0 else k + list.__len__(self))
                    list.__setitem__(self, k, st)
                    self._missing -= 1
                return st
        return list.__getitem__(self, k)

    def __iter__(self):
        return list.__iter__(self._fill())

    def __reduce_ex__(self, protocol):
        return (list, (list(self),))

    def __repr__(self):
        missing = getattr(self, "_missing", 0)
        return list.__repr__(self) if not missing else f"LazyStateList({len(self)} states, {len(self) - missing} built)"


def _filled(name):
    op = getattr(list, name)

    def method(self, *a, **k):
        return op(self._fill(), *a, **k)
    method.__name__ = name
    return method


for _name in ("__add__", "__radd__", "__iadd__", "__mul__", "__rmul__", "__imul__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__",
              "__ge__", "__contains__", "__reversed__", "__setitem__", "__delitem__", "append", "extend", "insert", "remove", "pop",
              "index", "count", "copy", "sort", "reverse", "clear"):
    if hasattr(list, _name):
        setattr(LazyStateList, _name, _filled(_name))
LazyStateList.__hash__ = None
