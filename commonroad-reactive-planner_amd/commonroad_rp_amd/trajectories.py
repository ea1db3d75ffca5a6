"""Trajectory containers of the planner's plug-in surface.

Same names, fields and meaning as ``commonroad_rp.trajectories`` (reference:
commonroad_rp/trajectories.py:18-22 FeasibilityStatus, :61-75 CartesianSample, :200-213
CurviLinearSample, :335-463 TrajectorySample, :466-558 TrajectoryBundle), so that code written
against the reference keeps working with what the GPU path returns.  The arrays inside are views
into the ``[14, N+1]`` state block the device produced; horizon extension (``enlarge``) has already
happened on the device (``current_time_step == N + 1``).
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional

import numpy as np

from ._capi import ARRAY_NAMES


class FeasibilityStatus(Enum):
    FEASIBLE = "feasible"
    INFEASIBLE_KINEMATIC = "infeasible_kinematic"
    INFEASIBLE_COLLISION = "infeasible_collision"


_LABELS = {0: None, 1: FeasibilityStatus.FEASIBLE, 2: FeasibilityStatus.INFEASIBLE_KINEMATIC,
           3: FeasibilityStatus.INFEASIBLE_COLLISION}


def label_from_status(status: int) -> Optional[FeasibilityStatus]:
    return _LABELS[int(status) & 3]


def bind_states(sample, block: np.ndarray, label_code: int, cost=None):
    """Attach a ``[14, N+1]`` device state block to a trajectory sample.  Works for this module's
    ``TrajectorySample`` and for foreign sample classes with the reference's interface (e.g. the
    reference's own objects coming out of a plug-in sampling space): the containers and the label
    enum are taken from the module that defines the sample's class."""
    n = block.shape[1]
    x, y, theta, v, a, kappa, kappa_dot, s, d, theta_cl, s_dot, s_ddot, d_dot, d_ddot = block   # ARRAY_NAMES order
    lab = _LABELS[int(label_code) & 3]
    if isinstance(sample, TrajectorySample):   # this module's classes
        sample.cartesian = CartesianSample(x, y, theta, v, a, kappa, kappa_dot, current_time_step=n)
        sample.curvilinear = CurviLinearSample(s, d, theta_cl, dd=d_dot, ddd=d_ddot, ss=s_dot, sss=s_ddot, current_time_step=n)
        sample.feasibility_label = lab
    else:
        import sys
        mod = sys.modules[type(sample).__module__]
        cart_cls = getattr(mod, "CartesianSample", CartesianSample)
        curv_cls = getattr(mod, "CurviLinearSample", CurviLinearSample)
        status_cls = getattr(mod, "FeasibilityStatus", FeasibilityStatus)
        sample.cartesian = cart_cls(x, y, theta, v, a, kappa, kappa_dot, current_time_step=n)
        sample.curvilinear = curv_cls(s, d, theta_cl, dd=d_dot, ddd=d_ddot, ss=s_dot, sss=s_ddot, current_time_step=n)
        sample.feasibility_label = None if lab is None else status_cls[lab.name]
    if cost is not None:
        sample._cost = cost
    try:
        sample._state_block = block if block.base is None else None   # (a row's .base is the block only if the block owns its data)
    except AttributeError:   # foreign sample classes with __slots__
        pass
    return sample


class Sample:
    def __init__(self, current_time_step: int):
        self.current_time_step = current_time_step

    def length(self) -> int:
        raise NotImplementedError

    def enlarge(self, dt: float):
        raise NotImplementedError("samples returned by the GPU path are already extended to the full horizon")


class CartesianSample(Sample):
    """x, y, theta, v, a, kappa, kappa_dot over the horizon (trajectories.py:61-75)."""

    def __init__(self, x, y, theta, v, a, kappa, kappa_dot, current_time_step: int):
        super().__init__(current_time_step)
        self.x, self.y, self.theta, self.v, self.a, self.kappa, self.kappa_dot = x, y, theta, v, a, kappa, kappa_dot

    def length(self) -> int:
        return len(self.x)


class CurviLinearSample(Sample):
    """s, d, theta, s_dot, s_ddot, d_dot, d_ddot over the horizon (trajectories.py:200-213)."""

    def __init__(self, s, d, theta, current_time_step: int, dd=None, ddd=None, ss=None, sss=None):
        super().__init__(current_time_step)
        self.s, self.d, self.theta = s, d, theta
        self.d_dot, self.d_ddot, self.s_dot, self.s_ddot = dd, ddd, ss, sss

    def length(self) -> int:
        return len(self.s)


class TrajectorySample(Sample):
    """One candidate: the two polynomials plus (after checking) its Cartesian / curvilinear states."""

    def __init__(self, horizon: float, dt: float, trajectory_long, trajectory_lat):
        self.horizon = horizon
        self.dt = dt
        self.trajectory_long = trajectory_long
        self.trajectory_lat = trajectory_lat
        self._cost = 0
        self._cost_function = None
        self.cartesian: Optional[CartesianSample] = None
        self.curvilinear: Optional[CurviLinearSample] = None
        self.feasibility_label: Optional[FeasibilityStatus] = None

    @property
    def cost(self) -> float:
        return self._cost

    @cost.setter
    def cost(self, cost_function):
        """``sample.cost = cost_function`` evaluates the plug-in, as the reference does (:397-404)."""
        self._cost = cost_function.evaluate(self)
        self._cost_function = cost_function

    def length(self) -> int:
        return self.cartesian.length()


class TrajectoryBundle:
    """Collection of samples with the reference's sort / filter helpers (trajectories.py:466-558)."""

    def __init__(self, trajectories: List[TrajectorySample], cost_function):
        self.trajectories = trajectories
        self._cost_function = cost_function
        self._is_sorted = False

    def sort(self):
        if not self._is_sorted:
            for t in self.trajectories:
                t.cost = self._cost_function
            self.trajectories.sort(key=lambda t: t.cost)   # stable: ties keep list order
            self._is_sorted = True

    def get_sorted_list(self) -> list:
        if not self._is_sorted:
            self.sort()
        return self.trajectories

    def min_costs(self):
        return self.trajectories[0] if self._is_sorted else None

    def max_costs(self):
        return self.trajectories[-1] if self._is_sorted else None

    def filter_goals_behind(self):
        self.trajectories = [t for t in self.trajectories if t.trajectory_long.x_0[0] < t.trajectory_long.x_d[0]]

    @property
    def empty(self) -> bool:
        return len(self.trajectories) == 0
