"""Cost plug-in surface (reference: commonroad_rp/cost_function.py:17-32 CostFunction ABC,
:35-71 DefaultCostFunction, :74-92 DefaultCostFunctionFailSafe).

The two default cost functions are *fused into the HIP kernel*: when the planner's cost function is
exactly one of these types, ``evaluate`` below is never called -- the kernel computes the same sums
per candidate.  ``evaluate`` exists for user subclasses (which run through the plug-in path: states
materialised on the device, ``evaluate`` called per feasible sample in Python, argmin on the device)
and so that ``super().evaluate(...)`` keeps working in such subclasses.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import numpy as np


class CostFunction(ABC):
    @abstractmethod
    def evaluate(self, trajectory) -> float:
        """Cost of one TrajectorySample (states over the extended horizon)."""


class DefaultCostFunction(CostFunction):
    def __init__(self, desired_speed: Optional[float] = None, desired_d: float = 0.0, desired_s: Optional[float] = None):
        self.desired_speed = desired_speed
        self.desired_d = desired_d
        self.desired_s = desired_s
        self.w_a = 5

    def evaluate(self, trajectory) -> float:
        ca, cu = trajectory.cartesian, trajectory.curvilinear
        costs = 0.0
        costs += np.sum((self.w_a * ca.a) ** 2)
        if self.desired_speed is not None:
            costs += np.sum((5 * (ca.v - self.desired_speed)) ** 2) + (50 * (ca.v[-1] - self.desired_speed) ** 2) + \
                     (100 * (ca.v[int(len(ca.v) / 2)] - self.desired_speed) ** 2)
        if self.desired_s is not None:
            costs += np.sum((0.25 * (self.desired_s - cu.s)) ** 2) + (20 * (self.desired_s - cu.s[-1])) ** 2
        costs += np.sum((0.25 * (self.desired_d - cu.d)) ** 2) + (20 * (self.desired_d - cu.d[-1])) ** 2
        costs += np.sum((0.25 * np.abs(cu.theta)) ** 2) + (5 * (np.abs(cu.theta[-1]))) ** 2
        return costs


class DefaultCostFunctionFailSafe(CostFunction):
    def evaluate(self, trajectory) -> float:
        ca, cu = trajectory.cartesian, trajectory.curvilinear
        costs = np.sum((1 * ca.a) ** 2)
        costs += np.sum((0.25 * cu.d) ** 2) + (20 * cu.d[-1]) ** 2
        costs += np.sum((0.25 * np.abs(cu.theta)) ** 2) + (5 * (np.abs(cu.theta[-1]))) ** 2
        return costs
