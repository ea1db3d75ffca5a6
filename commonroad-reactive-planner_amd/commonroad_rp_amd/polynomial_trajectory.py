"""Quartic / quintic polynomial trajectories of the sampling plug-in surface.

Interface of ``commonroad_rp.polynomial_trajectory`` (reference: polynomial_trajectory.py:17-271
PolynomialTrajectory, :274-320 QuinticTrajectory, :323-360 QuarticTrajectory).  On the hot path the
coefficients are computed on the device for the whole grid; these classes carry the winner's
coefficients back (``coeffs=`` argument) and serve plug-ins that build single polynomials.
"""
from __future__ import annotations

import numpy as np


class PolynomialTrajectory:
    def __init__(self, tau_0=0, delta_tau=0, x_0=np.zeros(3), x_d=np.zeros(3), power=5, coeffs=None):
        assert tau_0 >= 0, f"<PolynomialTrajectory/tau_0>: tau_0 not valid! tau_0={tau_0}"
        assert delta_tau > 0, f"<PolynomialTrajectory/delta_tau>: delta_tau not valid! delta_tau={delta_tau}"
        assert power in (4, 5)
        self.tau_0, self.delta_tau = tau_0, delta_tau
        self.x_0, self.x_d = np.asarray(x_0, dtype=float), np.asarray(x_d, dtype=float)
        self.power = power
        self._cost = None
        c = np.asarray(coeffs, dtype=float) if coeffs is not None else self.calc_coeffs()
        assert isinstance(c, np.ndarray) and len(c) == 6, f"<PolynomialTrajectory/coeffs>: coeffs length not valid! length={len(c)}"
        self.coeffs = c
        self._db = {}

    def calc_coeffs(self) -> np.ndarray:
        raise NotImplementedError

    @property
    def cost(self):
        return self._cost

    @cost.setter
    def cost(self, cost):
        assert cost >= 0
        self._cost = cost

    # --- evaluation (polynomial_trajectory.py:171-271) -------------------------------------------
    def calc_position(self, tau, tau2, tau3, tau4, tau5):
        c = self.coeffs
        return c[0] + c[1] * tau + c[2] * tau2 + c[3] * tau3 + c[4] * tau4 + c[5] * tau5

    def calc_velocity(self, tau, tau2, tau3, tau4):
        c = self.coeffs
        return c[1] + 2. * c[2] * tau + 3. * c[3] * tau2 + 4. * c[4] * tau3 + 5. * c[5] * tau4

    def calc_acceleration(self, tau, tau2, tau3):
        c = self.coeffs
        return 2 * c[2] + 6 * c[3] * tau + 12 * c[4] * tau2 + 20 * c[5] * tau3

    def calc_jerk(self, tau, tau2):
        c = self.coeffs
        return 6 * c[3] + 24 * c[4] * tau + 60 * c[5] * tau2

    def evaluate_state_at_tau(self, tau: float):
        if tau in self._db:
            return self._db[tau]
        key = tau
        tp = tau - self.tau_0
        if tp < 0:
            tau = self.tau_0
        elif tp > self.delta_tau:
            tau = self.delta_tau
        t2 = tau * tau
        t3, t4 = t2 * tau, t2 * t2
        t5 = t3 * t2
        res = np.array([self.calc_position(tau, t2, t3, t4, t5), self.calc_velocity(tau, t2, t3, t4),
                        self.calc_acceleration(tau, t2, t3)])
        self._db[key] = res
        return res

    def squared_jerk_integral(self, t):
        c = self.coeffs
        t2 = t * t
        t3, t4, t5 = t2 * t, t2 * t2, t2 * t2 * t
        return (36 * c[3] * c[3] * t + 144 * c[3] * c[4] * t2 + 240 * c[3] * c[5] * t3 + 192 * c[4] * c[4] * t3 +
                720 * c[4] * c[5] * t4 + 720 * c[5] * c[5] * t5)


class QuinticTrajectory(PolynomialTrajectory):
    """x_0 = [p, p', p''], x_d = [p, p', p''] at delta_tau (polynomial_trajectory.py:274-320)."""

    def __init__(self, tau_0=0, delta_tau=0, x_0=np.zeros(3), x_d=np.zeros(3), coeffs=None):
        super().__init__(tau_0=tau_0, delta_tau=delta_tau, x_0=x_0, x_d=x_d, power=5, coeffs=coeffs)

    def calc_coeffs(self) -> np.ndarray:
        p0, v0, a0 = self.x_0
        pf, vf, af = self.x_d
        T = self.delta_tau
        t2 = T * T
        t3, t4 = t2 * T, t2 * t2
        t5 = t4 * T
        A = np.array([[t3, t4, t5], [3. * t2, 4. * t3, 5. * t4], [6. * T, 12. * t2, 20. * t3]])
        b = np.array([pf - (p0 + v0 * T + .5 * a0 * t2), vf - (v0 + a0 * T), af - a0])
        x = np.linalg.solve(A, b)
        return np.array([p0, v0, .5 * a0, x[0], x[1], x[2]])


class QuarticTrajectory(PolynomialTrajectory):
    """x_0 = [p, p', p''], x_d = [p', p''] (velocity keeping; polynomial_trajectory.py:323-360)."""

    def __init__(self, tau_0=0, delta_tau=0, x_0=np.zeros(3), x_d=np.zeros(2), coeffs=None):
        self._desired_velocity = x_d[0]
        super().__init__(tau_0=tau_0, delta_tau=delta_tau, x_0=x_0, x_d=x_d, power=4, coeffs=coeffs)

    def calc_coeffs(self) -> np.ndarray:
        p0, v0, a0 = self.x_0
        T = self.delta_tau
        t2 = T * T
        t3 = t2 * T
        A = np.array([[3. * t2, 4. * t3], [6. * T, 12. * t2]])
        b = np.array([self._desired_velocity - v0 - a0 * T, -a0])
        x = np.linalg.solve(A, b)
        return np.array([p0, v0, .5 * a0, x[0], x[1], 0.])
