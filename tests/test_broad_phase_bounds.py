"""The analytic bounds behind the collision broad phase (csrc/rp_kernels.h: pair_step_bound), checked numerically on
the CPU with the host-side polynomial classes:

  lateral quintic from (d0, d0', d0'') to (d1, 0, 0) over [0, T]:
      |d(tau)| <= max(|d0|, |d1|) + 0.2 |d0'| T + 0.0173 |d0''| T^2          (Hermite basis)
"""
import numpy as np

from commonroad_rp_amd.polynomial_trajectory import QuinticTrajectory


def test_lateral_quintic_overshoot_bound():
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(4000):
        T = rng.uniform(0.2, 12.0)
        d0, d1 = rng.uniform(-4, 4, 2)
        v0 = rng.uniform(-3, 3) * rng.choice([0.0, 1.0, 1.0])
        a0 = rng.uniform(-6, 6) * rng.choice([0.0, 1.0, 1.0])
        q = QuinticTrajectory(tau_0=0, delta_tau=T, x_0=np.array([d0, v0, a0]), x_d=np.array([d1, 0.0, 0.0]))
        tau = np.linspace(0.0, T, 400)
        c = q.coeffs
        d = c[0] + c[1] * tau + c[2] * tau ** 2 + c[3] * tau ** 3 + c[4] * tau ** 4 + c[5] * tau ** 5
        bound = max(abs(d0), abs(d1)) + 0.2 * abs(v0) * T + 0.0173 * abs(a0) * T * T
        assert np.max(np.abs(d)) <= bound * (1 + 1e-9) + 1e-9, (T, d0, d1, v0, a0)
        if abs(v0) * T + abs(a0) * T * T > 1e-6:
            worst = max(worst, (np.max(np.abs(d)) - max(abs(d0), abs(d1))) / (0.2 * abs(v0) * T + 0.0173 * abs(a0) * T * T))
    assert 0.5 < worst <= 1.0 + 1e-9      # the bound is not only valid but reasonably tight


def test_hermite_basis_extrema():
    u = np.linspace(0.0, 1.0, 200001)
    A = 1 - 10 * u ** 3 + 15 * u ** 4 - 6 * u ** 5
    B = u - 6 * u ** 3 + 8 * u ** 4 - 3 * u ** 5
    Cc = 0.5 * u ** 2 - 1.5 * u ** 3 + 1.5 * u ** 4 - 0.5 * u ** 5
    assert A.min() >= -1e-12 and A.max() <= 1 + 1e-12
    assert 0.19 < np.abs(B).max() <= 0.2
    assert 0.017 < np.abs(Cc).max() <= 0.0173
