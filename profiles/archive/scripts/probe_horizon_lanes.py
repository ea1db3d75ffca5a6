"""Lanes per candidate with state rows, large batch (two-kernel path), as a function of the horizon: evaluation-kernel time on a
synthetic curved route, ~200 000 candidates, no obstacles, under RP_AMD_G of the environment.
usage (GPU box): for g in 16 64; do RP_AMD_G=$g python profiles/probe_horizon_lanes.py; done"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd._capi import RpContext, PlanInputs, make_params, make_cost, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
from commonroad_rp_amd.coordinate_system import CoordinateSystem
from commonroad_rp_amd.collision import ObstacleTables
from commonroad_rp_amd.workloads import VEHICLE2, traj_len_of

s = np.arange(0.0, 600.0, 1.0)
co = CoordinateSystem(np.stack((s, 25.0 * np.sin(s / 70.0)), axis=1))
dt = 0.1
res = []
for N in (40, 60, 64, 72, 80, 90, 100):
    T = np.array([dt * k for k in range(N // 2, N + 1)])
    nT = len(T)
    nL = 63
    nD = max(3, int(200000 / (nT * nL)))
    L, D = np.linspace(6.0, 14.0, nL), np.linspace(-3.0, 3.0, nD)
    s0 = co.ref_pos[40]
    p = make_params(dt=dt, N=N, x0_lon=[s0, 10.0, 0.1], x0_lat=[0.2, 0.0, 0.0], x0_orientation=float(co.ref_theta[40]), **VEHICLE2)
    p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
    inp = PlanInputs(p, make_cost(desired_speed=10.0), T, traj_len_of(T, dt), L, D)
    ctx = RpContext(0)
    ctx.set_coordinate_system(co); ctx.set_obstacles(ObstacleTables())
    ctx.set_profiling(1)
    ks = []
    for i in range(25):
        out = ctx.plan(inp)
        if i >= 5:
            ks.append(out.kernel_ms)
    ctx.close()
    C = inp.n_candidates
    k = float(np.median(ks))
    res.append("N=%d C=%d: %.0f us (%.0f %% of 8 TB/s)" % (N, C, 1e3 * k, 100 * C * (12 + 112 * (N + 1)) / (k * 1e-3) / 8e12))
print("G=%s | " % os.environ.get("RP_AMD_G", "default") + " | ".join(res))
