"""Lanes per candidate with state rows as a function of the batch size at N = 100 and N = 80: whole step (rp_plan) in us, under RP_AMD_G of the
environment.  usage (GPU box): for g in 16 64; do RP_AMD_G=$g python profiles/probe_lanes_by_count.py; done"""
import os, sys, time
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
for N in (100, 80):
    res = []
    for target in (500, 2000, 8000, 16000, 30000, 60000):
        T = np.array([dt * k for k in range(N - 14, N + 1)])
        nT, nD = len(T), 15
        nL = max(2, target // (nT * nD))
        L, D = np.linspace(6.0, 14.0, nL), np.linspace(-3.0, 3.0, nD)
        p = make_params(dt=dt, N=N, x0_lon=[co.ref_pos[40], 10.0, 0.1], x0_lat=[0.2, 0.0, 0.0], x0_orientation=float(co.ref_theta[40]), **VEHICLE2)
        p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
        inp = PlanInputs(p, make_cost(desired_speed=10.0), T, traj_len_of(T, dt), L, D)
        ctx = RpContext(0)
        ctx.set_coordinate_system(co); ctx.set_obstacles(ObstacleTables())
        for i in range(10):
            ctx.plan(inp)
        t0 = time.perf_counter()
        for i in range(50):
            ctx.plan(inp)
        el = (time.perf_counter() - t0) / 50
        ctx.close()
        res.append("%d: %.0f" % (inp.n_candidates, el * 1e6))
    print("G=%s N=%d step us | " % (os.environ.get("RP_AMD_G", "default"), N) + " | ".join(res))
