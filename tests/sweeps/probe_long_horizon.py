import sys, math
sys.path[:0] = ["/root/repo", "/root/repo/commonroad-reactive-planner_amd", "/root/repo/tests"]
import numpy as np
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, make_params, make_cost, copy_params, FLAG_MATERIALIZE_ALL, COST_DEFAULT, LON_VELOCITY_KEEPING
from commonroad_rp_amd.collision import ObstacleTables
from commonroad_rp_amd.coordinate_system import CoordinateSystem
from oracle import oracle
from _paths import LAUNCH_PATHS, launch_path_env
s = np.arange(0.0, 3000.0, 1.0)
co = CoordinateSystem(np.stack((s, 30.0 * np.sin(s / 200.0)), 1))
for N in (300, 1000, 2500, 4094):
    dt = 0.05
    T = np.array([dt * k for k in (N // 3, N // 2, N)])
    tl = W.traj_len_of(T, dt)
    p = make_params(dt=dt, N=N, factor=1, time_step0=0, low_vel_mode=False, lon_mode=LON_VELOCITY_KEEPING, constraint_mask=31, flags=0,
                    x0_lon=[20.0, 12.0, 0.0], x0_lat=[0.3, 0.0, 0.0], x0_orientation=0.15, **W.VEHICLE2)
    cost = make_cost(COST_DEFAULT, w_a=1.0, desired_speed=12.0, desired_d=0.0, desired_s=None)
    inp = PlanInputs(p, cost, T, tl, np.linspace(8, 14, 5), np.linspace(-2, 2, 5))
    dyn = np.full((2, N + 1, 5), np.nan)
    for q in range(N + 1):
        dyn[0, q] = (60 + 10.0 * q * dt, 3.0 + 30.0 * math.sin((60 + 10.0 * q * dt) / 200.0), 0.1, 2.2, 0.9)
    obs = ObstacleTables(dyn_obb=dyn, dyn_t0=0)
    tb = oracle.OracleTables.from_coordinate_system(co, obs)
    ref = oracle.plan(inp, tb, want_states=True)
    for path in LAUNCH_PATHS:
        with launch_path_env(path):
            ctx = RpContext(0); ctx.set_coordinate_system(co); ctx.set_obstacles(obs)
            p2 = copy_params(p); p2.flags |= FLAG_MATERIALIZE_ALL
            out = ctx.plan(PlanInputs(p2, cost, T, tl, inp.L, inp.D))
            st, cs = ctx.fetch_status(); states = ctx.fetch_states()
            ok = np.array_equal(st & 0xff, ref.status & 0xff) and out.best_index == ref.out.best_index
            have = ((ref.status & 3) == 1) | ((ref.status & 3) == 3)
            dev = np.max(np.abs(states[have] - ref.states[have])) if have.any() else 0.0
            print(N, path, "labels/winner ok" if ok else "MISMATCH", "feasible", int(have.sum()), "winner", out.best_index, "max state dev %.3g" % dev)
            ctx.close()
