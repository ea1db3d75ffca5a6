"""Small batch under dense traffic: cfg2's grid and route with n obstacles driving on and beside the ego's route (most candidates have
obstacles inside their (pair, step) masks, many collide) instead of the scenario's five distant ones.  Evaluation-kernel time, draw mode.
usage (GPU box): python profiles/probe_dense_traffic.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
from commonroad_rp_amd.collision import ObstacleTables


def traffic(co, s_ego, n_dyn, n_steps, seed=3, dt=0.1):
    rng = np.random.default_rng(seed)
    dyn = np.full((n_dyn, n_steps, 5), np.nan)
    for j in range(n_dyn):
        s0, vel, off = s_ego + rng.uniform(8.0, 45.0), rng.uniform(2.0, 7.0), rng.uniform(-3.5, 3.5)
        for k in range(n_steps):
            s = s0 + vel * dt * k
            if not (1.0 < s < co.ref_pos[-1] - 2.0):
                continue
            x, y = co.convert_to_cartesian_coords(s, off)
            kk = min(int(np.searchsorted(co.ref_pos, s, side="right")) - 1, len(co.ref_pos) - 2)
            dyn[j, k] = (x, y, co.ref_theta[kk], 2.25, 0.9)
    return dyn


w = W.cfg2()
co = w.coordinate_system
p = copy_params(w.inputs.params)
p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
for n_dyn in (0, 5, 10, 20, 40):
    ctx = RpContext(0)
    w.setup(ctx)
    tables = ObstacleTables(dyn_obb=traffic(co, float(p.x0_lon[0]), n_dyn, 60), dyn_t0=int(p.time_step0)) if n_dyn else ObstacleTables()
    ctx.set_obstacles(tables)
    ctx.set_profiling(1)
    ks = []
    for i in range(120):
        out = ctx.plan(inp)
        if i >= 20:
            ks.append(out.kernel_ms)
    ctx.close()
    print("%2d obstacles near the route: kernel %.1f us, colliding %d of %d feasible" % (n_dyn, 1e3 * float(np.median(ks)), out.n_collision, out.n_feasible))
