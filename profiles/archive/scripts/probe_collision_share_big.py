"""Share of the collision query in the production-mode evaluation kernel of the large obstacle-heavy workloads: kernel time with the
obstacle tables as they are, with RP_FLAG_SKIP_COLLISION (masks still computed by rp_lon_kernel) and without obstacles.
usage (GPU box): python profiles/probe_collision_share_big.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params
from commonroad_rp_amd.collision import ObstacleTables


def run(w, extra=0, obstacles=True, n=30):
    p = copy_params(w.inputs.params)
    p.flags |= extra
    inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
    ctx = RpContext(0)
    w.setup(ctx)
    if not obstacles:
        ctx.set_obstacles(ObstacleTables())
    ctx.set_profiling(1)
    ks = []
    for i in range(n + 5):
        out = ctx.plan(inp, want_best_states=False)
        if i >= 5:
            ks.append(out.kernel_ms)
    ctx.close()
    return 1e3 * float(np.median(ks)), out.n_feasible, out.n_collision


for name, w in (("cfg3", W.cfg3()), ("cfg5 + 50 obstacles", W.cfg5(obstacles=50)), ("cfg4", W.cfg4())):
    print(name, "| as they are %.0f us (feasible %d, colliding %d) | skip flag %.0f us | no obstacles %.0f us"
          % (run(w) + run(w, 4)[:1] + run(w, 0, False)[:1]))
