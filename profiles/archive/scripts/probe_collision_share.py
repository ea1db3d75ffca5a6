"""How much of the cfg2 evaluation kernel is the collision query?  Kernel time (HIP events, every step) of the draw-mode plan with
the obstacle tables as they are, with RP_FLAG_SKIP_COLLISION, and with no dynamic obstacles.  usage (GPU box): python profiles/probe_collision_share.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL

FLAG_SKIP = 4


def run(w, extra_flags=0, obstacles=True, n=200):
    p = copy_params(w.inputs.params)
    p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL | extra_flags
    inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
    ctx = RpContext(0)
    w.setup(ctx)
    if not obstacles:
        from commonroad_rp_amd.collision import ObstacleTables
        ctx.set_obstacles(ObstacleTables())
    ctx.set_profiling(1)
    ks, coll = [], 0
    for i in range(n + 20):
        out = ctx.plan(inp)
        if i >= 20:
            ks.append(out.kernel_ms)
        coll = out.n_collision
    return 1e3 * float(np.median(ks)), coll


w = W.cfg2()
for rep in range(2):
    print("as they are          %.2f us (colliding candidates %d)" % run(w))
    print("skip collision flag  %.2f us (%d)" % run(w, FLAG_SKIP))
    print("no obstacles         %.2f us (%d)" % run(w, 0, False))
