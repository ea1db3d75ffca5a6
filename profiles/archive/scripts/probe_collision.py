"""Evaluation-kernel time with and without the collision query (FLAG_SKIP_COLLISION), draw + states mode."""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL, FLAG_SKIP_COLLISION
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = W.WORKLOADS[name]()
ctx = RpContext(0); w.setup(ctx); ctx.set_profiling(True)
for skip in (0, 1):
    for mat in (0, 1):
        p = copy_params(w.inputs.params); p.flags |= FLAG_DRAW_ALL | (FLAG_MATERIALIZE_ALL if mat else 0) | (FLAG_SKIP_COLLISION if skip else 0)
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        for _ in range(3): ctx.plan(inp)
        ks = [ctx.plan(inp).kernel_ms for _ in range(20)]
        print(f"{name} skip_collision={skip} states={mat}: {np.mean(ks)*1e3:8.1f} us")
