"""Evaluation-kernel time with the road boundary (static rectangles) in the obstacle tables, fused and draw+states."""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
for name in sys.argv[1:] or ["cfg2"]:
    for rb in (False, True):
        w = getattr(W, name)(road_boundary=rb)
        ctx = RpContext(0); w.setup(ctx); ctx.set_profiling(True)
        out = []
        for extra in (0, FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL):
            p = copy_params(w.inputs.params); p.flags |= extra
            inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
            for _ in range(3): o = ctx.plan(inp)
            ks = [ctx.plan(inp).kernel_ms for _ in range(20)]
            out.append(f"{'draw+states' if extra else 'fused'} {np.mean(ks)*1e3:8.1f} us (colliding {o.n_collision}, winner {o.best_index})")
        print(f"{name} road_boundary={rb} static shapes {len(w.obstacles.static_obb)}: " + " | ".join(out))
        ctx.close()
