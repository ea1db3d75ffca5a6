"""Kernel time of rp_eval_kernel for the four (draw, materialise) combinations.
usage: [RP_AMD_LIBRARY=...] python profiles/probe_modes.py cfg2|cfg5 [steps]"""
import sys, os, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = W.WORKLOADS[name]()
ctx = RpContext(0); w.setup(ctx); ctx.set_profiling(True)
out = []
for draw in (0, 1):
    for mat in (0, 1):
        p = copy_params(w.inputs.params); p.flags |= (FLAG_DRAW_ALL if draw else 0) | (FLAG_MATERIALIZE_ALL if mat else 0)
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        for _ in range(3): ctx.plan(inp)
        ks = [ctx.plan(inp).kernel_ms for _ in range(steps)]
        out.append(f"draw={draw} mat={mat}: {np.mean(ks)*1e3:9.1f} us")
print(os.path.basename(os.environ.get("RP_AMD_LIBRARY", "librp_amd.so")), name, " | ".join(out))
