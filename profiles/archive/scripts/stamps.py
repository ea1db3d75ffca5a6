"""Diagnostic: print the in-kernel s_memtime stamps of the stamped build (block 0 / wave 0).
Usage (GPU box): RP_AMD_LIBRARY=.../librp_amd_stamps.so RP_AMD_PRINT_STAMPS=1 python profiles/stamps.py cfg2"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = W.cfg2(road_boundary=True) if name == "cfg2rb" else W.WORKLOADS[name]()
p = copy_params(w.inputs.params); p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
ctx = RpContext(0); w.setup(ctx)
for _ in range(4):
    ctx.plan(inp)
