"""A fixed number of plans of one workload under one launch policy: the program the counter passes (rocprofv3 --pmc) run.
usage: python3 profiles/run_plans.py <workload> <nocoll|eager|prod> <eval16|lane|chunk|auto> [plans]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params

name, mode, kern = sys.argv[1], sys.argv[2], sys.argv[3]
plans = int(sys.argv[4]) if len(sys.argv) > 4 else 10
w = W.cfg5(obstacles=50) if name == "cfg5obs" else (W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]())
ctx = RpContext(0)
w.setup(ctx)
ctx.set_option("fused_lon", 0)
ctx.set_option("auto_materialize", 0)
for k, v in {"eval16": {"chunk_kernel": 0, "cost_kernel": 0}, "lane": {"chunk_kernel": 0, "cost_kernel": 1}, "chunk": {"chunk_kernel": 1, "cost_kernel": 0}, "auto": {}}[kern].items():
    ctx.set_option(k, v)
p = copy_params(w.inputs.params)
if mode == "nocoll":
    p.flags |= FLAG_SKIP_COLLISION
if mode == "eager":
    ctx.set_option("lazy", 0)
inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
for _ in range(plans):
    out = ctx.plan(inp, want_best_states=False)
print(name, mode, kern, "winner", out.best_index, "kernel", ctx.last_kernel(), "path", ctx.last_path())
ctx.close()
