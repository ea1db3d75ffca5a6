import sys, os, time
sys.path[:0] = ["/root/repo", "/root/repo/commonroad-reactive-planner_amd"]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
w = W.WORKLOADS["cfg3"]()
ctx = RpContext(0); w.setup(ctx)
for _ in range(3): out = ctx.plan(w.inputs)
t=time.perf_counter(); out = ctx.plan(w.inputs); dt=time.perf_counter()-t
print("best", out.best_index, out.best_cost, "feasible", out.n_feasible, "coll", out.n_collision, "before", out.n_collision_before_best, "ms", dt*1e3)
