import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/commonroad-reactive-planner_amd"]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
w = W.WORKLOADS["cfg2"]()
ctx = RpContext(0); w.setup(ctx)
out = ctx.plan(w.inputs)
print("n_feasible", out.n_feasible, "n_collision", out.n_collision, "before", out.n_collision_before_best, "best", out.best_index)
for _ in range(20): ctx.count_collisions_before(out.best_cost, out.best_index)
t = time.perf_counter()
for _ in range(200): ctx.count_collisions_before(out.best_cost, out.best_index)
print("count_collisions_before us", (time.perf_counter() - t) / 200 * 1e6)
