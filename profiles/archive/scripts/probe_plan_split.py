"""Where the Python time of ReactivePlanner.plan() goes on a GPU box: timers around the pieces of the fast path (each timer adds
~0.3 us to what it wraps; read the split, not the sum).   usage (GPU box): python profiles/probe_plan_split.py [workload]"""
import math, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.harness import run_closed_loop
base = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]()
ctx = RpContext(0)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc.setdefault(name, []).append(time.perf_counter() - t)
        return r
    return w


class Shared:
    def __new__(cls, dev):
        return ctx
c = base.inputs.cost
dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
run_closed_loop(W.make_planner(base, backend_factory=Shared, device=0), max_steps=80, replanning_frequency=1, desired_velocity=dv)
real_c = ctx._lib.rp_plan_packed
ctx._lib.rp_plan_packed = timed("C call rp_plan_packed", real_c)
if getattr(ctx, "_fast", None) is not None:
    ctx._fast = (ctx._fast[0], ctx._fast[1], ctx._lib.rp_plan_packed)
ctx.plan_packed = timed("binding plan_packed (C call inside)", ctx.plan_packed)
plan_times = []
for _ in range(4):
    rp = W.make_planner(base, backend_factory=Shared, device=0)
    rp._plan_fast = timed("_plan_fast (everything below inside)", rp._plan_fast)
    rp._gpu_params = timed("_gpu_params", rp._gpu_params)
    rp._fast_output = timed("_fast_output", rp._fast_output)
    rp._reset_statistics = timed("_reset_statistics", rp._reset_statistics)
    sp = rp.sampling_space
    sp.grids_at_level = timed("grids_at_level", sp.grids_at_level)
    res = run_closed_loop(rp, max_steps=80, replanning_frequency=1, desired_velocity=dv)
    plan_times += res.plan_times
print(f"plan() p50 {np.median(plan_times) * 1e6:.1f} us over {len(plan_times)} cycles (with the timers)")
for k, v in acc.items():
    print(f"  {k:44s} p50 {np.median(v) * 1e6:6.2f} us  (n = {len(v)})")
