"""ReactivePlanner.plan() in closed loop on a workload, split into the C call (rp_plan_packed: launch, device, ticket, packing) and the
Python around it.  The package is taken from <package root> (default: this repo's) so that two versions of the host glue can be held
against each other on one box; the library is the repo's either way.
usage (GPU box): python profiles/probe_plan_glue.py [workload] [package root]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
root = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(REPO, "commonroad-reactive-planner_amd")
os.environ.setdefault("RP_AMD_LIBRARY", os.path.join(REPO, "commonroad-reactive-planner_amd", "lib", "librp_amd.so"))
sys.path[:0] = [root]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.harness import run_closed_loop
W.SCENARIO_DIR = os.path.join(REPO, "tests", "golden")   # (a package root elsewhere still reads this repo's scenario files)
import math
base = W.WORKLOADS[name]()
ctx = RpContext(0)
inner = []
real = ctx._lib.rp_plan_packed


def timed(*a):
    t = time.perf_counter(); r = real(*a); inner.append(time.perf_counter() - t); return r


class Shared:
    def __new__(cls, dev):
        return ctx
c = base.inputs.cost
dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
for rep in range(3):
    lat, calls, loops = [], [], 0
    while len(lat) < 300 and loops < 40:
        rp = W.make_planner(base, backend_factory=Shared, device=0)
        ctx._lib.rp_plan_packed = timed
        if getattr(ctx, "_fast", None) is not None:
            ctx._fast = (ctx._fast[0], ctx._fast[1], timed)
        inner.clear()
        res = run_closed_loop(rp, max_steps=80, replanning_frequency=1, desired_velocity=dv)
        n = len(res.plan_times) if res.completed else len(res.plan_times) - 1
        lat += res.plan_times[:n]; calls += inner[:n]
        loops += 1
    lat, calls = np.array(lat[20:]), np.array(calls[20:len(lat)])
    m = min(len(lat), len(calls))
    print(f"{name} [{os.path.relpath(root, REPO)}] plan() p50 {np.median(lat) * 1e6:6.1f} us | C call p50 {np.median(calls) * 1e6:6.1f} us | "
          f"Python around it p50 {np.median(lat[:m] - calls[:m]) * 1e6:6.1f} us  ({len(lat)} cycles)")
