"""cProfile of ReactivePlanner.plan() in closed loop on the GPU box (where the time of the host glue goes, device call included).
usage (GPU box): python profiles/probe_plan_cprofile.py [workload]"""
import cProfile, math, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.harness import run_closed_loop
base = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]()
ctx = RpContext(0)


class Shared:
    def __new__(cls, dev):
        return ctx
c = base.inputs.cost
dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
run_closed_loop(W.make_planner(base, backend_factory=Shared, device=0), max_steps=80, replanning_frequency=1, desired_velocity=dv)
pr = cProfile.Profile()
for _ in range(4):
    rp = W.make_planner(base, backend_factory=Shared, device=0)
    pr.enable()
    run_closed_loop(rp, max_steps=80, replanning_frequency=1, desired_velocity=dv)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
