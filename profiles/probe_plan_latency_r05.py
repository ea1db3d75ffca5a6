"""ReactivePlanner.plan() in closed loop (production mode, Python included): p50 / p90 per workload with the cycle's call made by the
binding's extension module (_rpfast, the default), through ctypes (round 4), and with rp_plan_cycle allowed for Cartesian starts.
usage (GPU box): python profiles/probe_plan_latency_r05.py [cfg2 cfg1 cfg3]"""
import math, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.harness import run_closed_loop


class NoCycle(RpContext):
    plan_cycle = None


class CtypesCall(RpContext):
    """the cycle's call through ctypes (as without commonroad_rp_amd/_rpfast.so)"""
    plan_cycle = None

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.plan_packed_fast = None


def loop(w, factory, level_planner, n=260, warm=30):
    lat, loops = [], 0
    c = w.inputs.cost
    dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
    ctx = factory(0)

    class Shared:
        def __new__(cls, dev):
            return ctx
    while len(lat) < n + warm and loops < 40:
        rp = W.make_level_planner(w, t_min=0.2, backend_factory=Shared, device=0) if level_planner else W.make_planner(w, backend_factory=Shared, device=0)
        res = run_closed_loop(rp, max_steps=80 if not level_planner else 60, replanning_frequency=1, desired_velocity=None if level_planner else dv)
        lat += res.plan_times if res.completed else res.plan_times[:-1]
        loops += 1
    ctx.close()
    a = np.asarray(lat[warm:]) * 1e6
    return float(np.percentile(a, 50)), float(np.percentile(a, 90)), len(a)


for name in sys.argv[1:] or ["cfg2", "cfg1", "cfg3"]:
    level_planner = name == "cfg1"
    w = W.cfg1(level=1, road_boundary=True) if level_planner else W.WORKLOADS[name]()
    for rep in range(2):
        for tag, fac in (("extension call", NoCycle), ("ctypes call", CtypesCall), ("rp_plan_cycle", RpContext)):
            p50, p90, k = loop(w, fac, level_planner)
            print(f"{name:5s} {tag:15s} plan() p50 {p50:7.1f} us  p90 {p90:7.1f} us  ({k} replans)", flush=True)
