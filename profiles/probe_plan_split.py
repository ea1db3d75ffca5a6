"""Where a closed-loop ReactivePlanner.plan() call spends its time on the GPU box: the whole call, the part inside
RpContext.plan (ctypes + rp_plan + result object) and the part inside rp_plan itself (C: launches + wait).
usage (GPU box): python profiles/probe_plan_split.py [cfg2]"""
import math, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
import ctypes as C

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]()
t_ctx, t_c = [], []


class Timed(RpContext):
    def plan(self, inp, *a, **k):
        t0 = time.perf_counter()
        out = super().plan(inp, *a, **k)
        t_ctx.append(time.perf_counter() - t0)
        return out

    def plan_packed(self, *a, **k):   # (what the stand-alone planner's cycle calls: one C call per level)
        t0 = time.perf_counter()
        out = super().plan_packed(*a, **k)
        t_ctx.append(time.perf_counter() - t0)
        return out


rp = W.make_planner(w, backend_factory=Timed, device=0)
lib = None
c = w.inputs.cost
dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
lat = []
x0 = rp.x_0
for cycle in range(260):
    rp.set_desired_velocity(desired_velocity=dv, current_speed=rp.x_0.velocity)
    t0 = time.perf_counter()
    res = rp.plan()
    lat.append(time.perf_counter() - t0)
    if res is None:
        break
    nxt = res[0].state_list[1]
    rp.reset(initial_state_cart=nxt, initial_state_curv=(res[2][1], res[3][1]), collision_checker=None, coordinate_system=None) if False else None
lat, t_ctx = np.array(lat[20:]) * 1e6, np.array(t_ctx[20:]) * 1e6
print("plan() p50 %.1f us | inside RpContext.plan p50 %.1f us | rest (Python around it) p50 %.1f us" % (np.median(lat), np.median(t_ctx), np.median(lat - t_ctx[:len(lat)])))
