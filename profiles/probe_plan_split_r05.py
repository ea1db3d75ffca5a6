"""Where the Python of ReactivePlanner.plan() goes (closed loop, production mode): the fast path's pieces timed one by one through
wrappers (each costs ~0.15 us itself, so the pieces sum to a little more than the undisturbed plan()).
usage (GPU box): python profiles/probe_plan_split_r05.py [cfg2 cfg1]"""
import math, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.harness import run_closed_loop
from commonroad_rp_amd import reactive_planner as RPM
from commonroad_rp_amd import sampling as SM

ns = time.perf_counter_ns
acc = {}


def wrap(owner, name, tag):
    real = getattr(owner, name)

    def timed(*a, **k):
        t = ns()
        try:
            return real(*a, **k)
        finally:
            acc.setdefault(tag, []).append(ns() - t)
    setattr(owner, name, timed)
    return real


def run(name):
    level_planner = name == "cfg1"
    w = W.cfg1(level=1, road_boundary=True) if level_planner else W.WORKLOADS[name]()
    c = w.inputs.cost
    dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
    ctx = RpContext(0)

    class Shared:
        def __new__(cls, dev):
            return ctx
    for instrument in (False, True):
        acc.clear()
        undo = []
        if instrument:
            probe = W.make_planner(w, backend_factory=Shared, device=0)
            names = (("_fast_path_ok", "fast_path_ok"), ("_gpu_cost", "gpu_cost"), ("_gpu_params", "gpu_params"), ("_gpu_params_struct", "gpu_params_struct"),
                     ("_gpu_ctx", "gpu_ctx"), ("_reset_statistics", "reset_statistics"), ("_fast_output", "fast_output"), ("_plan_fast", "plan_fast (all)"))
            for fn, tag in names:
                owner = next((k for k in type(probe).__mro__ if fn in k.__dict__), None)   # (the class of the MRO that defines it)
                if owner is not None:
                    undo.append((owner, fn, wrap(owner, fn, tag)))
            for owner, fn, tag in ((SM.FixedIntervalSampling, "grids_at_level", "grids_at_level"), (RpContext, "plan_packed", "plan_packed (all)"),
                                   (RpContext, "plan_packed_fast", "plan_packed_fast (all)")):
                if hasattr(owner, fn):
                    undo.append((owner, fn, wrap(owner, fn, tag)))
        lat, loops = [], 0
        while len(lat) < 300 and loops < 40:
            rp = W.make_level_planner(w, t_min=0.2, backend_factory=Shared, device=0) if level_planner else W.make_planner(w, backend_factory=Shared, device=0)
            if instrument and getattr(ctx, "_fast", None) is not None and not getattr(ctx, "_probe_wrapped", False):
                real = ctx._fast[2]

                def ccall(*a, _real=real):
                    t = ns(); r = _real(*a); acc.setdefault("C call rp_plan", []).append(ns() - t); return r
                ctx._fast = (ctx._fast[0], ctx._fast[1], ccall)
                ctx._probe_wrapped = True
            res = run_closed_loop(rp, max_steps=80 if not level_planner else 60, replanning_frequency=1, desired_velocity=None if level_planner else dv)
            lat += res.plan_times if res.completed else res.plan_times[:-1]
            loops += 1
        a = np.asarray(lat[30:]) * 1e6
        print(f"{name} {'instrumented' if instrument else 'undisturbed '} plan() p50 {np.percentile(a, 50):6.1f} us  p90 {np.percentile(a, 90):6.1f} us ({len(a)} replans)")
        if instrument:
            for tag, v in sorted(acc.items(), key=lambda kv: -np.median(kv[1])):
                v = np.asarray(v[30:]) / 1e3
                print(f"    {tag:22s} p50 {np.median(v):6.2f} us  x {len(v) / max(1, len(a)):.2f} per plan")
        for owner, fn, real in undo:
            setattr(owner, fn, real)
    ctx.close()


for name in sys.argv[1:] or ["cfg2", "cfg1"]:
    run(name)
