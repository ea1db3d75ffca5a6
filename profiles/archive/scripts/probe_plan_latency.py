"""Probe: ReactivePlanner.plan() latency on the plan-level fixtures and the cfg1 sampling levels (looking for slow paths)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")]
from _golden import build_planner_from_plan_golden, plan_case_names
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd import workloads as W
for name in plan_case_names():
    rp, z = build_planner_from_plan_golden(name, RpContext)
    for _ in range(10): res = rp.plan()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); res = rp.plan(); ts.append(time.perf_counter() - t0)
    print(f"{name:24s} planned {res is not None!s:5s} p50 {np.median(ts) * 1e6:7.1f} us  p90 {np.percentile(ts, 90) * 1e6:7.1f} us")
    rp.close()
for level in (1, 2, 3):
    w = W.cfg1(level)
    ctx = RpContext(0); w.setup(ctx)
    for _ in range(10): out = ctx.plan(w.inputs)
    t0 = time.perf_counter()
    for _ in range(200): out = ctx.plan(w.inputs)
    print(f"cfg1 level {level}: C={w.n_candidates:5d} raw plan step {(time.perf_counter() - t0) / 200 * 1e6:6.1f} us  winner {out.best_index} n_coll {out.n_collision}")
    ctx.close()
