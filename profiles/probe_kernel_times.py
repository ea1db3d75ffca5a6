"""Kernel time (HIP events on the launch) and step time of the batch's evaluation kernel for the production-mode plans of a few workloads
under the library's own launch policy -- the regression check of a change to a kernel.   usage (GPU box): python profiles/probe_kernel_times.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W, _capi
from commonroad_rp_amd._capi import RpContext, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL, PlanInputs, copy_params
print("library", _capi.source_hash())
for name, lazy in (("cfg3", 0), ("cfg3", -1), ("cfg3f", -1), ("cfg4", -1), ("cfg4rb", -1), ("cfg5", -1), ("cfg3", "draw"), ("cfg5", "draw")):
    w = W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]()
    ctx = RpContext(0); w.setup(ctx)
    p = copy_params(w.inputs.params)
    if lazy == "draw":
        p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
    else:
        ctx.set_option("lazy", lazy)
    inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
    for _ in range(8):
        ctx.plan(inp)
    ctx.set_profiling(1)
    kms = [ctx.plan(inp).kernel_ms for _ in range(30)]
    ctx.set_profiling(0)
    t0 = time.perf_counter()
    for _ in range(60):
        ctx.plan(inp)
    dt = (time.perf_counter() - t0) / 60
    print(f"{name:7s} {'draw' if lazy == 'draw' else ('eager' if lazy == 0 else 'production'):10s} kernel [{ctx.last_kernel()[3:-7]:5s}] {np.median(kms)*1e3:8.1f} us  step {dt*1e6:8.1f} us  path {ctx.last_path()}", flush=True)
    ctx.close()
