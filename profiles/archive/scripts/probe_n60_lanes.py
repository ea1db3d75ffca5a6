"""Lanes per candidate for draw-mode batches of 33-64 steps beyond the single-launch range (two-kernel path): evaluation-kernel
time of cfg3 (N = 60) -- whole batch, a 20 000-candidate slice, without obstacles -- under RP_AMD_G of the environment.
usage (GPU box): for g in 16 64; do RP_AMD_G=$g python profiles/probe_n60_lanes.py; done"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
from commonroad_rp_amd.collision import ObstacleTables


def run(w, lo, hi, obstacles=True, n=40):
    p = copy_params(w.inputs.params)
    p.flags |= FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
    inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
    ctx = RpContext(0)
    w.setup(ctx)
    if not obstacles:
        ctx.set_obstacles(ObstacleTables())
    ctx.set_profiling(1)
    ks = []
    for i in range(n + 5):
        out = ctx.plan(inp, lo, hi)
        if i >= 5:
            ks.append(out.kernel_ms)
    ctx.close()
    return 1e3 * float(np.median(ks))


w = W.cfg3()
C = w.inputs.n_candidates
g = os.environ.get("RP_AMD_G", "default")
print("G=%s  whole batch %.1f us | 20 000 candidates %.1f us | 17 000 %.1f us | whole batch without obstacles %.1f us"
      % (g, run(w, 0, C), run(w, 20000, 40000), run(w, 0, 17000), run(w, 0, C, obstacles=False)))
