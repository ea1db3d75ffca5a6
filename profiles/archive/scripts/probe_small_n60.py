"""Probe: small batches at the reference's default horizon (N = 60): step time per lanes-per-candidate choice."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
w = W.WORKLOADS["cfg3"]()
for nT, nL, nD in ((5, 9, 9), (9, 17, 17), (15, 20, 20), (20, 25, 25), (25, 30, 30), (31, 36, 36)):
    T = w.inputs.T[:: max(1, len(w.inputs.T) // nT)][:nT]; tl = w.inputs.traj_len[:: max(1, len(w.inputs.T) // nT)][:nT]
    L = np.linspace(w.inputs.L.min(), w.inputs.L.max(), nL); D = np.linspace(-3, 3, nD)
    for mode, fl in (("production", 0), ("draw", FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)):
        p = copy_params(w.inputs.params); p.flags |= fl
        inp = PlanInputs(p, w.inputs.cost, T, tl, L, D)
        for g in ("default", "16", "32", "64"):
            os.environ.pop("RP_AMD_G", None)
            if g != "default": os.environ["RP_AMD_G"] = g
            ctx = RpContext(0); w.setup(ctx)
            for _ in range(10): out = ctx.plan(inp)
            t0 = time.perf_counter()
            for _ in range(100): ctx.plan(inp)
            dt = (time.perf_counter() - t0) / 100
            print(f"C={inp.n_candidates:5d} N=60 {mode:10s} G={g:7s} step {dt * 1e6:7.1f} us  winner {out.best_index}")
            ctx.close()
