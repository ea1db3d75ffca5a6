"""Costs-only launches without the collision query: which kernel from how many candidates on?  rp_eval_kernel (16 lanes per candidate),
rp_cost_kernel (one lane per candidate), rp_chunk_kernel (one lane per candidate and step block) on slices of cfg3's grid (N = 60),
cfg4's (N = 100) and cfg2's (N = 30), two-kernel path.   usage (GPU box): python profiles/probe_chunk_threshold.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params
KERNELS = {"eval16": {"chunk_kernel": 0, "cost_kernel": 0}, "lane": {"chunk_kernel": 0, "cost_kernel": 1}, "chunk": {"chunk_kernel": 1, "cost_kernel": 0}}
for name, nTs, reps in (("cfg3", (2, 4, 6, 8, 12, 16, 24, 31), 1), ("cfg3", (31,), 2), ("cfg3", (31,), 3), ("cfg3", (31,), 4), ("cfg4", (1, 2, 4, 8, 16, 32, 63), 1), ("cfg2", (15,), 1), ("cfg2", (15,), 4), ("cfg2", (15,), 8)):
    w = W.WORKLOADS[name]()
    ctx = RpContext(0); w.setup(ctx)
    ctx.set_option("fused_lon", 0); ctx.set_option("auto_materialize", 0)
    for nT in nTs:
        p = copy_params(w.inputs.params); p.flags |= FLAG_SKIP_COLLISION
        sel = np.linspace(0, len(w.inputs.T) - 1, nT).round().astype(int)
        L = w.inputs.L if reps == 1 else np.linspace(w.inputs.L.min(), w.inputs.L.max(), len(w.inputs.L) * reps)   # (denser longitudinal grid: more candidates)
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T[sel], w.inputs.traj_len[sel], L, w.inputs.D)
        t = {}
        for kn, opts in KERNELS.items():
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_profiling(1)
            kms = [ctx.plan(inp, want_best_states=False).kernel_ms for _ in range(6)]
            ctx.set_profiling(0)
            t0 = time.perf_counter()
            for _ in range(30):
                ctx.plan(inp, want_best_states=False)
            t[kn] = ((time.perf_counter() - t0) / 30, float(np.median(kms[1:])), ctx.last_kernel())
        best = min(t, key=lambda k: t[k][1])
        print(f"{name} N={p.N:3d} C={inp.n_candidates:7d}: " + "  ".join(f"{kn} [{t[kn][2][3:-7]}] step {t[kn][0]*1e6:7.1f} kernel {t[kn][1]*1e3:7.1f} us" for kn in KERNELS) + f"   fastest kernel: {best}", flush=True)
    ctx.close()
