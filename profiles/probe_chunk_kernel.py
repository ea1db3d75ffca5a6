"""rp_chunk_kernel (one lane per candidate and step block of 16) against rp_eval_kernel (16 lanes per candidate) and rp_cost_kernel (one lane
per candidate) on costs-only plans: the same status words and the same cost BITS, the step time (compiled-free: ctx.plan without winner
rows) and the evaluation kernel's own time (HIP events on the launch) of each.
usage (GPU box): python profiles/probe_chunk_kernel.py [cfg3 cfg3f cfg3rb cfg4 cfg2 ...]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params

KERNELS = {"eval16": {"chunk_kernel": 0, "cost_kernel": 0, "sweep": 0}, "lane": {"chunk_kernel": 0, "cost_kernel": 1, "sweep": 0},
           "chunk": {"chunk_kernel": 1, "cost_kernel": 0, "sweep": 0}, "auto": {"chunk_kernel": -1, "cost_kernel": -1, "sweep": 0},
           "sweep": {"chunk_kernel": -1, "cost_kernel": -1, "sweep": 1}}
names = sys.argv[1:] or ["cfg3", "cfg3f", "cfg3rb", "cfg4"]
for name in names:
    w = W.cfg5(obstacles=50) if name == "cfg5obs" else (W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]())
    ctx = RpContext(0)
    w.setup(ctx)
    ctx.set_option("fused_lon", 0)   # (two-kernel path whatever the batch size: the lane kernels read profiles from rp_lon_kernel)
    ctx.set_option("auto_materialize", 0)
    for mode in ("no collision query", "eager query", "production (default path)", "production (cost-ordered forced)"):
        p = copy_params(w.inputs.params)
        ctx.set_option("lazy", 0 if mode == "eager query" else (1 if "forced" in mode else -1))
        if mode == "no collision query":
            p.flags |= FLAG_SKIP_COLLISION
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        got = {}
        for kn, opts in KERNELS.items():
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_profiling(1)
            kms = []
            for _ in range(6):
                out = ctx.plan(inp, want_best_states=False)
                kms.append(out.kernel_ms)
            ctx.set_profiling(0)
            for _ in range(5):
                out = ctx.plan(inp, want_best_states=False)
            t0 = time.perf_counter()
            for _ in range(40):
                out = ctx.plan(inp, want_best_states=False)
            dt = (time.perf_counter() - t0) / 40
            st, cs = ctx.fetch_status()
            got[kn] = (st, cs, out, dt, ctx.last_path(), float(np.median(kms[1:])), ctx.last_kernel())
        a = got["eval16"]
        line = f"{name:8s} {mode:26s} C={len(a[0]):8d} N={p.N:3d}"
        for kn in KERNELS:
            b = got[kn]
            same_status = np.array_equal(a[0], b[0]) if (a[4] == b[4] == 0 or mode == "no collision query") else None
            if same_status is None and a[4] == 0:   # cost-ordered against eager: every label in front of the winner, nothing wrong behind it
                la, lb = a[0] & 3, b[0] & 3
                d = la != lb
                before = (a[1] < a[2].best_cost) | ((a[1] == a[2].best_cost) & (np.arange(len(la)) < a[2].best_index)) if a[2].best_index >= 0 else np.ones(len(la), bool)
                same_status = "front" if (not np.any(d & before) and np.all((la[d] == 3) & (lb[d] == 1)) and np.array_equal(a[0] >> 4, b[0] >> 4)) else "WRONG"
            same_cost = np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
            line += (f" | {kn} [{b[6][3:-7]}] step {b[3]*1e6:7.1f} us kernel {b[5]*1e3:7.1f} us path {b[4]}"
                     + ("" if kn == "eval16" else f" status=={same_status} costbits=={same_cost} winner {'same' if a[2].best_index == b[2].best_index and a[2].n_collision_before_best == b[2].n_collision_before_best else 'DIFFERENT'}"))
        print(line, flush=True)
    ctx.close()
