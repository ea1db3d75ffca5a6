"""rp_cost_kernel (one lane per candidate) against rp_eval_kernel (16 lanes per candidate) on costs-only plans: the same status words
and the same cost BITS, and the step time of both.   usage (GPU box): python profiles/probe_cost_kernel.py [cfg4 cfg5 cfg5obs cfg3 cfg4rb]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params

names = sys.argv[1:] or ["cfg4", "cfg5", "cfg5obs", "cfg3"]
for name in names:
    # (<name>rb: the workload with its road boundary)
    w = W.cfg5(obstacles=50) if name == "cfg5obs" else (W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]())
    ctx = RpContext(0)
    w.setup(ctx)
    got = {}
    for skip in (True, False, "eager"):
        p = copy_params(w.inputs.params)
        os.environ.pop("RP_AMD_LAZY", None)
        if skip == "eager":   # production mode with the eager collision query pinned (RP_AMD_LAZY=0)
            os.environ["RP_AMD_LAZY"] = "0"
        elif skip:
            p.flags |= FLAG_SKIP_COLLISION
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        for ck in ("0", "1"):
            os.environ["RP_AMD_COST_KERNEL"] = ck
            for _ in range(5):
                out = ctx.plan(inp, want_best_states=False)
            t0 = time.perf_counter()
            for _ in range(30):
                out = ctx.plan(inp, want_best_states=False)
            dt = (time.perf_counter() - t0) / 30
            st, cs = ctx.fetch_status()
            got[(skip, ck)] = (st, cs, out, dt, ctx.last_path())
        a, b = got[(skip, "0")], got[(skip, "1")]
        same_status = np.array_equal(a[0], b[0]) if (a[4] == b[4] == 0 or skip is True) else None
        same_cost = np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
        print(f"{name:8s} {'no collision query' if skip is True else ('eager query pinned ' if skip == 'eager' else 'production mode    ')} C={len(a[0]):8d}  16 lanes/candidate {a[3]*1e6:8.1f} us  "
              f"lane per candidate {b[3]*1e6:8.1f} us  ({a[3]/b[3]:.2f} x)  status identical: {same_status}  cost bits identical: {same_cost}  "
              f"winner {a[2].best_index} / {b[2].best_index}  paths {a[4]}/{b[4]}", flush=True)
    ctx.close()
