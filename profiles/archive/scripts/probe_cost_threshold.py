"""From how many candidates on does rp_cost_kernel (one lane per candidate) beat rp_eval_kernel (16 lanes per candidate) on costs-only
plans?  Slices of cfg4's grid (N = 100) and of cfg3's (N = 60), collision query off.   usage (GPU box): python profiles/probe_cost_threshold.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params
os.environ["RP_AMD_NO_FUSED_LON"] = "1"
for name, nTs in (("cfg4", (1, 2, 4, 8, 16, 32, 63)), ("cfg3", (2, 4, 8, 16, 31))):
    w = W.WORKLOADS[name]()
    ctx = RpContext(0); w.setup(ctx)
    for nT in nTs:
        p = copy_params(w.inputs.params); p.flags |= FLAG_SKIP_COLLISION
        sel = np.linspace(0, len(w.inputs.T) - 1, nT).round().astype(int)
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T[sel], w.inputs.traj_len[sel], w.inputs.L, w.inputs.D)
        t = {}
        for ck in ("0", "1"):
            os.environ["RP_AMD_COST_KERNEL"] = ck
            for _ in range(5):
                ctx.plan(inp, want_best_states=False)
            t0 = time.perf_counter()
            for _ in range(40):
                ctx.plan(inp, want_best_states=False)
            t[ck] = (time.perf_counter() - t0) / 40
        print(f"{name} N={p.N:3d} C={inp.n_candidates:7d} ({inp.n_candidates // 64:5d} wavefronts of 64): 16 lanes/candidate {t['0']*1e6:7.1f} us   lane per candidate {t['1']*1e6:7.1f} us   ({t['0']/t['1']:.2f} x)", flush=True)
    ctx.close()
