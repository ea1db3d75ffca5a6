import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_SKIP_COLLISION, PlanInputs, copy_params
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]()
ctx = RpContext(0); w.setup(ctx)
p = copy_params(w.inputs.params); p.flags |= FLAG_SKIP_COLLISION
inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
res = {}
ctx.set_option("fused_lon", 0); ctx.set_option("auto_materialize", 0)
for ck, opts in (("eval16", {"cost_kernel": 0, "chunk_kernel": 0}), ("lane", {"cost_kernel": 1, "chunk_kernel": 0}), ("chunk", {"cost_kernel": 0, "chunk_kernel": 1})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.plan(inp, want_best_states=False)
    res[ck] = ctx.fetch_status() + (ctx.last_kernel(),)
for other in ("lane", "chunk"):
    a0, b0 = res["eval16"], res[other]
    okc = ~np.isnan(a0[1])
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'cfg4'}: {a0[2]} vs {b0[2]}: candidates with a cost {int(okc.sum())}, cost bits differing "
          f"{int(np.sum(okc & (a0[1].view(np.uint64) != b0[1].view(np.uint64))))}, status words differing {int(np.sum(a0[0] != b0[0]))}")
a, b = res["eval16"][1], res["lane"][1]
ok = ~np.isnan(a)
diff = np.flatnonzero(ok & (a.view(np.uint64) != b.view(np.uint64)))
print("candidates with a cost:", ok.sum(), "differing:", len(diff), "max rel", np.max(np.abs(a[diff] - b[diff]) / np.abs(a[diff])) if len(diff) else 0)
nT, nL, nD = len(inp.T), len(inp.L), len(inp.D)
iT, r = np.divmod(diff, nL * nD); iL, iD = np.divmod(r, nD)
print("by iT:", np.bincount(iT, minlength=nT).tolist())
print("by iD:", np.bincount(iD, minlength=nD).tolist())
print("traj_len:", inp.traj_len.tolist())
print("first few:", [(int(i), float(a[i]), float(b[i]), int(a.view(np.int64)[i] - b.view(np.int64)[i])) for i in diff[:8]])
