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
for ck in ("0", "1"):
    os.environ["RP_AMD_COST_KERNEL"] = ck
    ctx.plan(inp, want_best_states=False)
    res[ck] = ctx.fetch_status()
a, b = res["0"][1], res["1"][1]
ok = ~np.isnan(a)
diff = np.flatnonzero(ok & (a.view(np.uint64) != b.view(np.uint64)))
print("candidates with a cost:", ok.sum(), "differing:", len(diff), "max rel", np.max(np.abs(a[diff] - b[diff]) / np.abs(a[diff])) if len(diff) else 0)
nT, nL, nD = len(inp.T), len(inp.L), len(inp.D)
iT, r = np.divmod(diff, nL * nD); iL, iD = np.divmod(r, nD)
print("by iT:", np.bincount(iT, minlength=nT).tolist())
print("by iD:", np.bincount(iD, minlength=nD).tolist())
print("traj_len:", inp.traj_len.tolist())
print("first few:", [(int(i), float(a[i]), float(b[i]), int(a.view(np.int64)[i] - b.view(np.int64)[i])) for i in diff[:8]])
