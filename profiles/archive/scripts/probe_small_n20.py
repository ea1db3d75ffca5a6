"""Probe: small batches at N = 20 / 30 (the shipped configurations): step time per lanes-per-candidate choice."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
cases = [("cfg1 L1", W.cfg1(1), None), ("cfg1 L2", W.cfg1(2), None), ("cfg1 L3", W.cfg1(3), None)]
w2 = W.cfg2()
for frac in (0.14, 0.28, 0.55, 1.0):
    cases.append((f"cfg2 x{frac}", w2, int(w2.n_candidates * frac)))
for label, w, cnt in cases:
    for mode, fl in (("production", 0), ("draw", FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)):
        p = copy_params(w.inputs.params); p.flags |= fl
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        row = []
        for g in ("default", "16", "32"):
            os.environ.pop("RP_AMD_G", None)
            if g != "default": os.environ["RP_AMD_G"] = g
            ctx = RpContext(0); w.setup(ctx)
            args = (inp,) if cnt is None else (inp, 0, cnt)
            for _ in range(10): out = ctx.plan(*args)
            t0 = time.perf_counter()
            for _ in range(200): ctx.plan(*args)
            row.append(f"G={g}: {(time.perf_counter() - t0) / 200 * 1e6:6.1f} us")
            ctx.close()
        print(f"{label:10s} C={(w.n_candidates if cnt is None else cnt):5d} N={inp.params.N} {mode:10s} " + "  ".join(row))
