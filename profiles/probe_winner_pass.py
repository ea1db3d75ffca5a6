"""The re-evaluation of a production-mode plan's winner (its state rows; two-kernel path) with and without the collision query
(option "winner_skip_query"): rp_plan step time, the option switched on one context, alternating; the winner's rows must not change.
usage (GPU box): python profiles/probe_winner_pass.py [cfg3 cfg3f cfg4 cfg2]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W, _capi
from commonroad_rp_amd._capi import RpContext
print("library", _capi.source_hash())
for name in sys.argv[1:] or ["cfg3", "cfg3f", "cfg3rb", "cfg4", "cfg4rb"]:
    w = W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]()
    ctx = RpContext(0); w.setup(ctx)
    res = {0: [], 1: []}
    rows = {}
    for rep in range(4):
        for v in (0, 1):
            ctx.set_option("winner_skip_query", v)
            for _ in range(20):
                out = ctx.plan(w.inputs)
            rows[v] = (out.best_index, out.best_cost, None if out.best_states is None else out.best_states.copy())
            t = []
            for _ in range(150):
                t0 = time.perf_counter(); ctx.plan(w.inputs); t.append(time.perf_counter() - t0)
            res[v].append(np.median(t) * 1e6)
    same = rows[0][0] == rows[1][0] and rows[0][1] == rows[1][1] and (rows[0][2] is None) == (rows[1][2] is None) and \
        (rows[0][2] is None or np.array_equal(rows[0][2], rows[1][2]))
    print(f"{name:7s} production step, winner pass with the query {np.round(res[0], 1)} us | without {np.round(res[1], 1)} us | median {np.median(res[0]):.1f} -> {np.median(res[1]):.1f}"
          f" | winner {rows[1][0]}, rows bit-identical: {same}; path {ctx.last_path()}", flush=True)
    ctx.close()
