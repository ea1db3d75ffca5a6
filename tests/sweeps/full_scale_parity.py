"""One-off validation: labels / first-failure reasons / winner of the FULL benchmark workloads against the oracle's
brute force (the unit tests check slices).  usage (GPU box): python tests/sweeps/full_scale_parity.py"""
import sys, os, time, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL
from oracle import oracle
ctx = RpContext(0)
cases = [("cfg2", {}), ("cfg2", {"road_boundary": True}), ("cfg3", {}), ("cfg3", {"road_boundary": True}), ("cfg3f", {}), ("cfg3f", {"road_boundary": True}), ("cfg4", {}),
         ("cfg4", {"road_boundary": True}), ("cfg5", {"obstacles": 50})]
bad = 0
for name, kw in cases:
    w = getattr(W, name)(**kw)
    w.setup(ctx)
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    for extra in (0, FLAG_DRAW_ALL):
        p = copy_params(w.inputs.params); p.flags |= extra
        inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
        t0 = time.time()
        o = oracle.plan(inp, tb, want_states=False, nthreads=16)
        t1 = time.time()
        # both ways of answering the collision query (production-mode plans of large batches; rp_amd.h: rp_last_path): the eager
        # kernel labels every colliding candidate; the cost-ordered stage only the ones it looked at -- the others keep FEASIBLE,
        # n_collision counts the collisions found (>= the ones before the winner), everything else is the same
        # the launch policies a production-mode plan of a large batch can take (options of the context, rp_set_option): eager |
        # cost-ordered by the library's own rule (list rounds; rp_chunk_kernel / rp_cost_kernel for pass 1; bounded sweep behind
        # exhausted lists of large batches) | list rounds with the 16-lane kernel for pass 1 | bounded sweep instead of the lists
        for tag, opts in (("eager", {"lazy": 0}), ("cost-ordered", {"lazy": 1}), ("cost-ordered, 16-lane pass 1", {"lazy": 1, "chunk_kernel": 0, "cost_kernel": 0}),
                          ("bounded sweep", {"lazy": 1, "sweep": 1})):
            for k in ("lazy", "chunk_kernel", "cost_kernel", "sweep"):
                ctx.set_option(k, -1)
            for k, v in opts.items():
                ctx.set_option(k, v)
            out = ctx.plan(inp)
            path = ctx.last_path()
            st, cost = ctx.fetch_status()
            lab, olab = st & 3, o.status & 3
            if path in (1, 3):
                diff = lab != olab
                lab_ok = bool(np.all((olab[diff] == 3) & (lab[diff] == 1)))
                if out.best_index >= 0:   # every candidate in front of the winner carries its label
                    before = (cost < out.best_cost) | ((cost == out.best_cost) & (np.arange(len(cost)) < out.best_index))
                    lab_ok = lab_ok and not np.any(diff & before)
                coll_ok = o.out.n_collision_before_best <= out.n_collision <= o.out.n_collision and \
                    (out.n_collision == int(np.sum(lab == 3)) if path == 1 else out.n_collision == out.n_collision_before_best)
            else:
                lab_ok = np.array_equal(lab, olab)
                coll_ok = out.n_collision == o.out.n_collision
            lab_ok = lab_ok and np.array_equal((st >> 4) & 7, (o.status >> 4) & 7)
            has = ~np.isnan(o.cost)
            cost_ok = np.all(np.isnan(cost[~has])) and np.allclose(cost[has], o.cost[has], rtol=1e-9, atol=0)
            win_ok = out.best_index == o.out.best_index and coll_ok and \
                out.n_collision_before_best == o.out.n_collision_before_best and out.n_feasible == o.out.n_feasible
            ok = lab_ok and cost_ok and win_ok
            bad += not ok
            print(f"{name} {kw} draw={bool(extra)} [{tag}] path={('eager', 'cost-ordered', 'cost-ordered, then eager', 'bounded sweep')[path]} ({ctx.last_kernel()}): {inp.n_candidates} candidates, "
                  f"colliding {o.out.n_collision} (found {out.n_collision}), winner {out.best_index}: "
                  f"{'OK' if ok else 'MISMATCH'} (labels {lab_ok}, costs {cost_ok}, winner/counters {win_ok}; oracle {t1 - t0:.1f} s)", flush=True)
            if extra or path == 0 and tag == "eager" and inp.n_candidates <= 16384:
                break   # (draw-mode plans and small batches have one path only)
print("all full-scale checks passed" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
