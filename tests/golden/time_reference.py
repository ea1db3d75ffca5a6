#!/usr/bin/env python3
"""B1 / B0 calibration of the Python-loop CPU baseline.

``bench.py``'s ``cpu_baseline.numpy_loop`` leg times ``oracle/numpy_loop.py`` -- a loop-faithful restatement of the reference's
per-candidate Python loop -- because the reference itself cannot travel to the GPU box.  This script, which runs ONLY in the
build container (it imports the unmodified reference from /root/reference under the third-party stand-ins of
``_ref_shims.py``, exactly as ``make_golden.py`` does), times on the SAME inputs, on one core:

  B0  the reference: ``generate_trajectories_at_level`` (sampling.py:202-242) -> ``_check_kinematics``
      (reactive_planner.py:715-969) -> ``TrajectoryBundle.sort`` -> ``_check_collisions`` (reactive_planner.py:1019-1063),
      ``multiproc=False`` -- the call path of one sampling level of ``plan()``;
  B1  ``oracle/numpy_loop.py: plan`` on the fixture of the same case (grids in the reference's iteration order);

on cfg1 level 3 (ZAM_Over-1_1, 3 060 candidates, N = 20: the reference's shipped size) and cfg2 (ZAM_Tjunction 15x15x31,
7 440 candidates, N = 30), and writes ``tests/golden/cpu_calibration.json`` -- DATA only (rates, their ratio, the host CPU).
``bench.py`` reads that file and reports ``cpu_baseline.numpy_loop.ratio_to_reference`` = B1 / B0 beside the B1 rate it measures
on the GPU box's host, so that the rate of "the reference's NumPy/Python loop" there is B1(box) / ratio.

    python tests/golden/time_reference.py [--reps 3]
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, REPO)

import make_golden as mg              # noqa: E402  (installs the stand-ins, imports the reference)
import make_baseline_golden as mb     # noqa: E402

from _golden import Golden            # noqa: E402
from oracle import numpy_loop         # noqa: E402

CASES = ["cfg1_ref_l3", "cfg2_ref"]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_reference(bc, reps):
    """One sampling level through the reference's own methods; returns (candidates, winner, best-of-reps seconds per phase)."""
    best = None
    for _ in range(reps):
        case, _w = mb.case_of(bc)
        rp = mg.make_planner(case)
        assert rp.config.debug.multiproc is False
        t0 = time.perf_counter()
        bundle = mg.sample(rp, case)                       # A1-A7: grids, polynomial solves, TrajectorySample objects
        t1 = time.perf_counter()
        trajs = list(bundle.trajectories)
        opt = rp._get_optimal_trajectory(bundle)           # A8-A15: kinematics + cost, sort, lazy collision walk
        t2 = time.perf_counter()
        rec = (len(trajs), trajs.index(opt) if opt is not None else -1, t1 - t0, t2 - t1)
        if best is None or rec[2] + rec[3] < best[2] + best[3]:
            best = rec
    return best


def time_numpy_loop(name, reps):
    g = Golden(name)
    tb = g.oracle_tables()
    best, r = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = numpy_loop.plan(g.inputs, tb)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    return g.inputs.n_candidates, int(r["best_index"]), best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(HERE, "cpu_calibration.json"))
    args = ap.parse_args()
    by_name = {bc["name"]: bc for bc in mb.baseline_cases()}
    out = {"what": "B0 = the reference's own per-level call path (sample -> _check_kinematics -> sort -> _check_collisions, multiproc off), "
                   "B1 = oracle/numpy_loop.py on the same inputs; one core, best of --reps; ratio = B1 / B0",
           "generator": "tests/golden/time_reference.py", "host_cpu": cpu_model(), "nproc": os.cpu_count(), "reps": args.reps,
           "python": platform.python_version(), "cases": {}}
    ratios = []
    for name in CASES:
        C, win, t_s, t_o = time_reference(by_name[name], args.reps)
        C1, win1, t_n = time_numpy_loop(name, args.reps)
        assert C == C1, (C, C1)
        assert win == win1, f"{name}: reference winner {win}, numpy_loop winner {win1}"
        b0, b1 = C / (t_s + t_o), C / t_n
        out["cases"][name] = {"candidates": C, "winner": win, "reference_sampling_s": t_s, "reference_check_sort_collide_s": t_o,
                              "reference_candidates_per_s": b0, "numpy_loop_s": t_n, "numpy_loop_candidates_per_s": b1,
                              "ratio_numpy_loop_to_reference": b1 / b0}
        ratios.append(b1 / b0)
        print(f"{name}: C={C} winner={win}  B0 {b0:8.1f} cand/s (sampling {t_s:.2f} s + check/sort/collide {t_o:.2f} s)   "
              f"B1 {b1:8.1f} cand/s ({t_n:.2f} s)   B1/B0 = {b1 / b0:.3f}", flush=True)
    out["ratio_numpy_loop_to_reference"] = float(sum(ratios) / len(ratios))
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
