#!/usr/bin/env python3
"""Benchmark of the hot path: candidate trajectories / s (sample + convert + check + cost + collide
+ select) per replanning step, through the C ABI of librp_amd.so.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--mode draw|materialize|fused]

One *step* = one ``rp_plan`` call over the whole candidate batch of one replanning cycle (grids and parameters staged
host->device inside the call: a few hundred bytes; tables resident), including the winner's state block coming back to
the host.  Consecutive steps are consecutive cycles of a REPLANNING SEQUENCE (``workloads.replan_sequence``: the scenario
driven in closed loop, every step a new initial state, velocity grid and time index; cfg5: a seeded list of initial
states), not one repeated input.  A timed region is exactly K steps; regions are repeated until at least
``--min-seconds`` of timed work has accumulated, ``ms_per_step`` is the median region, the spread is reported.

N = 1 (default): headline = cfg3 (BASELINE.json configs[2], the largest configuration tagged 1 x MI355X) in draw mode, plus --
in the detail file, their key numbers in the line --
  ``configs``          draw- and production-mode records of cfg1, cfg2, cfg3f (cfg3's grid in mostly-free traffic), <cfg>rb (cfg2, cfg3, cfg3f, cfg4 with the road boundary),
                       cfg4, cfg5, each with its own roofline
  ``fused_mode``       the headline workload in production mode (12 B per candidate leave the kernel)
  ``plan_latency_ms``  p50 / p90 of ReactivePlanner.plan() over closed-loop replans (Python boundary included)
  ``cpu_baseline``     the C port of the reference algorithm (oracle/) on the host cores, bounded sample
N > 1: one process per GPU (the driver's ``python -m torch.distributed.run ... bench.py --gpus N``; started without a
  launcher, this script starts the N ranks itself as child processes BEFORE anything touches the GPU and relays rank
  0's line).  Headline = WEAK scaling of the N = 1 workload: the grid's longitudinal samples densified N times, every rank
  evaluates its contiguous range of the candidate index -- a shard the size of the N = 1 grid, the same per-GPU work as
  `--gpus 1` -- and the ranks exchange one {cost, index, counters, winner block} message per step
  (commonroad_rp_amd/distributed.py; no data-path collective).  Beside it (`strong`): BASELINE.json configs[3], cfg4's
  512 064-candidate grid cut into N ranges, the same regions with the other exchange transport, the exchange time per step
  of both, rank 0 alone on the whole grid.  `--scaling strong` makes that record the headline.

Prints ONE JSON line (rank 0): the COMPACT record (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, p50 / p90 of plan(), one
number pair per side configuration, the N > 1 exchange keys); the full record of the run (every side record with its spread, roofline
and description) is written to `gpurun_out/bench_detail_n<N>.json` (the repo root without that directory) and named in the line's
`detail`.  Modes:
  draw        every candidate fully evaluated (no pre-filter / early exit, the reference's draw_traj_set semantics)
              and all 14 state rows of every candidate written to HBM.  Default: work per candidate is data independent
              and the byte count is SURVEY 8(d)'s bytes = C*12 + C*112*(N+1) + 112*(N+1).
  materialize production early-exit semantics, state rows of feasible candidates written
  fused       production semantics, 12 B per candidate leave the kernel
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # half the FP32 vector peak of the same table (157.3 TFLOPS): 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2.4 GHz
# counter files written by profiles/collect_pmc.sh / collect_fp64.sh; every entry names the hash of the sources of the library it
# was measured on (rp_source_hash): an entry is reported only when that is the library this run has loaded
PMC_FILE = os.path.join(REPO, "profiles", "r05_pmc_traffic.json")
FLOP_FILE = os.path.join(REPO, "profiles", "r05_fp64_flops.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, help="default: cfg3 (the largest configuration BASELINE.json tags 1 x MI355X); at N > 1 the shard every rank gets")
    ap.add_argument("--mode", default="draw", choices=["draw", "materialize", "fused"])
    ap.add_argument("--caller", default="c", choices=["c", "python"],
                    help="who calls rp_plan inside the timed regions: a compiled host loop over the C ABI, or the ctypes binding")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N > 1: which record is the headline -- weak: every rank a shard the size of the N = 1 workload (its longitudinal "
                         "grid densified N times), the same per-GPU work as `--gpus 1`; strong: cfg4's grid cut into N ranges")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="timed work per measured record (regions of K steps are repeated)")
    ap.add_argument("--sequence", type=int, default=32, help="replanning cycles in the input sequence")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the cfg3 / cfg4 / cfg5 records")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed region of the headline (no other leg): counter passes")
    ap.add_argument("--road-boundary", action="store_true",
                    help="scenario workloads: add the road boundary (thin rectangles, collision.road_boundary_obb) to the obstacle tables")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    # Only the JSON line may reach stdout: native libraries (RCCL prints a version banner) write to fd 1
    # directly, so fd 1 is pointed at stderr for the duration of the run and restored for the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import logging
    logging.getLogger("RP_LOGGER").setLevel(logging.ERROR)   # (the all-levels-fail leg would warn once per cycle on stderr)
    try:
        result = run(args)
    finally:
        sys.stdout.flush()
        sys.stderr.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if result is not None:
        print(emit(result, args), flush=True)


def self_launch(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes (torch.distributed.run),
    before this process has imported torch or touched the GPU (never exec from a process that has), relay rank 0's JSON
    line and return the children's status."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and out.endswith("}"):
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


# --------------------------------------------------------------------------------------------------------------------
def with_mode(seq, mode):
    """The sequence's inputs with the flags of a mode."""
    from commonroad_rp_amd._capi import PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
    flags = {"draw": FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL, "materialize": FLAG_MATERIALIZE_ALL, "fused": 0}[mode]
    out = []
    for q in seq:
        p = copy_params(q.params)
        p.flags = (p.flags & ~(FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)) | flags
        out.append(PlanInputs(p, q.cost, q.T, q.traj_len, q.L, q.D))
    return out


def measure(step, n_inputs, steps, warmup, min_seconds, sync, barrier=None, max_regions=4000, region=None):
    """Timed regions of exactly ``steps`` steps (barrier + device sync on both sides), repeated until ``min_seconds`` of
    timed work; step(k) runs cycle k of the input sequence -- or region(k0, steps) runs ``steps`` of them in one call (the
    compiled host loop, _capi.HostLoop).  Returns the per-region step times (ms)."""
    k = 0
    if region is not None and warmup > 0:
        region(0, warmup)
        k = warmup
    else:
        for _ in range(warmup):
            step(k % n_inputs)
            k += 1
    regions, total = [], 0.0
    while (total < min_seconds and len(regions) < max_regions) or not regions:
        if barrier:
            barrier()
        sync()
        t0 = time.perf_counter()
        if region is not None:
            region(k, steps)
            k += steps
        else:
            for _ in range(steps):
                step(k % n_inputs)
                k += 1
        sync()
        if barrier:
            barrier()
        el = time.perf_counter() - t0
        if barrier:
            el = barrier.max_over_ranks(el)   # every rank sees the same number -> the same number of regions
        regions.append(el / steps * 1e3)
        total += el
    return regions, total


def spread(measured):
    r, total = np.asarray(measured[0]), measured[1]
    return {"regions": int(len(r)), "timed_s": float(total), "median": float(np.median(r)),
            "min": float(r.min()), "p10": float(np.percentile(r, 10)), "p90": float(np.percentile(r, 90)), "max": float(r.max())}


def load_json(path):
    try:
        return json.load(open(path))
    except Exception:
        return {}


def counter_entry(path, key):
    """(entry, note): the entry of a counter file if it was measured on the sources of the loaded library, else (None, why)."""
    from commonroad_rp_amd import _capi
    e = load_json(path).get(key)
    if not e:
        return None, None
    have = _capi.source_hash()
    if e.get("source_hash") != have:
        return None, {"file": os.path.basename(path), "key": key, "measured_on": e.get("source_hash"), "loaded": have}
    return e, None


def roofline_record(name, mode, n_steps_plus1, cand_mean, feasible_mean, kernel_ms, single_gpu=True, kernel="rp_eval_kernel"):
    """Roofline of the kernel that evaluates the batch (``kernel``: rp_last_kernel of the record's steps) for one record.
    draw / materialize: HBM write stream, algorithmic bytes per launch / average kernel duration (HIP events inside
    rp_plan, on the context's stream).  fused: 12 B per candidate leave the kernel, the kernel is FP64-VALU bound: counted FP64
    flops of the executed instruction stream per (candidate, step) (SQ counters of the kernel this workload takes,
    profiles/fp64_summary.py -> FLOP_FILE) x candidates x steps / kernel duration against the FP64 vector peak."""
    blk = 112 * n_steps_plus1
    pmc, stale = counter_entry(PMC_FILE, f"{name}:{mode}") if single_gpu else (None, None)
    traffic = pmc.get("traffic_bytes") if pmc else None
    extra = {"stale_profile": stale} if stale else {}
    if mode == "fused":
        # (<name>rb, road boundary, when it has no count of its own: the flop count of the plain workload -- the static-shape walk adds tests, so the fraction is a lower bound)
        fl, stale_f = counter_entry(FLOP_FILE, name)
        if fl is None and name.endswith("rb"):
            fl, stale_f = counter_entry(FLOP_FILE, name[:-2])
        if stale_f:
            extra = {"stale_profile": stale_f}
        if fl and kernel_ms > 0:
            flops = float(fl["flops_per_candidate_step"]) * cand_mean * n_steps_plus1
            ach = flops / (kernel_ms * 1e-3) / 1e12
            return {"bound": "valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                    "traffic": traffic, "kernel": kernel, "kernel_ms": kernel_ms, "flops_per_launch": flops,
                    "flops_per_candidate_step": fl["flops_per_candidate_step"], "flop_model": fl.get("model"), **extra}
        bytes_per_launch = cand_mean * 12
    elif mode == "draw":
        bytes_per_launch = cand_mean * 12 + cand_mean * blk
    else:
        bytes_per_launch = cand_mean * 12 + feasible_mean * blk
    ach = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else float("nan")
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_launch": bytes_per_launch, **extra}


def run_record(ctx, w, seq, mode, steps, warmup, min_seconds, sync, name=None, profile_every=8, caller="c"):
    """One measured record on one GPU: the sequence in one mode.  ``caller``: who calls rp_plan inside the timed regions --
    "c": a compiled host loop over the C ABI (csrc/rp_hostloop.c), "python": the ctypes binding, one ``ctx.plan`` per step."""
    from commonroad_rp_amd._capi import HostLoop
    inputs = with_mode(seq, mode)
    kms, feas, paths = [], [], [0, 0, 0, 0]
    ctx.set_profiling(profile_every)   # HIP events around the evaluation kernel of every 8th step (a bracket costs ~8 us of stream time)

    def step(k):
        out = ctx.plan(inputs[k])
        if out.kernel_ms > 0:
            kms.append(out.kernel_ms)
        feas.append(out.n_feasible)
        paths[ctx.last_path()] += 1
    region = None
    if caller == "c":
        loop = HostLoop(ctx, inputs)
        tot = {"kms": 0.0, "kn": 0, "feas": 0, "n": 0}

        def region(k0, n):
            st = loop.run(k0, n)
            tot["kms"] += st.kernel_ms_sum; tot["kn"] += st.kernel_ms_n; tot["feas"] += st.feasible_sum; tot["n"] += n
            for i in range(4):
                paths[i] += st.paths[i]
    regions = measure(step, len(inputs), steps, warmup, min_seconds, sync, region=region)
    ctx.set_profiling(0)
    sp = spread(regions)
    cand = float(np.mean([q.n_candidates for q in inputs]))
    if region is not None:
        kms = [tot["kms"] / tot["kn"]] if tot["kn"] else []
        feas = [tot["feas"] / max(tot["n"], 1)]
    kernel_ms = float(np.mean(kms)) if kms else float("nan")
    n1 = inputs[0].params.N + 1
    rec = {"mode": mode, "candidates_per_step": cand, "ms_per_step": sp["median"], "value": cand / (sp["median"] * 1e-3),
           "unit": "candidates/s", "steps": steps, "spread_ms": sp, "kernel_ms": kernel_ms, "sequence": len(inputs),
           "caller": "compiled host loop over the C ABI (rp_hostloop.c)" if region is not None else "Python binding (ctypes), one call per step",
           # how the steps answered the collision query: eager (every pose of every candidate), cost-ordered stage, stage + eager fallback
           "collision_path_steps": {"eager": paths[0], "cost_ordered": paths[1], "cost_ordered_then_eager": paths[2], "cost_ordered_sweep": paths[3]},
           "roofline": roofline_record(name or w.name, mode, n1, cand, float(np.mean(feas)) if feas else 0.0, kernel_ms,
                                       kernel=ctx.last_kernel())}
    return rec


def steps_for(ms_guess, steps):
    """K for the side records: about 50 ms per region, at most the headline's K."""
    return int(max(3, min(steps, 50.0 / max(ms_guess, 1e-3))))


# --------------------------------------------------------------------------------------------------------------------
def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    # RP_BENCH_REHEARSE=1: several ranks on ONE GPU (gloo group, every rank on cuda:0) -- a functional rehearsal of
    # the N > 1 path on a one-GPU box; its timings mean nothing.  RP_BENCH_FORCE_DIST=1: the N > 1 code path with one rank.
    rehearse = os.environ.get("RP_BENCH_REHEARSE") == "1"
    force_dist = os.environ.get("RP_BENCH_FORCE_DIST") == "1"
    if rehearse:
        local_rank = 0
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    try:
        if dist is None:
            result = run_single(args, torch, local_rank)
        else:
            result = run_multi(args, torch, dist, rank, local_rank, world, rehearse)
    finally:
        if dist is not None:
            from commonroad_rp_amd.distributed import close_exchanges
            close_exchanges()
            dist.destroy_process_group()
    return result if rank == 0 else None


# --------------------------------------------------------------------------------------------------------------------
# The driver reads rank 0's line out of a 16-KB tail of stdout: the line is the COMPACT record (< 4 KB, tests/test_bench_line.py);
# everything else the run measured goes to a file next to it, named in the line.
LINE_LIMIT = 4096
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_launch", "flops_per_launch")


def _r(v, digits=5):
    """floats to ``digits`` significant digits (the line is read by people and parsed by json.loads)"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _r(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full: dict, detail_path=None) -> dict:
    """The record the driver parses: the contract's keys, ``roofline`` and ``cpu_baseline`` of the headline, p50 / p90 of plan(), one
    number pair per side configuration, the N > 1 exchange keys -- nothing that grows with the number of side records' fields."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    out["config"] = _pick(full.get("config", {}), ("workload", "mode", "candidates_per_step", "candidates_per_gpu", "horizon_steps",
                                                    "n_obstacles", "parallelism", "exchange"))
    if len(out["config"].get("workload", "")) > 160:
        out["config"]["workload"] = out["config"]["workload"][:157] + "..."
    if "roofline" in full:
        out["roofline"] = _pick(full["roofline"], ROOFLINE_KEYS)
        if full["roofline"].get("stale_profile"):
            out["roofline"]["stale_profile"] = True
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = cb.get("sample_short") or cb.get("sample", "")[:120]
        if "all_cores" in cb:
            c["all_cores"] = _pick(cb["all_cores"], ("value", "cores", "nproc"))
        if "numpy_loop" in cb:
            c["numpy_loop"] = _pick(cb["numpy_loop"], ("value", "cores", "ratio_to_reference", "reference_value"))
        out["cpu_baseline"] = c
    pl = full.get("plan_latency_ms")
    if pl:
        out["plan_latency_ms"] = _pick(pl, ("p50", "p90", "workload", "world", "error"))
    fm = full.get("fused_mode")
    if fm:   # the headline workload in production mode (12 B per candidate leave the kernel)
        out["production_mode"] = dict(_pick(fm, ("ms_per_step", "value")), roofline=_pick(fm.get("roofline", {}), ("bound", "frac", "kernel", "kernel_ms")))
    if "python_binding" in full:
        out["python_binding_ms_per_step"] = full["python_binding"].get("ms_per_step")
    side = {}
    for name, rec in (full.get("configs") or {}).items():   # [draw ms, draw roofline frac, production ms, production roofline frac]
        side[name] = [rec.get(m, {}).get(k) if k == "ms_per_step" else rec.get(m, {}).get("roofline", {}).get("frac")
                      for m in ("draw", "fused") for k in ("ms_per_step", "frac")]
    if side:
        out["side_configs"] = dict(side, _cols="draw ms/step, draw roofline frac, production ms/step, production roofline frac")
    ll = full.get("level_loop_latency_ms")
    if isinstance(ll, dict) and "error" not in ll:
        out["level_loop_p50_ms"] = {pol: {"closed_loop": rec.get("closed_loop", {}).get("p50"), "all_levels_fail": rec.get("all_levels_fail", {}).get("p50")}
                                    for pol, rec in ll.items() if isinstance(rec, dict)}
    for k in ("exchange", "exchange_ms_per_step", "ranks_seen_by_rccl", "other_transport", "one_gpu_same_grid", "strong_scaling", "rehearsal",
              "wait_mode"):
        if k in full:
            out[k] = full[k]
    if detail_path:
        out["detail"] = detail_path
    return _r(out)


def emit(full: dict, args) -> str:
    """Write the full record to ``bench_detail*.json`` (gpurun_out/ when it exists: that directory travels back from a GPU box) and
    return the compact line.  Fields are dropped from the tail of the line, never the contract's, should it ever pass LINE_LIMIT."""
    name = f"bench_detail_n{full.get('n_gpus', 1)}.json"
    out_dir = os.path.join(REPO, "gpurun_out") if os.path.isdir(os.path.join(REPO, "gpurun_out")) else REPO
    path = os.path.join(out_dir, name)
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        rel = os.path.relpath(path, REPO)
    except OSError as e:
        print(f"bench.py: cannot write {path}: {e}", file=sys.stderr)
        rel = None
    rec = compact_line(full, rel)
    line = json.dumps(rec, separators=(",", ":"))
    for k in ("level_loop_p50_ms", "side_configs", "python_binding_ms_per_step", "production_mode", "strong_scaling"):
        if len(line) < LINE_LIMIT:
            break
        rec.pop(k, None)
        line = json.dumps(rec, separators=(",", ":"))
    return line


def run_single(args, torch, device):
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    name = args.workload or "cfg3"
    base = W.WORKLOADS[name](road_boundary=True) if args.road_boundary else W.WORKLOADS[name]()
    ctx = RpContext(device)
    base.setup(ctx)
    sync = torch.cuda.synchronize
    seq = W.replan_sequence(base, args.sequence, device=device)
    base.setup(ctx)
    key = name + ("rb" if args.road_boundary else "")   # (the counter files' key of this workload)
    head = run_record(ctx, base, seq, args.mode, args.steps, args.warmup, args.min_seconds, sync, name=key, caller=args.caller)
    N = base.inputs.params.N
    result = {
        "metric": "candidate trajectories/sec (sample+cost+collision) per replan",
        "value": head["value"], "unit": "candidates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{base.name}: {base.description}", "mode": args.mode,
                   "candidates_per_step": head["candidates_per_step"], "candidates_per_gpu": head["candidates_per_step"],
                   "horizon_steps": N, "n_obstacles": int(base.obstacles.dyn_obb.shape[0] + len(base.obstacles.static_obb)),
                   "inputs": f"replanning sequence of {len(seq)} consecutive cycles (closed loop over the scenario; cfg5: seeded initial states)",
                   "parallelism": "one GPU", "exchange": "none"},
        "timing": dict(head["spread_ms"], what="K-step regions bracketed by device syncs, repeated until min-seconds; ms_per_step = median region",
                       caller=head["caller"]),
        "roofline": head["roofline"],
        # what "value" has meant round by round (rounds 1-2 are not comparable with 3-4 on `value`; the riders keep the old lines)
        "headline_definition": {"since_round": 3, "workload": "cfg3 (BASELINE.json configs[2]), draw mode", "caller": "compiled host loop over the C ABI",
                                "n_gpus_gt_1": "weak scaling",
                                "changed_from": {"rounds": "1-2", "workload": "cfg2 (configs[1]), draw mode", "caller": "Python binding (ctypes)",
                                                 "n_gpus_gt_1": "strong scaling on cfg4"},
                                "old_definition_rides_in": ["configs.cfg2", "python_binding", "strong"]},
    }
    if not args.main_only:
        if args.caller == "c":   # the same regions with the Python binding making the calls (what round 1 and 2 reported)
            py = run_record(ctx, base, seq, args.mode, args.steps, args.warmup, min(args.min_seconds, 0.5), sync, name=key, caller="python")
            result["python_binding"] = {k: py[k] for k in ("ms_per_step", "value", "unit", "caller", "kernel_ms")}
        if args.mode != "fused":
            result["fused_mode"] = run_record(ctx, base, seq, "fused", args.steps, args.warmup, args.min_seconds, sync, name=key)
        ctx.close()
        ctx = None
        if not args.no_configs:
            result["configs"] = side_configs(args, torch, device, skip=base.name)
        result["plan_latency_ms"] = plan_latency(base, device)
        try:
            result["level_loop_latency_ms"] = level_loop_latency(device)
        except Exception as e:   # (a record beside the headline: never the reason for a missing line)
            result["level_loop_latency_ms"] = {"error": repr(e)}
        result["corridor_sampling"] = corridor_sampling_cost(base, device)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(base, with_mode(seq, args.mode)[0], args.cpu_seconds)
    if ctx is not None:
        ctx.close()
    return result


def side_configs(args, torch, device, skip):
    """Draw- and production-mode records of the other configurations of BASELINE.json (cfg2, cfg4), of the stress grid cfg5, of
    cfg2 with its road boundary and of cfg3f -- cfg3's grid in mostly-free traffic (20 % of the feasible candidates collide, 95 % on
    cfg3 proper), where the collision query cannot end at a candidate's first hit -- so that the driver-run line carries them."""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    out = {}
    guess = {"cfg1": 0.03, "cfg2": 0.04, "cfg2rb": 0.05, "cfg3": 0.2, "cfg3rb": 0.2, "cfg3f": 0.2, "cfg3frb": 0.2, "cfg4": 1.4, "cfg4rb": 1.6, "cfg5": 2.5}
    # <name>rb: the scenario's road boundary in the static obstacle tables (thin rectangles along the outer border of the lanelet
    # network, what the reference's collision checker holds by default: reactive_planner.py:246-250) -- cfg2 / cfg4 (T-junction) 85
    # rectangles, cfg3 (DEU_Test) 8.  With the boundary no candidate of cfg3 survives (the reference's own answer,
    # tests/golden/cfg3_ref_rb.npz); cfg3frb -- cfg3's grid, mostly-free traffic -- keeps 30 070 of 61 271.
    for name in ("cfg1", "cfg2", "cfg2rb", "cfg3", "cfg3rb", "cfg3f", "cfg3frb", "cfg4", "cfg4rb", "cfg5"):   # (cfg1: the size of the reference's shipped configurations, N = 20)
        if name == skip:
            continue
        w = W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]()
        ctx = RpContext(device)
        w.setup(ctx)
        seq = W.replan_sequence(w, 16 if name != "cfg5" else 8, device=device)
        w.setup(ctx)
        rec = {"workload": f"{w.name}: {w.description}", "horizon_steps": int(w.inputs.params.N)}
        for mode in ("draw", "fused"):
            k = steps_for(guess[name] * (1.0 if mode == "draw" else 0.5), args.steps)
            rec[mode] = run_record(ctx, w, seq, mode, k, 3, args.min_seconds, torch.cuda.synchronize, name=name)
        ctx.close()
        out[name] = rec
    return out


def plan_latency(base, device, n_replans=220, warm=20):
    """Second half of BASELINE.json's metric: latency of a whole ReactivePlanner.plan() call -- Python boundary, output packing,
    production mode, one sampling level of the workload's grid -- over closed-loop replans of the scenario
    (harness.run_closed_loop, replanning every time step; the loop is restarted from the initial state when the route ends)."""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    from commonroad_rp_amd.harness import run_closed_loop
    import math
    if base.n_candidates > 100000 or base.name == "cfg5":
        return None
    ctx = RpContext(device)

    class Shared:   # the loops share one context (tables stay resident; closing is done here)
        def __new__(cls, dev):
            return ctx
    lat, loops, failed = [], 0, 0
    c = base.inputs.cost
    dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
    while len(lat) < n_replans + warm and loops < 40:
        rp = W.make_planner(base, backend_factory=Shared, device=device)
        res = run_closed_loop(rp, max_steps=80, replanning_frequency=1, desired_velocity=dv)
        lat += res.plan_times if res.completed else res.plan_times[:-1]   # (a failed plan() runs every sampling level: not a replan)
        failed += 0 if res.completed else 1
        loops += 1
    ctx.close()
    lat = np.asarray(lat[warm:]) * 1e3
    if len(lat) == 0:
        return None
    return {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)), "p99": float(np.percentile(lat, 99)),
            "n": int(len(lat)), "loops": loops, "what": "ReactivePlanner.plan() wall time per closed-loop replan (every time step a new "
            "state, velocity grid and time index), one sampling level of the workload's grid, production mode, Python included"}


def level_loop_latency(device, n_replans=220, warm=20):
    """``ReactivePlanner.plan()`` on BASELINE.json configs[0] as the reference drives it (run_planner.py:28-107): ZAM_Over-1_1 with the
    reference's OWN sampling levels (120 / 630 / 3 060 candidates, N = 20, t_min 0.2), its static obstacle and the road boundary, in
    closed loop.  plan() hands the grids of all three levels to the device in one call (rp_plan_levels: one round trip whatever
    level delivers).  Second record: the same scene behind a wall -- no level has a winner, every cycle visits all three."""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    from commonroad_rp_amd.collision import ObstacleTables
    from commonroad_rp_amd.harness import run_closed_loop
    w = W.cfg1(level=1, road_boundary=True)
    ctx = RpContext(device)

    class Shared:
        def __new__(cls, dev):
            return ctx
    out = {}
    wall = ObstacleTables(static_circ=[[float(w.coordinate_system.reference[70][0]), float(w.coordinate_system.reference[70][1]), 60.0]])
    w2 = W.Workload(w.name, w.description, w.inputs, w.coordinate_system, wall)
    for policy in ("adaptive", "chain", "sequential"):   # ReactivePlanner.level_policy (adaptive: the default)
        rec = {}
        lat, loops, levels = [], 0, []
        while len(lat) < n_replans + warm and loops < 40:
            rp = W.make_level_planner(w, t_min=0.2, backend_factory=Shared, device=device)
            rp.level_policy = policy
            res = run_closed_loop(rp, max_steps=60, replanning_frequency=1)
            lat += res.plan_times if res.completed else res.plan_times[:-1]
            loops += 1
        a = np.asarray(lat[warm:]) * 1e3
        if len(a):
            rec["closed_loop"] = {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99)),
                                  "n": int(len(a)), "loops": loops}
        # every level fails: a wall across the road -- three levels per cycle
        rp = W.make_level_planner(w2, t_min=0.2, backend_factory=Shared, device=device)
        rp.level_policy = policy
        t = []
        for _ in range(warm + 100):
            t0 = time.perf_counter()
            r = rp.plan()
            t.append(time.perf_counter() - t0)
            assert r is None
        a = np.asarray(t[warm:]) * 1e3
        rec["all_levels_fail"] = {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "n": int(len(a)),
                                  "first_cycle": float(t[0] * 1e3), "candidates": [120, 630, 3060]}
        out[policy] = rec
    ctx.close()
    out["what"] = ("ReactivePlanner.plan() wall time, ZAM_Over-1_1 with the reference's sampling levels 1-3 (120 / 630 / 3 060 candidates, "
                   "N = 20) + road boundary, production mode, Python included, per ReactivePlanner.level_policy: chain = all levels in one "
                   "rp_plan_levels call every cycle, adaptive (default) = the first level alone, the remaining ones in one call when it fails "
                   "and whole cycles for a while after that, sequential = one call per level; all_levels_fail: a wall across the road")
    return out


def corridor_sampling_cost(base, device, reps=15):
    """A data-dependent sampling space (CorridorSampling, reference sampling.py:273-397: no (T, L, D) product form) on the
    explicit-polynomial entry rp_plan_coeffs: what one sampling level costs through the reference's object interface (one
    TrajectorySample per candidate, coefficients extracted) and through the batch view (coefficient arrays), and the device
    call itself.  Synthetic corridor around the workload's initial state, same time samples as the workload."""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    from commonroad_rp_amd.corridor import ReachBox
    from commonroad_rp_amd.sampling import CorridorSampling
    if base.n_candidates > 100000 or base.name == "cfg5":
        return None
    rp = W.make_planner(base, device=device)
    p = base.inputs.params
    sp = CorridorSampling(rp.config)
    sp.samples_t._dict_level_to_sample_set[1] = set(float(t) for t in base.inputs.T)
    # (21 samples per interval at every level; written into the table itself: set_dict_number_of_samples(dict_level_to_num_samples=...)
    #  only checks the keys, in the reference -- sampling.py:334-338 -- and here)
    sp._dict_level_to_num_samples = {k: 21 for k in range(rp.config.sampling.num_sampling_levels)}
    s0, v0 = p.x0_lon[0], max(p.x0_lon[1], 1.0)
    cor = {}
    for q in range(p.N + 2):
        k = p.time_step0 + q
        t = q * p.dt
        cor[k] = [ReachBox(s0 - 1.0, s0 + 1.6 * v0 * t + 4.0, -2.8, -0.3, 0.5 * v0, 1.3 * v0 + 1.0),
                  ReachBox(s0 + 0.3 * v0 * t, s0 + 1.8 * v0 * t + 6.0, -0.5, 1.4, 0.6 * v0, 1.4 * v0 + 1.0),
                  ReachBox(s0 + 0.8 * v0 * t + 2.0, s0 + 2.0 * v0 * t + 8.0, 2.0, 3.0, 0.9 * v0, 1.5 * v0 + 1.0)]
    sp.driving_corridor = cor
    rp.set_sampling_space(sp)
    x0_lon, x0_lat = rp.x_0_cl
    mode = rp.config.sampling.longitudinal_mode
    t_obj, t_batch, t_dev, t_level, C = [], [], [], [], 0
    ctx = rp._gpu_ctx()
    # (as the planner sets it up in _create_trajectory_bundle: the batch view writes into the context's pinned arena, rp_coeffs_arena)
    sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
    sp.__dict__["_arena_owner"] = rp
    for _ in range(3):   # (a third of a second each: kept out of the loop below, where it would leave the GPU idle between the calls)
        t0 = time.perf_counter()
        trajs = sp.generate_trajectories_at_level(1, x0_lon, x0_lat, mode, False)
        lon = np.array([t.trajectory_long.coeffs for t in trajs])
        lat = np.array([t.trajectory_lat.coeffs for t in trajs])
        lon_T = np.array([t.trajectory_long.delta_tau for t in trajs], dtype=float)
        tl = np.array([len(np.arange(0, np.round(tt + p.dt, 5), p.dt)) for tt in lon_T], dtype=np.int32)
        t_obj.append(time.perf_counter() - t0)
    for _ in range(reps + 5):
        t0 = time.perf_counter()
        blon, blat, bT, btl, _, _ = sp.coeffs_at_level(1, x0_lon, x0_lat, mode, False)
        t_batch.append(time.perf_counter() - t0)
        C = len(bT)
        params = rp._gpu_params(x0_lon, x0_lat, 0)
        cost = rp._gpu_cost()
        t0 = time.perf_counter()
        lg = sp.__dict__.get("_last_groups")   # (as the planner does: groups of exactly these arrays)
        out = ctx.plan_coeffs(params, cost, blon, blat, bT, btl, groups=lg[1] if (lg is not None and lg[0] is bT) else None)
        t_dev.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        rp._get_optimal_trajectory(rp._create_trajectory_bundle(x0_lon, x0_lat, samp_level=1))
        t_level.append(time.perf_counter() - t0)
    rp.close()
    t_batch, t_dev, t_level = t_batch[5:], t_dev[5:], t_level[5:]   # (warm-up)
    ms = lambda v: float(np.median(v) * 1e3)   # noqa: E731
    return {"candidates": C, "objects_ms": ms(t_obj), "batch_view_ms": ms(t_batch), "rp_plan_coeffs_ms": ms(t_dev),
            "one_sampling_level_ms": ms(t_level), "winner": int(out.best_index), "candidates_per_s_device": C / (ms(t_dev) * 1e-3),
            "what": "CorridorSampling on a synthetic 3-lane corridor, 21 samples per interval: candidate generation through the "
                    "reference's object interface vs the batch view (coefficient arrays), rp_plan_coeffs on the device, and the "
                    "planner's whole per-level hot path (batch view + device + winner sample)"}


# --------------------------------------------------------------------------------------------------------------------
class Barrier:
    def __init__(self, torch, dist, device):
        self.dist, self.t = dist, torch.zeros(1, dtype=torch.float64, device=device)

    def __call__(self):
        self.dist.barrier()

    def max_over_ranks(self, v):
        self.t[0] = v
        self.dist.all_reduce(self.t, op=self.dist.ReduceOp.MAX)
        return float(self.t.item())


def run_multi(args, torch, dist, rank, local_rank, world, rehearse):
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    from commonroad_rp_amd.distributed import shard_range, make_exchange
    device = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if rehearse else device   # (rehearsal: gloo group, host tensors)
    bar = Barrier(torch, dist, xdev)
    sync = torch.cuda.synchronize
    ctx = RpContext(local_rank)
    from commonroad_rp_amd.distributed import wait_mode_for_group, local_world_size, cpu_quota
    ctx.set_wait_mode(wait_mode_for_group(local_world_size(dist)))   # (yield between polls of the completion ticket when ranks outnumber CPUs / 2)

    def bcast(obj):
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def sharded(w, seq, mode, steps, transport):
        """every rank evaluates its contiguous range of every cycle's grid and joins the winner exchange"""
        inputs = with_mode(seq, mode)
        n1 = inputs[0].params.N + 1
        ex = make_exchange(dist, xdev, n1, transport)
        kms, t_ex = [], []
        ctx.set_profiling(8)

        def step(k):
            q = inputs[k]
            lo, hi = shard_range(q.n_candidates, rank, world)
            out = ctx.plan(q, lo, hi)
            if out.kernel_ms > 0:
                kms.append(out.kernel_ms)
            t0 = time.perf_counter()
            ex(ctx, out)
            t_ex.append(time.perf_counter() - t0)
        # the mailbox transport through the compiled loop (rp_hostloop_run_sharded: rp_plan on the rank's range + both messages of
        # the exchange, no Python between the steps -- as the N = 1 regions); the collectives need torch between the steps
        region, caller = None, "Python binding (ctypes) + exchange object, per step"
        if args.caller == "c" and type(ex).__name__ == "MailboxExchange":
            from commonroad_rp_amd._capi import HostLoop
            loop = HostLoop(ctx, inputs, ranges=[shard_range(q.n_candidates, rank, world) for q in inputs])
            tot = {"kms": 0.0, "kn": 0, "tex": 0.0, "n": 0}
            caller = "compiled host loop over the C ABI (rp_hostloop.c: rp_plan + rp_mailbox_exchange / rp_mailbox_sum)"

            def region(k0, n):
                st, tex = loop.run_sharded(ex, k0, n)
                tot["kms"] += st.kernel_ms_sum; tot["kn"] += st.kernel_ms_n; tot["tex"] += tex; tot["n"] += n
        regions = measure(step, len(inputs), steps, args.warmup, args.min_seconds, sync, bar, region=region)
        ctx.set_profiling(0)
        sp = spread(regions)
        cand = float(np.mean([q.n_candidates for q in inputs]))
        loc = cand / world
        if region is not None:
            kms = [tot["kms"] / tot["kn"]] if tot["kn"] else []
            t_ex = [tot["tex"] / max(tot["n"], 1)]
        kernel_ms = float(np.mean(kms)) if kms else float("nan")
        return {"mode": mode, "candidates_per_step": cand, "candidates_per_gpu": loc, "ms_per_step": sp["median"],
                "value": cand / (sp["median"] * 1e-3), "unit": "candidates/s", "steps": steps, "spread_ms": sp, "kernel_ms": kernel_ms,
                "caller": caller,
                "exchange": type(ex).__name__, "exchange_ms_per_step": float(np.median(t_ex) * 1e3) if t_ex else None,
                "roofline": roofline_record(w.name, mode, n1, loc, 0.0, kernel_ms, single_gpu=False, kernel=ctx.last_kernel())}

    def sequence(w, n, nL=None):
        """rank 0 drives the closed loop, every rank gets the same inputs"""
        seq = None
        if rank == 0:
            w.setup(ctx)
            s = W.replan_sequence(w, n, device=local_rank, nL=nL)
            seq = [(bytes(q.params), bytes(q.cost), q.T, q.traj_len, q.L, q.D) for q in s]
        seq = bcast(seq)
        from commonroad_rp_amd._capi import PlanInputs, RpParams, RpCost
        return [PlanInputs(RpParams.from_buffer_copy(a), RpCost.from_buffer_copy(b), T, tl, L, D) for a, b, T, tl, L, D in seq]

    # ---- strong scaling: BASELINE.json configs[3], the same grid whatever N
    sname = "cfg4"
    ws = W.WORKLOADS[sname]()
    seq_s = sequence(ws, min(args.sequence, 16))
    ws.setup(ctx)
    k_s = steps_for(1.8 / world if sname == "cfg4" else 0.05, args.steps)
    default_transport = os.environ.get("RP_AMD_EXCHANGE", "auto")
    strong = sharded(ws, seq_s, args.mode, k_s, default_transport)
    other = "collective" if strong["exchange"] == "MailboxExchange" else "mailbox"
    try:
        strong_other = sharded(ws, seq_s, args.mode, k_s, other)
    except Exception as e:   # (e.g. no shared memory for the mailbox)
        strong_other = {"error": f"{type(e).__name__}: {e}"}
    # rank 0 alone on the whole grid: the N = 1 point of the same workload, measured in the same run
    alone = None
    if rank == 0:
        alone = run_record(ctx, ws, seq_s, args.mode, steps_for(1.8, args.steps), 3, args.min_seconds, sync, caller=args.caller)   # (same caller as the sharded steps)
    dist.barrier()
    # ---- weak scaling: every rank a shard the size of the N = 1 headline workload (its longitudinal grid densified N times):
    #      per-GPU work as in `bench.py --gpus 1`, so that the per-N values of a scaling run refer to the same unit of work
    ww = W.WORKLOADS[args.workload or "cfg3"]()
    seq_w = sequence(ww, min(args.sequence, 16), nL=len(ww.inputs.L) * world)
    ww.setup(ctx)
    weak = sharded(ww, seq_w, args.mode, args.steps, default_transport)
    try:
        weak_other = sharded(ww, seq_w, args.mode, args.steps, other)
    except Exception as e:
        weak_other = {"error": f"{type(e).__name__}: {e}"}
    # rank 0 alone on ONE shard's grid (the un-densified workload: exactly what `bench.py --gpus 1` times), in the same run
    seq_1 = sequence(ww, min(args.sequence, 16))
    alone_w = None
    if rank == 0:
        ww.setup(ctx)
        alone_w = run_record(ctx, ww, seq_1, args.mode, args.steps, 3, args.min_seconds, sync, caller=args.caller)
    dist.barrier()
    # ranks RCCL itself carried data for: an all-reduce of ones over the nccl group (0 in a gloo rehearsal)
    rccl_ranks = 0
    if dist.get_backend() == "nccl":
        ones = torch.ones(1, dtype=torch.int32, device=device)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
    ctx_wait_mode = ctx.wait_mode_name() + " (cpu quota %s, affinity %d)" % cpu_quota()
    ctx.close()
    # ---- ReactivePlanner.plan() with the planner's process group set: the sharding decision, the shard's rp_plan and the winner
    #      exchange all happen inside plan() (every rank drives the same closed loop on the strong-scaling workload)
    plan_lat = None
    try:
        plan_lat = sharded_plan_latency(ws, dist, xdev, local_rank, default_transport)
    except Exception as e:
        plan_lat = {"error": f"{type(e).__name__}: {e}"}
    if rank != 0:
        return None
    head, hw, kind = (strong, ws, "strong") if args.scaling == "strong" else (weak, ww, "weak")
    other_rec = strong_other if kind == "strong" else weak_other
    alone_rec = alone if kind == "strong" else alone_w
    brief = lambda r: ({k: r[k] for k in ("value", "ms_per_step", "exchange", "exchange_ms_per_step", "candidates_per_step", "error") if k in r}   # noqa: E731
                       if isinstance(r, dict) else None)
    return {
        "metric": "candidate trajectories/sec (sample+cost+collision) per replan",
        "value": head["value"], "unit": "candidates/s", "n_gpus": world, "steps": head["steps"], "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": kind, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{hw.name}: {hw.description}" + (f", L densified x{world}" if kind == "weak" else ""),
                   "mode": args.mode, "candidates_per_step": head["candidates_per_step"], "candidates_per_gpu": head["candidates_per_gpu"],
                   "horizon_steps": int(hw.inputs.params.N),
                   "inputs": "replanning sequence (closed loop over the scenario, driven by rank 0, same inputs on every rank)",
                   "parallelism": f"candidate-range sharding x{world}", "exchange": head["exchange"]},
        "timing": dict(head["spread_ms"], what="K-step regions bracketed by barrier + device sync, max over ranks, repeated until min-seconds; median region"),
        "roofline": head["roofline"],
        # what a reader of a SCALE record needs beside the headline: how the winner messages travelled and what that cost per step,
        # the same regions over the other transport, how many ranks RCCL itself saw (an all-reduce of ones over the nccl group; a
        # rehearsal's gloo group reports 0), and rank 0 ALONE on one shard's grid in the same run (= `bench.py --gpus 1`'s workload
        # for the weak headline: to be checked against the BENCH record)
        "exchange": head["exchange"], "exchange_ms_per_step": head["exchange_ms_per_step"],
        "other_transport": brief(other_rec),
        "ranks_seen_by_rccl": rccl_ranks,
        "one_gpu_same_grid": brief(alone_rec),
        "strong_scaling": None if kind == "strong" else dict(brief(strong), one_gpu_same_grid_ms=alone["ms_per_step"] if alone else None),
        "wait_mode": ctx_wait_mode,
        "strong": {"sharded": strong, "sharded_other_transport": strong_other, "one_gpu_same_grid": alone},
        "weak": {"sharded": weak, "sharded_other_transport": weak_other, "one_gpu_same_grid": alone_w},
        "plan_latency_ms": plan_lat,
        "rehearsal": bool(rehearse),
    }


def sharded_plan_latency(w, dist, xdev, local_rank, transport, n_replans=60, warm=8):
    """p50 of ``ReactivePlanner.plan()`` on every rank of the group, the planner sharding the level itself
    (``GpuBackendMixin.set_process_group``): same closed loop on every rank, identical inputs, identical decisions."""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd.harness import run_closed_loop
    import math
    c = w.inputs.cost
    dv = None if math.isnan(c.desired_speed) else float(c.desired_speed)
    lat, loops = [], 0
    while len(lat) < n_replans + warm and loops < 8:
        rp = W.make_planner(w, device=local_rank)
        rp.set_process_group(dist, xdev, transport)
        res = run_closed_loop(rp, max_steps=40, replanning_frequency=1, desired_velocity=dv)
        lat += res.plan_times if res.completed else res.plan_times[:-1]
        loops += 1
        rp.close()
    lat = np.asarray(lat[warm:]) * 1e3
    if len(lat) == 0:
        return None
    return {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)), "n": int(len(lat)), "workload": w.name,
            "world": dist.get_world_size(), "shard_min_candidates": int(rp.shard_min_candidates),
            "what": "ReactivePlanner.plan() wall time per closed-loop replan on rank 0, the planner's own sharding (set_process_group): "
                    "shard_range -> rp_plan on the shard -> exchange_winner inside _get_optimal_trajectory"}


def cpu_quota():
    """CPUs the control group of this process may use at a time (cgroup cpu.max / cfs quota), or None without a limit."""
    from commonroad_rp_amd.distributed import cpu_quota as q
    return q()[0]


# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, inp, budget_s: float):
    """The CPU oracle (C port of the reference's algorithm, oracle/rp_oracle.c) timed on one host
    core on the same workload and mode, repeated for about ``budget_s`` seconds."""
    from oracle import oracle
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    C = inp.n_candidates
    sample = min(C, 20000)
    keep = {}   # (the output arrays are reused between the calls: page faults of fresh ones are not part of the algorithm)
    oracle.plan(inp, tb, 0, sample, want_states=True, scratch=keep)   # (library load, page faults of the output arrays)
    t0 = time.perf_counter()
    oracle.plan(inp, tb, 0, sample, want_states=True, scratch=keep)
    one = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(one, 1e-4)))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.plan(inp, tb, 0, sample, want_states=True, scratch=keep)
    el = time.perf_counter() - t0
    out = {"value": sample * reps / el, "unit": "candidates/s", "cores": 1, "kind": "port",
           "sample_short": f"first {sample} candidates of cycle 0, same mode, x{reps}, {el:.1f} s, C port on 1 of {os.cpu_count()} cores",
           "sample": f"first {sample} candidates of the first cycle of the same sequence and mode, {reps} repetitions, "
                     f"{el:.1f} s on 1 of {os.cpu_count()} host cores (C port; the Python reference itself ran "
                     f"~3.9e3 candidates/s/core in the build container, BASELINE.md)"}
    # the same port on the GPU box's CPU share (OpenMP over candidates), a few seconds ("1 core and all cores")
    try:
        affinity = max(1, len(os.sched_getaffinity(0)))   # every core this process may run on ...
    except AttributeError:
        affinity = max(1, os.cpu_count() or 1)
    quota = cpu_quota()                                    # ... of which the container's CPU share lets this many run at a time
    threads = max(1, min(affinity, quota) if quota else affinity)
    if threads > 1:
        big = min(C, 20000 * threads)
        keep = {}
        oracle.plan(inp, tb, 0, big, want_states=True, nthreads=threads, scratch=keep)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < min(3.0, budget_s / 3):
            oracle.plan(inp, tb, 0, big, want_states=True, nthreads=threads, scratch=keep)
            n += 1
        out["all_cores"] = {"value": big * n / (time.perf_counter() - t0), "unit": "candidates/s", "cores": threads,
                            "nproc": os.cpu_count(), "affinity": affinity, "cpu_quota": quota,
                            "what": "the same C port, OpenMP over candidates on every core this process can use: the affinity mask, "
                                    "capped by the CPU share of its control group (cpu.max) -- more threads than that only contend"}
    # the reference's own execution model -- one Python iteration per candidate, NumPy per trajectory, a
    # scalar Python loop over the steps, sort + lazy collision walk (oracle/numpy_loop.py) -- on one core, a few seconds
    from oracle import numpy_loop
    n_py = min(C, 1500)
    numpy_loop.plan(inp, tb, 0, min(C, 50))
    t0 = time.perf_counter()
    reps_py = 0
    while reps_py == 0 or time.perf_counter() - t0 < min(5.0, budget_s / 3):
        numpy_loop.plan(inp, tb, 0, n_py)
        reps_py += 1
    el_py = time.perf_counter() - t0
    out["numpy_loop"] = {"value": n_py * reps_py / el_py, "unit": "candidates/s", "cores": 1, "kind": "port",
                         "sample": f"first {n_py} candidates, {reps_py} repetitions, {el_py:.1f} s: loop-faithful NumPy "
                                   f"restatement of the reference's per-candidate Python loop (lazy collision walk)"}
    # B1 / B0: this restatement timed beside the imported reference on the same inputs in the build
    # container (tests/golden/time_reference.py -> cpu_calibration.json: data only, the reference does not travel)
    cal = load_json(os.path.join(REPO, "tests", "golden", "cpu_calibration.json"))
    if cal.get("ratio_numpy_loop_to_reference"):
        ratio = float(cal["ratio_numpy_loop_to_reference"])
        out["numpy_loop"].update(ratio_to_reference=ratio, reference_value=out["numpy_loop"]["value"] / ratio,
                                 calibration={"file": "tests/golden/cpu_calibration.json", "host_cpu": cal.get("host_cpu"),
                                              "cases": {k: {"reference_candidates_per_s": v["reference_candidates_per_s"],
                                                            "numpy_loop_candidates_per_s": v["numpy_loop_candidates_per_s"]}
                                                        for k, v in cal.get("cases", {}).items()},
                                              "what": "reference_value = this box's numpy_loop rate / (numpy_loop rate / reference rate measured "
                                                      "in the build container on cfg1 level 3 and cfg2, one core)"})
    return out


if __name__ == "__main__":
    main()
