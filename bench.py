#!/usr/bin/env python3
"""Benchmark of the hot path: candidate trajectories / s (sample + convert + check + cost + collide
+ select) per replanning step, through the C ABI of librp_amd.so.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--mode draw|materialize|fused]

One *step* = one ``rp_plan`` call over the whole candidate batch of one replanning cycle (grids and
parameters staged host->device inside the call: a few hundred bytes; tables resident), including
the winner's state block coming back to the host.  N > 1: one process per GPU, every rank evaluates
its contiguous shard of an N-times denser longitudinal grid (weak scaling) and the ranks exchange
one {cost, index, counters, winner block} message per step (commonroad_rp_amd/distributed.py: shared-memory
mailbox between the ranks of one node -- the result blocks are already in pinned host memory --, RCCL
collectives otherwise or with RP_AMD_EXCHANGE=collective); the group, barriers and the max-over-ranks
timing run over torch.distributed (nccl).

Prints ONE JSON line (rank 0).  Modes:
  draw        every candidate fully evaluated (no pre-filter / early exit, the reference's
              draw_traj_set semantics) and all 14 state rows of every candidate written to HBM.
              Default: work per candidate is data independent and the byte count is SURVEY 8(d)'s
              bytes = C*12 + C*112*(N+1) + 112*(N+1).
  materialize production early-exit semantics, state rows of feasible candidates written
  fused       production semantics, 12 B per candidate leave the kernel
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    # Only the JSON line may reach stdout: native libraries (RCCL prints a version banner) write to fd 1
    # directly, so fd 1 is pointed at stderr for the duration of the run and restored for the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--mode", default="draw", choices=["draw", "materialize", "fused"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed region (no production-mode leg, no plan() latency leg, no cpu_baseline): counter passes")
    ap.add_argument("--road-boundary", action="store_true",
                    help="scenario workloads: add the road boundary (thin rectangles, collision.road_boundary_obb) to the obstacle tables")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import torch
    dist = None
    force_dist = os.environ.get("RP_BENCH_FORCE_DIST") == "1"   # rehearse the N > 1 code path on one GPU
    # RP_BENCH_REHEARSE=1: several ranks on ONE GPU (gloo group, every rank on cuda:0) -- a functional rehearsal of
    # the N > 1 path on a one-GPU box; its timings mean nothing
    rehearse = os.environ.get("RP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import (RpContext, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL)
    from commonroad_rp_amd.distributed import shard_range, make_exchange, close_exchanges

    flags = {"draw": FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL, "materialize": FLAG_MATERIALIZE_ALL, "fused": 0}[args.mode]
    base = W.WORKLOADS[args.workload](road_boundary=True) if args.road_boundary else W.WORKLOADS[args.workload]()
    w = W.replicate_for_ranks(base, world)
    p = copy_params(w.inputs.params)
    p.flags |= flags
    inp = PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
    C_total = inp.n_candidates
    lo, hi = shard_range(C_total, rank, world)
    N = p.N

    ctx = RpContext(local_rank)
    w.setup(ctx)
    # HIP events bracket the evaluation kernel of every 4th step of the timed region (each bracket costs ~8 us of
    # stream time on this 20-us kernel; profiles/host_overhead.py), the average is over the sampled launches
    ctx.set_profiling(4)

    exchange = make_exchange(dist, torch.device("cuda", local_rank), N + 1) if dist is not None else None

    def step():
        out = ctx.plan(inp, lo, hi)
        if exchange is not None:
            return exchange(ctx, out), out
        return out, out

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    n_feasible = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, loc = step()
        if loc.kernel_ms > 0:
            kernel_ms.append(loc.kernel_ms)
        n_feasible = loc.n_feasible
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if rehearse else torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (rp_eval_kernel): algorithmic bytes per launch / avg duration
    C_loc = hi - lo
    blk = 112 * (N + 1)
    if args.mode == "draw":
        bytes_per_launch = C_loc * 12 + C_loc * blk
    elif args.mode == "materialize":
        bytes_per_launch = C_loc * 12 + n_feasible * blk
    else:
        bytes_per_launch = C_loc * 12
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else float("nan")

    # HBM bytes per launch from the PMC counters: cannot be collected inside this process (rocprofv3 --pmc wraps the
    # command), so the figure measured for this workload + mode by profiles/collect_pmc.sh is read back
    # (WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md); null if never measured
    traffic = None
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json"))).get(base.name)
        if pmc and pmc.get("mode") == args.mode and world == 1:
            traffic = pmc["traffic_bytes"]
    except Exception:
        traffic = None

    result = {
        "metric": "candidate trajectories/sec (sample+cost+collision) per replan",
        "value": C_total * args.steps / elapsed,
        "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{base.name}: {w.description}", "mode": args.mode, "candidates_per_step": C_total,
                   "candidates_per_gpu": C_loc, "horizon_steps": N,
                   "n_obstacles": int(w.obstacles.dyn_obb.shape[0] + len(w.obstacles.static_obb)),
                   "parallelism": f"candidate-range sharding x{world}",
                   "exchange": ({"MailboxExchange": "shared-memory mailbox (ranks of one node; RP_AMD_EXCHANGE=collective for RCCL)",
                                 "CollectiveExchange": "torch.distributed collectives (RCCL)"}.get(type(exchange).__name__, "?")
                                if exchange is not None else "none")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "rp_eval_kernel",
                     "kernel_ms": k_ms, "bytes_per_launch": bytes_per_launch},
    }

    if rank == 0 and world == 1 and not args.main_only:
        # production-mode rate beside the headline (same workload, early exits, 12 B / candidate)
        if args.mode != "fused":
            pf = copy_params(w.inputs.params)
            inf = PlanInputs(pf, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D)
            for _ in range(5):
                ctx.plan(inf)
            t1 = time.perf_counter()
            kk = []
            for _ in range(max(20, args.steps // 4)):
                km = ctx.plan(inf).kernel_ms
                if km > 0:
                    kk.append(km)
            el = time.perf_counter() - t1
            result["fused_mode"] = {"value": C_total * max(20, args.steps // 4) / el, "unit": "candidates/s",
                                    "ms_per_step": el / max(20, args.steps // 4) * 1e3, "kernel_ms": float(np.mean(kk))}
        # second half of BASELINE.json's metric: p50 latency of a whole planner.plan() call, Python boundary and
        # output packing included (production mode), >= 200 replans after 20 warm-ups
        if base.n_candidates <= 100000:
            rp = W.make_planner(base, device=local_rank)
            for _ in range(20):
                rp.plan()
            lat = []
            for _ in range(200):
                t1 = time.perf_counter()
                res = rp.plan()
                lat.append(time.perf_counter() - t1)
            rp.close()
            result["plan_latency_ms"] = {"p50": float(np.percentile(lat, 50) * 1e3), "p90": float(np.percentile(lat, 90) * 1e3),
                                         "n": len(lat), "planned": res is not None,
                                         "what": "ReactivePlanner.plan() wall time, one sampling level, fused mode"}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w, inp, args.cpu_seconds)
    ctx.close()
    if dist is not None:
        close_exchanges()
        dist.destroy_process_group()
    return json.dumps(result) if rank == 0 else None


def cpu_baseline(w, inp, budget_s: float):
    """The CPU oracle (C port of the reference's algorithm, oracle/rp_oracle.c) timed on one host
    core on the same workload and mode, repeated for about ``budget_s`` seconds."""
    from oracle import oracle
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    C = inp.n_candidates
    sample = min(C, 20000)
    oracle.plan(inp, tb, 0, sample, want_states=True)   # (library load, page faults of the output arrays)
    t0 = time.perf_counter()
    oracle.plan(inp, tb, 0, sample, want_states=True)
    one = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(one, 1e-4)))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.plan(inp, tb, 0, sample, want_states=True)
    el = time.perf_counter() - t0
    out = {"value": sample * reps / el, "unit": "candidates/s", "cores": 1, "kind": "port",
           "sample": f"first {sample} candidates of the same workload and mode, {reps} repetitions, "
                     f"{el:.1f} s on 1 of {os.cpu_count()} host cores (C port; the Python reference itself ran "
                     f"~3.9e3 candidates/s/core in the build container, BASELINE.md)"}
    # the same port on the GPU box's CPU share (OpenMP over candidates), a few seconds ("1 core and all cores")
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        threads = max(1, min(16, os.cpu_count() or 1))
    if threads > 1:
        big = min(C, 20000 * threads)
        oracle.plan(inp, tb, 0, big, want_states=True, nthreads=threads)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < min(3.0, budget_s / 3):
            oracle.plan(inp, tb, 0, big, want_states=True, nthreads=threads)
            n += 1
        out["all_cores"] = {"value": big * n / (time.perf_counter() - t0), "unit": "candidates/s", "cores": threads}
    return out


if __name__ == "__main__":
    main()
