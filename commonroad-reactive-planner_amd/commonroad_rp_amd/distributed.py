"""Multi-GPU sharding of one replanning step.

Candidates are independent given (grids, tables, parameters): every rank holds the (tiny) tables,
evaluates a contiguous range of the reference's list index and the ranks exchange

  1. one all_gather of {cost, index, counters} (12 x 8 B per rank)  -> global (cost, index) min-loc
  2. one all_gather of {colliding-before-winner count, winner block} -> infeasible_count_collision
     and the winner's states on every rank

over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
This replaces the reference's only "communication backend", the multiprocessing.Queue fan-out of
ReactivePlanner._get_optimal_trajectory (commonroad_rp/reactive_planner.py:1084-1111).
Messages are latency-bound (<= 4 KB); no state block ever crosses GPUs except the winner's.
"""
from __future__ import annotations

import numpy as np

from ._capi import N_ARRAYS, PlanOutput


def shard_range(n_candidates: int, rank: int, world_size: int):
    """Contiguous range of the reference list index owned by ``rank``:
    [r * ceil(C / R), min(C, (r + 1) * ceil(C / R)))."""
    per = -(-n_candidates // world_size) if world_size > 0 else n_candidates
    lo = min(n_candidates, rank * per)
    return lo, min(n_candidates, lo + per)


def _pack_head(out: PlanOutput) -> np.ndarray:
    h = np.zeros(12, dtype=np.float64)
    hi = h.view(np.int64)
    h[0] = out.best_cost if out.best_index >= 0 else np.inf
    hi[1] = out.best_index
    hi[2] = out.n_candidates
    hi[3] = out.n_feasible
    hi[4] = out.n_collision
    hi[5:12] = out.reason_counts[0:7]
    return h


def combine_heads(heads: np.ndarray):
    """heads: [world, 12] as produced by ``_pack_head`` -> (owner_rank, cost, index, counters)."""
    hi = heads.view(np.int64)
    best = (-1, np.nan, -1)
    for r in range(heads.shape[0]):
        idx, cost = int(hi[r, 1]), float(heads[r, 0])
        if idx >= 0 and (best[2] < 0 or cost < best[1] or (cost == best[1] and idx < best[2])):
            best = (r, cost, idx)
    return best[0], best[1], best[2], hi[:, 2:12].sum(axis=0)


def exchange_winner(ctx, out: PlanOutput, dist, device) -> PlanOutput:
    """Combine the per-rank results of one sharded ``rp_plan`` into the global result.
    ``ctx`` must still hold the rank's last plan (for the second pass)."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = (out.best_states.shape[1] if out.best_states is not None else ctx._N + 1)
    head = torch.from_numpy(_pack_head(out)).to(device)
    heads = torch.empty((world, 12), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(heads, head) if hasattr(dist, "all_gather_into_tensor") and device.type != "cpu" \
        else dist.all_gather(list(heads.unbind(0)), head)
    owner, gcost, gidx, counters = combine_heads(heads.cpu().numpy().copy())

    # second pass: colliding feasible samples that sort before the global winner
    n_before = ctx.count_collisions_before(gcost if gidx >= 0 else 0.0, gidx)
    tail = np.zeros(1 + 13 + N_ARRAYS * n, dtype=np.float64)
    tail.view(np.int64)[0] = n_before
    if rank == owner and out.best_states is not None:
        tail[1:7] = out.best_lon_coeffs
        tail[7:13] = out.best_lat_coeffs
        tail[13] = out.best_lat_T
        tail[14:] = out.best_states.reshape(-1)
    t = torch.from_numpy(tail).to(device)
    tails = torch.empty((world, tail.size), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(tails, t) if hasattr(dist, "all_gather_into_tensor") and device.type != "cpu" \
        else dist.all_gather(list(tails.unbind(0)), t)
    tails = tails.cpu().numpy().copy()
    total_before = int(tails.view(np.int64)[:, 0].sum())
    reasons = np.zeros(8, dtype=np.int64)
    reasons[0:7] = counters[3:10]
    if gidx < 0:
        return PlanOutput(-1, float("nan"), int(counters[0]), int(counters[1]), total_before, int(counters[2]), reasons,
                          np.full(6, np.nan), np.full(6, np.nan), float("nan"), out.kernel_ms, None)
    w = tails[owner]
    return PlanOutput(gidx, gcost, int(counters[0]), int(counters[1]), total_before, int(counters[2]), reasons,
                      w[1:7].copy(), w[7:13].copy(), float(w[13]), out.kernel_ms, w[14:].reshape(N_ARRAYS, n).copy())
