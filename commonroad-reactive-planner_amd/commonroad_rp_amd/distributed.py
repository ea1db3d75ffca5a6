"""Multi-GPU sharding of one replanning step.

Candidates are independent given (grids, tables, parameters): every rank holds the (tiny) tables,
evaluates a contiguous range of the reference's list index and the ranks exchange

  1. ONE all_gather of {cost, index, counters, winner coefficients, winner state block} per rank
     (26 + 14 (N+1) doubles, 3.7 KB at N = 30)  -> global lexicographic (cost, index) min-loc, the
     counters and the winner's states on every rank;
  2. only if some rank saw colliding candidates: a second pass on every rank (colliding feasible
     candidates that sort before the *global* winner, ``rp_count_collisions_before``) and one
     all_reduce(sum) of 8 bytes -> ``infeasible_count_collision``

over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
This replaces the reference's only "communication backend", the multiprocessing.Queue fan-out of
ReactivePlanner._get_optimal_trajectory (commonroad_rp/reactive_planner.py:1084-1111).
Messages are latency-bound; no state block other than a rank's winner ever crosses GPUs.
"""
from __future__ import annotations

import numpy as np

from ._capi import N_ARRAYS, PlanOutput

HEAD = 26   # cost, index, n_candidates, n_feasible, n_collision, reasons[0..6], lon[6], lat[6], lat_T, pad


def shard_range(n_candidates: int, rank: int, world_size: int):
    """Contiguous range of the reference list index owned by ``rank``:
    [r * ceil(C / R), min(C, (r + 1) * ceil(C / R)))."""
    per = -(-n_candidates // world_size) if world_size > 0 else n_candidates
    lo = min(n_candidates, rank * per)
    return lo, min(n_candidates, lo + per)


def pack_result(out: PlanOutput, n: int, buf: np.ndarray = None) -> np.ndarray:
    """One rank's message: 26 header doubles (integers stored bit-exactly through an int64 view)
    followed by its local winner's [14][n] state block."""
    m = buf if buf is not None else np.zeros(HEAD + N_ARRAYS * n, dtype=np.float64)
    mi = m.view(np.int64)
    m[0] = out.best_cost if out.best_index >= 0 else np.inf
    mi[1] = out.best_index
    mi[2] = out.n_candidates
    mi[3] = out.n_feasible
    mi[4] = out.n_collision
    mi[5:12] = out.reason_counts[0:7]
    if out.best_index >= 0 and out.best_states is not None:
        m[12:18] = out.best_lon_coeffs
        m[18:24] = out.best_lat_coeffs
        m[24] = out.best_lat_T
        m[HEAD:] = out.best_states.reshape(-1)
    return m


def combine_results(msgs: np.ndarray, n: int):
    """msgs: [world, HEAD + 14 n] -> (global PlanOutput without the collision-before count, owner rank)."""
    mi = msgs.view(np.int64)
    owner, cost, idx = -1, np.nan, -1
    for r in range(msgs.shape[0]):
        i, c = int(mi[r, 1]), float(msgs[r, 0])
        if i >= 0 and (idx < 0 or c < cost or (c == cost and i < idx)):
            owner, cost, idx = r, c, i
    counters = mi[:, 2:12].sum(axis=0)
    reasons = np.zeros(8, dtype=np.int64)
    reasons[0:7] = counters[3:10]
    if idx < 0:
        out = PlanOutput(-1, float("nan"), int(counters[0]), int(counters[1]), 0, int(counters[2]), reasons,
                         np.full(6, np.nan), np.full(6, np.nan), float("nan"), 0.0, None)
    else:
        w = msgs[owner]
        out = PlanOutput(idx, cost, int(counters[0]), int(counters[1]), 0, int(counters[2]), reasons, w[12:18].copy(),
                         w[18:24].copy(), float(w[24]), 0.0, w[HEAD:].reshape(N_ARRAYS, n).copy())
    return out, owner


class WinnerExchange:
    """Persistent buffers + the two collectives of one sharded replanning step."""

    def __init__(self, dist, device, n: int):
        import torch
        self.dist, self.device, self.n = dist, device, n
        self.world = dist.get_world_size()
        size = HEAD + N_ARRAYS * n
        pin = device.type != "cpu"
        self.h_send = torch.zeros(size, dtype=torch.float64, pin_memory=pin)
        self.h_recv = torch.zeros((self.world, size), dtype=torch.float64, pin_memory=pin)
        self.d_send = torch.zeros(size, dtype=torch.float64, device=device)
        self.d_recv = torch.zeros((self.world, size), dtype=torch.float64, device=device)
        self.d_cnt = torch.zeros(1, dtype=torch.int64, device=device)
        self._np_send = self.h_send.numpy()
        self._np_recv = self.h_recv.numpy()

    def __call__(self, ctx, out: PlanOutput) -> PlanOutput:
        dist = self.dist
        pack_result(out, self.n, self._np_send)
        if self.device.type == "cpu":
            dist.all_gather(list(self.h_recv.unbind(0)), self.h_send)
        else:
            self.d_send.copy_(self.h_send, non_blocking=True)
            dist.all_gather_into_tensor(self.d_recv, self.d_send)
            self.h_recv.copy_(self.d_recv, non_blocking=False)
        glob, _owner = combine_results(self._np_recv, self.n)
        glob.kernel_ms = out.kernel_ms
        if glob.n_collision > 0:   # second pass only when some rank saw a colliding candidate
            n_before = ctx.count_collisions_before(glob.best_cost if glob.best_index >= 0 else 0.0, glob.best_index)
            self.d_cnt.fill_(int(n_before))
            dist.all_reduce(self.d_cnt, op=dist.ReduceOp.SUM)
            glob.n_collision_before_best = int(self.d_cnt.item())
        return glob


_exchanges = {}


def exchange_winner(ctx, out: PlanOutput, dist, device) -> PlanOutput:
    """Combine the per-rank results of one sharded ``rp_plan`` into the global result.
    ``ctx`` must still hold the rank's last plan (for the optional second pass)."""
    n = (out.best_states.shape[1] if out.best_states is not None else ctx._N + 1)
    key = (id(dist), str(device), n, dist.get_world_size())
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = WinnerExchange(dist, device, n)
    return ex(ctx, out)
