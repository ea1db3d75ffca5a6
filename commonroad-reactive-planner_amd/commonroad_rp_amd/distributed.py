"""Multi-GPU sharding of one replanning step.

Candidates are independent given (grids, tables, parameters): every rank holds the (tiny) tables,
evaluates a contiguous range of the reference's list index and the ranks exchange

  1. ONE all_gather of {cost, index, counters, winner coefficients, winner state block} per rank
     (26 + 14 (N+1) doubles, 3.7 KB at N = 30)  -> global lexicographic (cost, index) min-loc, the
     counters and the winner's states on every rank;
  2. only if some rank saw colliding candidates: every rank's count of its colliding feasible candidates
     that sort before the *global* winner (from its own result where that already says it, else a
     second pass on the device, ``rp_count_collisions_before``) and one all_reduce(sum) of 8 bytes
     -> ``infeasible_count_collision``

This replaces the reference's only "communication backend", the multiprocessing.Queue fan-out of
ReactivePlanner._get_optimal_trajectory (commonroad_rp/reactive_planner.py:1084-1111).
Messages are latency-bound; no state block other than a rank's winner ever crosses GPUs.

Two transports carry the same two messages:

  * ``CollectiveExchange``  ``torch.distributed`` collectives (backend "nccl" = RCCL over xGMI on the GPU box,
    "gloo" in CPU tests).  Works across nodes.  On a 40-us replanning step the fixed cost of a device collective
    (pack, H2D, RCCL launch + ring latency, D2H, sync: ~60 us measured on MI355X) is larger than the step.
  * ``MailboxExchange``     ranks of ONE node: every rank's result block already sits in its pinned host memory
    when ``rp_plan`` returns (the kernels write it there), so the ranks post their 3.7 KB messages to a POSIX
    shared-memory mailbox and spin on per-rank sequence words -- one cache-line hand-over per peer instead of a
    device collective.  Chosen automatically when all ranks report the same host (``transport="auto"``);
    ``torch.distributed`` is still what sets the group up and carries barriers.
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
    for r, (c, i) in enumerate(zip(msgs[:, 0].tolist(), mi[:, 1].tolist())):
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


def local_collisions_before(ctx, out: PlanOutput, glob: PlanOutput, is_owner: bool) -> int:
    """This rank's share of ``infeasible_count_collision``: its colliding feasible candidates that sort before the GLOBAL
    winner.  The device pass (``rp_count_collisions_before``, ~18 us) is needed only when the rank's own result does not
    already say it: the owner's local winner IS the global one; without a global winner every collision counts; a rank
    with nothing colliding before its own winner has nothing before the global winner either (which sorts earlier)."""
    if out.n_collision == 0:
        return 0
    if glob.best_index < 0:
        return int(out.n_collision)
    if is_owner:
        return int(out.n_collision_before_best)
    if out.best_index >= 0 and out.n_collision_before_best == 0:
        return 0
    return int(ctx.count_collisions_before(glob.best_cost, glob.best_index))


class MailboxExchange:
    """Shared-memory mailbox of the ranks of one node, driven by the library's host-only entry points
    ``rp_mailbox_exchange`` / ``rp_mailbox_sum`` (include/rp_amd.h, csrc/rp_host.hip): every rank posts its
    ``rp_result`` + winner state block into its slot, spins on the peers' sequence words and combines.  Two slot
    parities alternate between consecutive steps: a rank can run at most one step ahead of the slowest reader,
    so the slot it overwrites two steps later has been read by everyone."""

    def __init__(self, dist, n: int, name_hint: str = ""):
        from multiprocessing import shared_memory
        from . import _capi
        self._capi = _capi
        self._lib = _capi.load_library()
        self.dist, self.n = dist, n
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        size = int(self._lib.rp_mailbox_bytes(self.world, n))
        # Set-up is collective and fails on every rank or on none: a rank that fell back to the collectives on its own
        # would leave its peers waiting here.
        names, self.shm = [None], None
        if self.rank == 0:
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=size)
                self.shm.buf[:size] = bytes(size)
                names = [self.shm.name]
            except OSError:
                self.shm = None
        dist.broadcast_object_list(names, src=0)
        if names[0] is not None and self.rank != 0:
            try:
                self.shm = shared_memory.SharedMemory(name=names[0])
                try:   # the creator unlinks; do not let this process's resource tracker do it as well
                    from multiprocessing import resource_tracker
                    resource_tracker.unregister(self.shm._name, "shared_memory")
                except Exception:
                    pass
            except OSError:
                self.shm = None
        attached = [None] * self.world
        dist.all_gather_object(attached, self.shm is not None)
        if not all(attached):
            self.close()
            raise OSError(f"shared-memory mailbox unavailable on rank(s) {[r for r, ok in enumerate(attached) if not ok]}")
        import ctypes as C
        tmp = C.c_char.from_buffer(self.shm.buf)      # (only to learn the mapping's address: a lasting ctypes export
        self._region = C.c_void_p(C.addressof(tmp))   #  would keep SharedMemory.close() from releasing the buffer)
        del tmp
        self.seq = 0
        self._broken = ""
        self._local, self._glob = _capi.RpResult(), _capi.RpResult()
        self._states = np.empty((N_ARRAYS, n))
        self._owner = C.c_int32(0)
        self._total = C.c_int64(0)
        dist.barrier()   # everyone is attached before the first post

    def close(self):
        try:
            self._region = None
            if self.shm is not None:
                self.shm.close()
                if self.rank == 0:
                    self.shm.unlink()
            self.shm = None
        except Exception:
            pass

    def __call__(self, ctx, out: PlanOutput) -> PlanOutput:
        import ctypes as C
        self.seq += 1
        lo, dp = self._local, self._capi.dptr
        if self._broken:
            raise RuntimeError(f"MailboxExchange: unusable after a time-out ({self._broken})")
        raw = getattr(ctx, "_res", None)
        if raw is not None and out.serial and out.serial == getattr(ctx, "_serial", None):
            # the context still holds the C result of the very call that produced `out`: post it as it is
            raw_states = getattr(ctx, "_last_best", None) if out.best_index >= 0 else None
            rc = self._lib.rp_mailbox_exchange(self._region, self.world, self.rank, self.seq, self.n, C.byref(raw),
                                               dp(raw_states) if raw_states is not None else None, C.byref(self._glob),
                                               dp(self._states), C.byref(self._owner))
            return self._finish(ctx, rc, out)
        lo.best_index, lo.best_cost = out.best_index, out.best_cost
        lo.n_candidates, lo.n_feasible, lo.n_collision = out.n_candidates, out.n_feasible, out.n_collision
        lo.n_collision_before_best = 0
        lo.kernel_ms = out.kernel_ms
        lo.best_lat_T = out.best_lat_T
        for k in range(8):
            lo.reason_counts[k] = int(out.reason_counts[k])
        if out.best_index >= 0:
            for k in range(6):
                lo.best_lon_coeffs[k] = out.best_lon_coeffs[k]
                lo.best_lat_coeffs[k] = out.best_lat_coeffs[k]
        st = out.best_states if (out.best_index >= 0 and out.best_states is not None) else None
        rc = self._lib.rp_mailbox_exchange(self._region, self.world, self.rank, self.seq, self.n, C.byref(lo),
                                           dp(np.ascontiguousarray(st)) if st is not None else None, C.byref(self._glob),
                                           dp(self._states), C.byref(self._owner))
        return self._finish(ctx, rc, out)

    def _timeout(self, what: str, rc: int, item: str) -> TimeoutError:
        """After a time-out the ranks' sequence numbers no longer agree (a late rank posts into a slot nobody reads):
        the mailbox is marked unusable; the caller is expected to tear the process group down."""
        stalled = int(self._lib.rp_mailbox_stalled_rank())
        self._broken = f"{what} -> {rc}: rank {stalled} did not post {item} for exchange {self.seq} within the wait budget " \
                       f"(RP_AMD_MAILBOX_TIMEOUT_S / rp_mailbox_set_timeout)"
        return TimeoutError(f"rank {self.rank}: {self._broken}")

    def _finish(self, ctx, rc: int, out: PlanOutput) -> PlanOutput:
        import ctypes as C
        if rc != 0:
            raise self._timeout("rp_mailbox_exchange", rc, "its result")
        glob = PlanOutput.from_c(self._glob, self._states.copy() if self._glob.best_index >= 0 else None)
        if glob.n_collision > 0:   # second message only when some rank saw a colliding candidate
            n_before = local_collisions_before(ctx, out, glob, int(self._owner.value) == self.rank)
            rc = self._lib.rp_mailbox_sum(self._region, self.world, self.rank, self.seq, self.n, int(n_before), C.byref(self._total))
            if rc != 0:
                raise self._timeout("rp_mailbox_sum", rc, "its count")
            glob.n_collision_before_best = int(self._total.value)
        return glob


def same_host(dist) -> bool:
    """True when every rank of the default group runs on this host."""
    import socket
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, socket.gethostname())
    return all(nm == names[0] for nm in names)


def local_world_size(dist) -> int:
    """ranks of the default group that run on THIS host (they share its CPUs)"""
    import socket
    names = [None] * dist.get_world_size()
    me = socket.gethostname()
    dist.all_gather_object(names, me)
    return sum(1 for nm in names if nm == me)


class _DeviceBlock:
    """A device address range as an object ``torch.as_tensor`` can alias (``__cuda_array_interface__``, float64 words)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class CollectiveExchange:
    """The two messages of one sharded replanning step over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the
    GPU box, "gloo" in CPU tests).

    Device path (GPU group, real ``RpContext``): the result block ``rp_plan`` left in device memory -- header + winner state
    rows -- is the send buffer AS IT IS: ``all_gather_into_tensor`` straight from it (no host packing, no H2D / D2H
    copies), then ``rp_combine_results`` enqueues a one-workgroup kernel behind the collective on the same stream, which
    picks the global winner, sums the counters and writes the combined block to pinned host memory with a completion
    ticket the host spins on.  Host path (CPU groups, oracle-backed test contexts, or a plan whose winner rows exist on
    the host only): messages packed on the host, ``all_gather`` of host tensors (through the device for a GPU group)."""

    def __init__(self, dist, device, n: int, device_path=None):
        import torch
        self.torch = torch
        self.dist, self.device, self.n = dist, device, n
        self.world = dist.get_world_size()
        # Which path a step takes is decided ONCE, for the whole group: the two paths issue different collectives with
        # different buffer sizes, so a rank-local choice per step could leave ranks in mismatched collectives.  (The
        # constructor is collective by contract -- make_exchange.)  A step that cannot take the agreed path raises.
        want = (device.type != "cpu") if device_path is None else bool(device_path and device.type != "cpu")
        flag = torch.tensor([1 if want else 0], dtype=torch.int32, device=device)
        if self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self.device_path = bool(int(flag.item()))
        size = HEAD + N_ARRAYS * n
        pin = device.type != "cpu"
        self.h_send = torch.zeros(size, dtype=torch.float64, pin_memory=pin)
        self.h_recv = torch.zeros((self.world, size), dtype=torch.float64, pin_memory=pin)
        self.d_send = torch.zeros(size, dtype=torch.float64, device=device)
        self.d_recv = torch.zeros((self.world, size), dtype=torch.float64, device=device)
        self.d_cnt = torch.zeros(1, dtype=torch.int64, device=device)
        self._np_send = self.h_send.numpy()
        self._np_recv = self.h_recv.numpy()
        self._blk = None        # (ptr, bytes, send view, gather buffer) of the context's device result block
        self.device_path_steps = 0

    def _device_buffers(self, ptr: int, nbytes: int):
        if self._blk is None or self._blk[0] != ptr or self._blk[1] != nbytes:
            send = self.torch.as_tensor(_DeviceBlock(ptr, nbytes), device=self.device)
            recv = self.torch.empty((self.world, nbytes // 8), dtype=self.torch.float64, device=self.device)
            self._blk = (ptr, nbytes, send, recv)
        return self._blk[2], self._blk[3]

    def __call__(self, ctx, out: PlanOutput) -> PlanOutput:
        dist = self.dist
        glob = owner = None
        if self.device_path:
            if not (hasattr(ctx, "result_device") and out.serial and out.serial == getattr(ctx, "_serial", None)):
                raise RuntimeError(
                    "CollectiveExchange (device path): `out` is not the result of the last call on `ctx` (serial "
                    f"{out.serial} vs {getattr(ctx, '_serial', None)}) or the context has no device result block; the group "
                    "agreed on the device path at construction -- exchange right after plan(), or build the exchange with "
                    "device_path=False on every rank")
            ptr, nbytes, _ = ctx.result_device()
            send, recv = self._device_buffers(ptr, nbytes)
            dist.all_gather_into_tensor(recv, send)
            glob, owner, rows_ok = ctx.combine_results(recv.data_ptr(), self.world, self.torch.cuda.current_stream().cuda_stream)
            if rows_ok:
                self.device_path_steps += 1
            else:   # the owner's rows were on its host only (every rank reads the same flag): host-packed messages this time
                glob = None
        if glob is None:
            pack_result(out, self.n, self._np_send)
            if self.device.type == "cpu":
                dist.all_gather(list(self.h_recv.unbind(0)), self.h_send)
            else:
                self.d_send.copy_(self.h_send, non_blocking=True)
                dist.all_gather_into_tensor(self.d_recv, self.d_send)
                self.h_recv.copy_(self.d_recv, non_blocking=False)
            glob, owner = combine_results(self._np_recv, self.n)
        glob.kernel_ms = out.kernel_ms
        if glob.n_collision > 0:   # second message only when some rank saw a colliding candidate
            n_before = local_collisions_before(ctx, out, glob, owner == dist.get_rank())
            self.d_cnt.fill_(int(n_before))
            dist.all_reduce(self.d_cnt, op=dist.ReduceOp.SUM)
            glob.n_collision_before_best = int(self.d_cnt.item())
        return glob


WinnerExchange = CollectiveExchange   # former name

_exchanges = {}


def exchange_winner(ctx, out: PlanOutput, dist, device, transport: str = "auto") -> PlanOutput:
    """Combine the per-rank results of one sharded ``rp_plan`` into the global result.
    ``ctx`` must still hold the rank's last plan (for the optional second pass).
    ``transport``: "mailbox" (ranks of one node), "collective" (``torch.distributed``), or "auto".
    The choice is collective on first use (all ranks must pass the same value)."""
    n = (out.best_states.shape[1] if out.best_states is not None else ctx._N + 1)
    return make_exchange(dist, device, n, transport)(ctx, out)


def cpu_quota():
    """CPUs the control group of this process may use at a time (cgroup v2 cpu.max / v1 cfs quota) -- ``None`` without a limit --
    and the size of its affinity mask."""
    import os
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else max(1, int(round(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else max(1, int(round(q / per)))
        except Exception:
            quota = None
    try:
        affinity = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        affinity = max(1, os.cpu_count() or 1)
    return quota, affinity


def wait_mode_for_group(local_world: int) -> int:
    """How the ranks of one node should wait for their plans (``rp_set_wait_mode``): every rank's host thread polls the completion
    ticket of its plan -- a busy core per rank -- which is fine while there are two CPUs per rank (one for the poll, one for the
    binding and the exchange); with fewer (eight ranks under a 16-CPU quota is the edge) the polls give their core away between
    looks (``WAIT_YIELD``).  ``RP_AMD_WAIT_MODE`` in the environment overrides (read by rp_create)."""
    import os
    from . import _capi
    if os.environ.get("RP_AMD_WAIT_MODE"):
        return int(os.environ["RP_AMD_WAIT_MODE"])
    quota, affinity = cpu_quota()
    cpus = min(quota, affinity) if quota else affinity
    return _capi.WAIT_YIELD if cpus < 2 * max(1, int(local_world)) else _capi.WAIT_SPIN


def make_exchange(dist, device, n: int, transport: str = "auto"):
    """The exchange object itself (callable ``ex(ctx, out) -> PlanOutput``), for loops that should not pay the
    lookup of ``exchange_winner`` per step.  Collective: every rank must call it with the same arguments."""
    import os
    transport = os.environ.get("RP_AMD_EXCHANGE", transport)
    key = (id(dist), str(device), n, dist.get_world_size(), transport)
    ex = _exchanges.get(key)
    if ex is None:
        if transport == "mailbox" or (transport == "auto" and same_host(dist)):
            try:
                ex = MailboxExchange(dist, n)
            except (OSError, ImportError):
                if transport == "mailbox":
                    raise
        if ex is None:
            ex = CollectiveExchange(dist, device, n)
        _exchanges[key] = ex
    return ex


def close_exchanges():
    """Release the shared-memory mailboxes (call before ``destroy_process_group``)."""
    for ex in _exchanges.values():
        if hasattr(ex, "close"):
            ex.close()
    _exchanges.clear()
