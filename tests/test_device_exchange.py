"""Device-side winner exchange (rp_result_device / rp_combine_results, commonroad_rp_amd.distributed.CollectiveExchange):
the result blocks the ranks' rp_plan calls leave in device memory are gathered as they are and combined by one kernel.
On the one-GPU box the "ranks" are separate contexts on the same device and the gather is a copy; the collective
itself runs in a single-rank nccl group.  Replaces reactive_planner.py:1084-1111 (multiprocessing.Queue fan-in)."""
import os

import numpy as np
import pytest

from _golden import Golden
from commonroad_rp_amd._capi import FLAG_MATERIALIZE_ALL, PlanInputs, RpContext, copy_params
from commonroad_rp_amd.distributed import _DeviceBlock, local_collisions_before, shard_range

pytestmark = pytest.mark.gpu


def _gather(blocks, torch):
    """[(ptr, nbytes)] -> one device tensor [world][nbytes / 8] (what all_gather_into_tensor produces)"""
    dev = torch.device("cuda", 0)
    views = [torch.as_tensor(_DeviceBlock(p, nb), device=dev) for p, nb in blocks]
    return torch.stack(views).contiguous()


@pytest.mark.parametrize("name", ["arc_hv_l2_obs", "straight_hv_l2_ties", "arc_all_collide", "arc_n70_factor2", "rand_02_hv", "rand_09_hv"])
@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_combine_kernel_equals_unsharded_plan(name, world):
    import torch
    g = Golden(name)
    full_ctx = RpContext(0)
    g.setup_context(full_ctx)
    full = full_ctx.plan(g.inputs)
    C = g.inputs.n_candidates
    ctxs, outs, blocks = [], [], []
    for r in range(world):
        c = RpContext(0)
        g.setup_context(c)
        lo, hi = shard_range(C, r, world)
        outs.append(c.plan(g.inputs, lo, hi))
        ptr, nbytes, rows_ok = c.result_device()
        assert rows_ok and ptr and nbytes == (45 + 14 * (g.inputs.params.N + 1)) * 8
        ctxs.append(c)
        blocks.append((ptr, nbytes))
    msgs = _gather(blocks, torch)
    torch.cuda.synchronize()
    for r in (0, world - 1):     # any rank combines to the same result
        glob, owner, rows_ok = ctxs[r].combine_results(msgs.data_ptr(), world, torch.cuda.current_stream().cuda_stream)
        assert rows_ok
        assert glob.best_index == full.best_index == int(g["winner"])
        assert (glob.n_candidates, glob.n_feasible, glob.n_collision) == (full.n_candidates, full.n_feasible, full.n_collision)
        np.testing.assert_array_equal(glob.reason_counts, full.reason_counts)
        if full.best_index >= 0:
            lo, hi = shard_range(C, owner, world)
            assert lo <= full.best_index < hi
            assert glob.best_cost == full.best_cost
            np.testing.assert_array_equal(glob.best_states, full.best_states)
            np.testing.assert_array_equal(glob.best_lon_coeffs, full.best_lon_coeffs)
            np.testing.assert_array_equal(glob.best_lat_coeffs, full.best_lat_coeffs)
        else:
            assert owner == -1 and glob.best_states is None and np.isnan(glob.best_cost)
        # second message: colliding candidates before the GLOBAL winner
        total = sum(local_collisions_before(ctxs[q], outs[q], glob, q == owner) for q in range(world))
        assert total == full.n_collision_before_best == int(g["n_infeasible_collision"])
    for c in ctxs + [full_ctx]:
        c.close()


def test_rows_on_host_only_are_flagged():
    """A plan that materialises nothing re-evaluates its winner straight into host memory: the header says so, the combine
    step reports it to every rank and does not hand out rows."""
    import torch
    from commonroad_rp_amd import workloads as W
    w = W.cfg3()
    ctx = RpContext(0)
    w.setup(ctx)
    out = ctx.plan(w.inputs)                     # 62 496 candidates x 61 steps: above the auto-materialise limit
    ptr, nbytes, rows_ok = ctx.result_device()
    assert not rows_ok
    msgs = _gather([(ptr, nbytes)], torch)
    torch.cuda.synchronize()
    glob, owner, ok = ctx.combine_results(msgs.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    assert glob.best_index == out.best_index and owner == (0 if out.best_index >= 0 else -1)
    assert ok == (out.best_index < 0)
    # the same plan with state rows kept: rows travel on the device
    p = copy_params(w.inputs.params)
    p.flags |= FLAG_MATERIALIZE_ALL
    out2 = ctx.plan(PlanInputs(p, w.inputs.cost, w.inputs.T, w.inputs.traj_len, w.inputs.L, w.inputs.D))
    ptr, nbytes, rows_ok = ctx.result_device()
    assert rows_ok
    msgs = _gather([(ptr, nbytes)], torch)
    torch.cuda.synchronize()
    glob, owner, ok = ctx.combine_results(msgs.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    assert ok and glob.best_index == out2.best_index == out.best_index
    if out2.best_index >= 0:
        np.testing.assert_array_equal(glob.best_states, out2.best_states)
    ctx.close()


def test_collective_exchange_device_path_single_rank_group():
    """The real collective (RCCL all_gather_into_tensor from the context's device block) in a one-rank nccl group."""
    import torch
    import torch.distributed as dist
    from commonroad_rp_amd.distributed import CollectiveExchange
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for name in ("arc_hv_l2_obs", "arc_all_collide", "rand_06_hv"):
            g = Golden(name)
            ctx = RpContext(0)
            g.setup_context(ctx)
            ex = CollectiveExchange(dist, torch.device("cuda", 0), g.inputs.params.N + 1)
            for _ in range(3):
                out = ctx.plan(g.inputs)
                glob = ex(ctx, out)
                assert glob.best_index == out.best_index and glob.n_collision_before_best == out.n_collision_before_best
                assert (glob.n_feasible, glob.n_collision, glob.n_candidates) == (out.n_feasible, out.n_collision, out.n_candidates)
                if out.best_index >= 0:
                    np.testing.assert_array_equal(glob.best_states, out.best_states)
                    np.testing.assert_array_equal(glob.best_lat_coeffs, out.best_lat_coeffs)
            assert ex.device_path_steps == 3
            # the path is decided once for the group: a result that is not the context's last one is refused (a rank that
            # switched to the host-packed collectives on its own would leave its peers in a different collective)
            stale = ctx.plan(g.inputs)
            ctx.plan(g.inputs, 0, 5)
            with pytest.raises(RuntimeError, match="device path"):
                ex(ctx, stale)
            # ... and a group built without the device path packs on the host
            ex_host = CollectiveExchange(dist, torch.device("cuda", 0), g.inputs.params.N + 1, device_path=False)
            assert ex.device_path and not ex_host.device_path
            glob = ex_host(ctx, stale)
            assert glob.best_index == stale.best_index and ex_host.device_path_steps == 0
            ctx.close()
    finally:
        dist.destroy_process_group()
