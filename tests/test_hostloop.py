"""The compiled host loop over the C ABI (csrc/rp_hostloop.c, _capi.HostLoop) that bench.py times its regions through: it
must be the same sequence of ``rp_plan`` calls as one ``ctx.plan`` per step of the Python binding."""
import ctypes as C
import os

import numpy as np
import pytest

from commonroad_rp_amd import _capi


def test_hostloop_library_loads_and_exports():
    if not os.path.exists(_capi.HOSTLOOP_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_capi.HOSTLOOP_PATH)
    assert hasattr(lib, "rp_hostloop_run")
    assert C.sizeof(_capi.RpHostLoopStats) == 8 * 8


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("cfg2", "draw"), ("cfg2", "fused"), ("cfg3", "fused")])
def test_hostloop_runs_the_same_plans_as_the_binding(name, mode):
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext, HostLoop, PlanInputs, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
    w = W.WORKLOADS[name]()
    ctx = RpContext(0)
    w.setup(ctx)
    seq = W.replan_sequence(w, 6, device=0)
    w.setup(ctx)
    inputs = []
    for q in seq:
        p = copy_params(q.params)
        p.flags = (p.flags & ~(FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL)) | ((FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL) if mode == "draw" else 0)
        inputs.append(PlanInputs(p, q.cost, q.T, q.traj_len, q.L, q.D))
    outs = [ctx.plan(q) for q in inputs]
    loop = HostLoop(ctx, inputs)
    st = loop.run(0, len(inputs))
    assert st.feasible_sum == sum(o.n_feasible for o in outs)
    assert st.winners == sum(o.best_index >= 0 for o in outs)
    assert sum(st.paths) == len(inputs)
    # the context holds the last step's result, as after ctx.plan(inputs[-1])
    assert ctx._res.best_index == outs[-1].best_index
    assert ctx._res.best_cost == outs[-1].best_cost or (np.isnan(ctx._res.best_cost) and np.isnan(outs[-1].best_cost))
    if outs[-1].best_index >= 0:
        np.testing.assert_array_equal(loop.best_states(), outs[-1].best_states)
    # a partial cycle, starting in the middle of it
    st2 = loop.run(4, 5)     # inputs 4, 5, 0, 1, 2
    assert st2.feasible_sum == sum(outs[i].n_feasible for i in (4, 5, 0, 1, 2))
    ctx.close()


class _SoloGroup:
    """the part of torch.distributed a one-rank MailboxExchange needs"""
    def get_world_size(self): return 1
    def get_rank(self): return 0
    def broadcast_object_list(self, objs, src=0): return None
    def all_gather_object(self, out, obj): out[0] = obj
    def barrier(self): return None


@pytest.mark.gpu
def test_sharded_hostloop_is_plan_plus_exchange():
    """rp_hostloop_run_sharded (what bench.py --gpus N times with the mailbox transport): per step the result the Python path
    gets from ctx.plan(range) + MailboxExchange -- winner, counters, the second message's count"""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext, HostLoop, PlanInputs, copy_params
    from commonroad_rp_amd.distributed import MailboxExchange
    w = W.WORKLOADS["cfg2"]()
    ctx = RpContext(0)
    w.setup(ctx)
    seq = W.replan_sequence(w, 5, device=0)
    w.setup(ctx)
    inputs = [PlanInputs(copy_params(q.params), q.cost, q.T, q.traj_len, q.L, q.D) for q in seq]
    # (a one-rank group: the range is the whole grid for some inputs, a part of it for others -- the exchange does not care)
    ranges = [(0, q.n_candidates) if k % 2 == 0 else (q.n_candidates // 3, q.n_candidates) for k, q in enumerate(inputs)]
    ex = MailboxExchange(_SoloGroup(), inputs[0].params.N + 1)
    want = []
    for q, (lo, hi) in zip(inputs, ranges):
        out = ctx.plan(q, lo, hi)
        want.append(ex(ctx, out))
    loop = HostLoop(ctx, inputs, ranges=ranges)
    for k in range(len(inputs)):
        st, t_ex = loop.run_sharded(ex, k, 1)
        g = ex._glob
        assert g.best_index == want[k].best_index and g.n_feasible == want[k].n_feasible
        assert g.n_collision == want[k].n_collision and g.n_collision_before_best == want[k].n_collision_before_best
        assert g.best_cost == want[k].best_cost or (np.isnan(g.best_cost) and np.isnan(want[k].best_cost))
        if g.best_index >= 0:
            np.testing.assert_array_equal(ex._states, want[k].best_states)
        assert t_ex > 0.0 and st.winners == (1 if g.best_index >= 0 else 0)
    st, _ = loop.run_sharded(ex, 0, 2 * len(inputs))
    assert st.winners == 2 * sum(o.best_index >= 0 for o in want)
    ex.close()
    ctx.close()
