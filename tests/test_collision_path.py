"""rp_set_collision_path / rp_last_path / rp_last_kernel (include/rp_amd.h): which candidates carry a collision label is a setting of the
context with a default that depends on the SEQUENCE of plans only -- the same calls label the same candidates in every run."""
import numpy as np
import pytest

from commonroad_rp_amd import _capi
from commonroad_rp_amd._capi import FLAG_DRAW_ALL, PlanInputs, copy_params

pytestmark = pytest.mark.gpu


def _draw(inp: PlanInputs) -> PlanInputs:
    p = copy_params(inp.params)
    p.flags |= FLAG_DRAW_ALL
    return PlanInputs(p, inp.cost, inp.T, inp.traj_len, inp.L, inp.D)


@pytest.fixture(autouse=True)
def _no_path_switches():
    _capi.set_default_options(None)   # (the library's own launch policy: no option pinned by an earlier test)


def test_collision_path_is_a_setting_of_the_context():
    """cfg3f (62 496 candidates in mostly-free traffic: the cost-ordered stage finds the winner in its first round)."""
    from oracle import oracle
    from commonroad_rp_amd import workloads as W
    w = W.cfg3f()
    tb = oracle.OracleTables.from_coordinate_system(w.coordinate_system, w.obstacles)
    o = oracle.plan(w.inputs, tb, want_states=False, nthreads=8)
    ctx = _capi.RpContext(0)
    try:
        w.setup(ctx)
        seen = {}
        for mode in (_capi.COLLISION_AUTO, _capi.COLLISION_EAGER, _capi.COLLISION_COST_ORDERED, _capi.COLLISION_AUTO):
            ctx.set_collision_path(mode)
            out = ctx.plan(w.inputs)
            status, cost = ctx.fetch_status()
            lab = status & 3
            assert out.best_index == o.out.best_index and out.n_collision_before_best == o.out.n_collision_before_best
            assert out.n_feasible == o.out.n_feasible
            if mode == _capi.COLLISION_EAGER:
                assert ctx.last_path() == 0
                np.testing.assert_array_equal(lab, o.status & 3)          # every colliding candidate carries the label
                assert out.n_collision == o.out.n_collision
            else:
                assert ctx.last_path() in (1, 3)
                diff = lab != (o.status & 3)                                # the ones the stage never looked at keep FEASIBLE
                assert np.all(((o.status & 3)[diff] == 3) & (lab[diff] == 1))
                assert o.out.n_collision_before_best <= out.n_collision <= o.out.n_collision
            seen[mode] = lab.copy()
            # the stage's costs-only first pass over a batch of this size: one lane per candidate and step block; with the query at this size: lanes over steps
            assert ctx.last_kernel() == ("rp_eval_kernel" if mode == _capi.COLLISION_EAGER else "rp_chunk_kernel")
        # a plan that keeps state rows answers the query for every pose whatever the setting, with the lanes-over-steps kernel
        ctx.set_collision_path(_capi.COLLISION_COST_ORDERED)
        out = ctx.plan(_draw(w.inputs))
        assert ctx.last_path() == 0 and ctx.last_kernel() == "rp_eval_kernel" and out.n_collision == o.out.n_collision
        with pytest.raises(_capi.RpError):
            ctx.set_collision_path(17)
    finally:
        ctx.close()


def test_default_collision_path_depends_on_the_sequence_of_plans_only():
    """cfg3 (97 % of the candidates collide: the stage runs dry, the eager kernel decides and the next plans stay eager for 1, 2, 4 ..
    plans) and cfg3f in turn: two contexts that see the same calls take the same path at every plan and label the same candidates."""
    from commonroad_rp_amd import workloads as W
    a, b = W.cfg3(), W.cfg3f()
    runs = []
    for _ in range(2):
        ctx = _capi.RpContext(0)
        try:
            paths, labels = [], []
            for w, reps in ((b, 2), (a, 4), (b, 3), (a, 3)):
                w.setup(ctx)
                for _ in range(reps):
                    ctx.plan(w.inputs)
                    paths.append(ctx.last_path())
                    labels.append(ctx.fetch_status()[0] & 3)
            runs.append((paths, labels))
        finally:
            ctx.close()
    assert runs[0][0] == runs[1][0]
    assert 2 in runs[0][0] and 0 in runs[0][0] and 1 in runs[0][0]      # stage exhausted, eager while backing off, cost-ordered
    for x, y in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(x, y)
