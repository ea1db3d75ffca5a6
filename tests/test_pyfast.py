"""The binding's extension module for the cycle's one foreign call (csrc/rp_pyfast.c -> commonroad_rp_amd/_rpfast.so,
``RpContext.plan_packed_fast``): what it hands to ``rp_plan`` (CPU: a recording stand-in for the entry, no device) and that a plan
through it equals the ctypes path's (GPU)."""
import ctypes as C

import numpy as np
import pytest

from commonroad_rp_amd import _capi
from commonroad_rp_amd._capi import RpParams, RpCost, RpGrids, RpResult, N_ARRAYS, PLAN_PACKED


def test_extension_is_built_against_this_header():
    assert _capi._rpfast is not None, "commonroad_rp_amd/_rpfast.so missing or built against another rp_amd.h (make -C commonroad-reactive-planner_amd/csrc)"
    assert _capi._rpfast.ABI_VERSION == _capi.ABI_VERSION
    assert _capi._rpfast.SIZEOF_PARAMS == C.sizeof(RpParams) and _capi._rpfast.SIZEOF_RESULT == C.sizeof(RpResult)


PLAN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RpParams), C.POINTER(RpCost), C.POINTER(RpGrids), C.c_int64, C.c_int64, C.c_uint32,
                      C.POINTER(RpResult), C.POINTER(C.c_double))


def _packed(seen, rc=0, nbytes=32768):
    buf = (C.c_char * nbytes)()
    res = RpResult()

    def fake(ctx, p, cost, g, lo, hi, flags, result, out):
        p, g = p.contents, g.contents
        nd = g.nT + g.nL + g.nD
        f64 = np.frombuffer(buf, dtype=np.float64)
        i32 = np.frombuffer(buf, dtype=np.int32)
        seen.update(ctx=ctx, lo=lo, hi=hi, flags=flags, dims=(g.struct_size, g.nT, g.nL, g.nD), null_arrays=not (g.T or g.L or g.D or g.traj_len),
                    T=f64[:g.nT].copy(), L=f64[g.nT:g.nT + g.nL].copy(), D=f64[g.nT + g.nL:nd].copy(), traj_len=i32[2 * nd:2 * nd + g.nT].copy(),
                    params=(p.struct_size, p.N, p.time_step0, p.low_vel_mode, p.flags, tuple(p.x0_lon), tuple(p.x0_lat), p.x0_orientation, p.dt),
                    cost_kind=cost.contents.kind, result_is_ours=C.addressof(result.contents) == C.addressof(res))
        result.contents.best_index = 7
        out[0] = 42.0
        out[(N_ARRAYS + 13) * (p.N + 1) - 1] = 43.0
        return rc
    fn = PLAN_FN(fake)
    pk = _capi._rpfast.Packed(C.cast(fn, C.c_void_p).value, 0x1234, C.addressof(buf), nbytes, C.addressof(res))
    return pk, res, (fn, buf)   # (keep the callback and the buffer alive)


def test_what_the_extension_hands_to_rp_plan():
    seen = {}
    pk, res, keep = _packed(seen)
    p, cost = RpParams(), RpCost()
    p.N, p.dt, cost.kind = 30, 0.1, 1
    T, tl = np.array([1.0, 2.0, 3.0]), np.array([11, 21, 31], dtype=np.int32)
    L, D = np.array([5.0, 6.0]), np.array([-1.0, 0.0, 1.0, 0.25])
    T.flags.writeable = False   # (the sampling space hands out read-only arrays)
    out = np.zeros((N_ARRAYS + 13, 31))
    rc = pk.plan(p, cost, T, tl, L, D, out, 17, True, 0, [1.0, 2.0, 3.0], (4.0, np.float64(5.0), 6.0), 0.5)
    assert rc == 0
    assert seen["ctx"] == 0x1234 and (seen["lo"], seen["hi"], seen["flags"]) == (0, -1, PLAN_PACKED)
    assert seen["dims"] == (C.sizeof(RpGrids), 3, 2, 4) and seen["null_arrays"]
    assert np.array_equal(seen["T"], T) and np.array_equal(seen["L"], L) and np.array_equal(seen["D"], D) and np.array_equal(seen["traj_len"], tl)
    assert seen["params"] == (C.sizeof(RpParams), 30, 17, 1, 0, (1.0, 2.0, 3.0), (4.0, 5.0, 6.0), 0.5, 0.1)
    assert seen["cost_kind"] == 1 and seen["result_is_ours"]
    assert res.best_index == 7 and out[0, 0] == 42.0 and out[-1, -1] == 43.0
    # the fields are written into the CALLER's struct (the planner reads them back nowhere, rp_plan_wait's users do)
    assert (p.time_step0, p.low_vel_mode, tuple(p.x0_lat)) == (17, 1, (4.0, 5.0, 6.0))


def test_sets_go_in_their_own_iteration_order():
    """L and D as the sample SETS: CPython's iteration order, D united with the current lateral offset as set.union does (sampling.py:218-226)"""
    rng = np.random.default_rng(3)
    for trial in range(200):
        seen = {}
        pk, res, keep = _packed(seen)
        p, cost = RpParams(), RpCost()
        p.N = 4
        T, tl = np.array([1.0, 2.0]), np.array([5, 5], dtype=np.int32)
        nv, nd_ = int(rng.integers(1, 40)), 2 ** int(rng.integers(1, 6)) + 1
        v_set = set(np.linspace(float(rng.uniform(0, 5)), float(rng.uniform(6, 30)), nv))
        d_set = set(np.linspace(-float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3)), nd_))
        d0 = float(rng.choice(sorted(d_set))) if trial % 3 == 0 else float(rng.normal(0, 1))
        out = np.zeros((N_ARRAYS + 13, 5))
        assert pk.plan(p, cost, T, tl, v_set, d_set, out, 0, False, 0, (1.0, 2.0, 3.0), [d0, 0.0, 0.0], 0.0) == 0
        u = d_set.union((d0,))
        assert np.array_equal(seen["L"], np.fromiter(v_set, np.float64, len(v_set)))
        assert np.array_equal(seen["D"], np.fromiter(u, np.float64, len(u))), (trial, d0)
        assert np.array_equal(seen["traj_len"], tl) and seen["dims"][1:] == (2, len(v_set), len(u))
        assert d0 in d_set or len(u) == len(d_set) + 1
        assert len(d_set) == nd_   # (the caller's set is not touched)
    # frozenset, a set too large for the buffer, a set with something that is no number
    seen = {}
    pk, res, keep = _packed(seen)
    p, cost = RpParams(), RpCost()
    p.N = 4
    T, tl, out = np.array([1.0]), np.array([5], dtype=np.int32), np.zeros((N_ARRAYS + 13, 5))
    assert pk.plan(p, cost, T, tl, frozenset((3.0,)), {0.5}, out, 0, False, 0, (0.0,) * 3, (0.5, 0.0, 0.0), 0.0) == 0
    assert list(seen["L"]) == [3.0] and list(seen["D"]) == [0.5]
    with pytest.raises(ValueError):
        pk.plan(p, cost, T, tl, set(np.arange(5000.0)), {0.5}, out, 0, False, 0, (0.0,) * 3, (0.5, 0.0, 0.0), 0.0)
    with pytest.raises(TypeError):
        pk.plan(p, cost, T, tl, {1.0, "a"}, {0.5}, out, 0, False, 0, (0.0,) * 3, (0.5, 0.0, 0.0), 0.0)
    assert pk.plan(p, cost, T, tl, set(), {0.5}, out, 0, False, 0, (0.0,) * 3, (0.5, 0.0, 0.0), 0.0) == -1   # (RP_EINVAL: an empty bundle is the caller's to skip)


def test_error_codes_and_bad_arguments():
    seen = {}
    pk, res, keep = _packed(seen, rc=-3)
    p, cost = RpParams(), RpCost()
    p.N = 4
    T, tl, L, D = np.array([1.0]), np.array([5], dtype=np.int32), np.array([1.0]), np.array([0.0])
    out = np.zeros((N_ARRAYS + 13, 5))
    args = lambda **kw: tuple({**dict(p=p, cost=cost, T=T, tl=tl, L=L, D=D, out=out, t0=0, low=False, flags=0, lon=(0.0, 0.0, 0.0), lat=(0.0, 0.0, 0.0), th=0.0), **kw}.values())  # noqa: E731
    assert pk.plan(*args()) == -3          # (the entry's code comes back as it is: the binding turns it into RpError)
    with pytest.raises(TypeError):
        pk.plan(*args(T=np.array([1.0], dtype=np.float32)))
    with pytest.raises(TypeError):
        pk.plan(*args(tl=np.array([5], dtype=np.int64)))
    with pytest.raises(ValueError):
        pk.plan(*args(tl=np.array([5, 6], dtype=np.int32)))
    with pytest.raises(ValueError):
        pk.plan(*args(out=np.zeros(10)))
    with pytest.raises(ValueError):
        pk.plan(*args(lon=(0.0, 0.0)))
    with pytest.raises((TypeError, ValueError)):
        pk.plan(*args(T=np.zeros((2, 2))))
    with pytest.raises(ValueError):
        pk.plan(*args(p=(C.c_char * 8)()))
    with pytest.raises(TypeError):
        pk.plan(p, cost)
    big = np.zeros(5000)
    with pytest.raises(ValueError):        # (grids beyond the context's 32-KB buffer: the ctypes path raises the same way)
        pk.plan(*args(L=big))
    out_ro = np.zeros((N_ARRAYS + 13, 5)); out_ro.flags.writeable = False
    with pytest.raises((BufferError, ValueError, TypeError)):
        pk.plan(*args(out=out_ro))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg1_ref_l3", "cfg2_ref", "cfg2_ref_rb"])
def test_fast_call_equals_the_ctypes_call(name):
    from _golden import Golden
    from commonroad_rp_amd._capi import RpContext, copy_params
    g = Golden(name)
    ctx = RpContext(0)
    assert ctx.plan_packed_fast is not None, "_rpfast.so did not travel with the package"
    g.setup_context(ctx)
    q = g.inputs
    p = q.params
    res, blk, buf = ctx.plan_packed(copy_params(p), q.cost, q.T, q.traj_len, q.L, q.D)
    ref = (res.best_index, res.best_cost, res.n_feasible, res.n_collision_before_best, list(res.reason_counts), blk.copy(), buf.copy())
    blank = copy_params(p)
    blank.time_step0, blank.low_vel_mode, blank.x0_orientation = -5, 1 - p.low_vel_mode, 9.0
    blank.x0_lon[0] = blank.x0_lat[0] = 1e9
    res, blk2, buf2 = ctx.plan_packed_fast(blank, q.cost, q.T, q.traj_len, q.L, q.D, p.time_step0, bool(p.low_vel_mode), p.flags, list(p.x0_lon), list(p.x0_lat),
                                           p.x0_orientation)
    assert (res.best_index, res.best_cost, res.n_feasible, res.n_collision_before_best, list(res.reason_counts)) == ref[:5]
    assert np.array_equal(blk2, ref[5]) and np.array_equal(buf2, ref[6])
    assert res.best_index == int(g["winner"])
    st, cost = ctx.fetch_status()   # (the context's bookkeeping of the last plan is in place)
    assert len(st) == q.n_candidates
    ctx.close()
    with pytest.raises(_capi.RpError):
        ctx.plan_packed_fast(blank, q.cost, q.T, q.traj_len, q.L, q.D, 0, False, 0, [0.0] * 3, [0.0] * 3, 0.0)


@pytest.mark.gpu
def test_planner_takes_the_fast_call_and_agrees_with_the_ctypes_path():
    """a closed loop through ReactivePlanner.plan() with and without the extension: the same trajectories"""
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import RpContext
    from commonroad_rp_amd.harness import run_closed_loop
    w = W.WORKLOADS["cfg2"]()
    calls = {"fast": 0}

    def loop(use_fast):
        ctx = RpContext(0)
        if use_fast:
            real = ctx.plan_packed_fast

            def counted(*a):
                calls["fast"] += 1
                return real(*a)
            ctx.plan_packed_fast = counted
        else:
            ctx.plan_packed_fast = None

        class Shared:
            def __new__(cls, dev):
                return ctx
        rp = W.make_planner(w, backend_factory=Shared, device=0)
        res = run_closed_loop(rp, max_steps=12, replanning_frequency=1)
        ctx.close()
        return res
    a, b = loop(True), loop(False)
    assert calls["fast"] >= 10
    assert len(a.states) == len(b.states)
    for sa, sb in zip(a.states, b.states):
        assert np.array_equal(np.asarray(sa.position), np.asarray(sb.position)) and sa.velocity == sb.velocity and sa.orientation == sb.orientation
