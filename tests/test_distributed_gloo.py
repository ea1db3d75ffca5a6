"""N > 1 path on CPU: world_size-2 (and 3) gloo groups, one rank per shard of the candidate range;
the combined result must equal the unsharded oracle run on every rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from _golden import Golden
from commonroad_rp_amd.distributed import shard_range, pack_result, combine_results
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
# (with and without collisions before the winner, no winner at all, ties, every rank / one rank holding the collisions)
CASES = ["arc_hv_l2_obs", "straight_hv_l2_ties", "arc_all_collide", "arc_n70_factor2", "rand_02_hv", "rand_06_hv", "rand_09_hv",
         "rand_16_hv"]


def test_shard_range_covers_everything():
    for C in (0, 1, 7, 540, 7440, 1009899):
        for R in (1, 2, 3, 8):
            spans = [shard_range(C, r, R) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == C
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))


@pytest.mark.parametrize("world,transport", [(2, "mailbox"), (3, "mailbox"), (2, "collective"), (3, "auto")])
def test_sharded_exchange_matches_unsharded(tmp_path, world, transport):
    """Both transports of the winner exchange: the shared-memory mailbox of the ranks of one node (what
    bench.py --gpus N uses) and the torch.distributed collectives (gloo here, RCCL on the GPU box)."""
    port = 29650 + world + {"mailbox": 0, "collective": 10, "auto": 20}[transport]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1",
               RP_AMD_EXCHANGE=transport)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(tmp_path)] + CASES,
                              env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    for name in CASES:
        g = Golden(name)
        full = oracle.plan(g.inputs, g.oracle_tables()).out
        for r in range(world):
            res = json.load(open(tmp_path / f"rank{r}.json"))[name]
            assert res["best_index"] == full.best_index == int(g["winner"])
            assert res["n_before"] == full.n_collision_before_best == int(g["n_infeasible_collision"])
            assert res["n_feasible"] == full.n_feasible and res["n_collision"] == full.n_collision
            assert res["n_candidates"] == g.inputs.n_candidates
            assert res["reasons"][1:7] == full.reason_counts[1:7].tolist()
            if full.best_index >= 0:
                assert res["best_cost"] == full.best_cost
                assert res["states_sum"] == float(np.sum(full.best_states))
                assert res["lon"] == full.best_lon_coeffs.tolist()


@pytest.mark.parametrize("broken_rank", [0, 1])
def test_mailbox_setup_fails_on_every_rank_or_none(tmp_path, broken_rank):
    """One rank without shared memory: "auto" must end on the collectives on ALL ranks (a rank falling back on its own
    would leave the others waiting in the mailbox set-up), and the results stay right."""
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29690 + broken_rank), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1", RP_AMD_EXCHANGE="auto", TEST_BREAK_SHM_ON_RANK=str(broken_rank))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(tmp_path), CASES[0]],
                              env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    g = Golden(CASES[0])
    for r in range(world):
        res = json.load(open(tmp_path / f"rank{r}.json"))
        assert res["_transport"] == ["CollectiveExchange"]
        assert res[CASES[0]]["best_index"] == int(g["winner"])


def test_pack_and_combine_roundtrip():
    g = Golden("arc_hv_l2_obs")
    C = g.inputs.n_candidates
    n = g.inputs.params.N + 1
    full = oracle.plan(g.inputs, g.oracle_tables()).out
    msgs = np.stack([pack_result(oracle.plan(g.inputs, g.oracle_tables(), *shard_range(C, r, 4)).out, n) for r in range(4)])
    glob, owner = combine_results(msgs, n)
    assert glob.best_index == full.best_index and glob.best_cost == full.best_cost
    assert shard_range(C, owner, 4)[0] <= full.best_index < shard_range(C, owner, 4)[1]
    np.testing.assert_array_equal(glob.best_states, full.best_states)
    np.testing.assert_array_equal(glob.reason_counts, full.reason_counts)
    assert (glob.n_feasible, glob.n_collision, glob.n_candidates) == (full.n_feasible, full.n_collision, C)


@pytest.mark.parametrize("name", CASES + ["rand_01_hv", "rand_20_hv", "arc_hv_l1_draw"])
def test_local_share_of_the_collision_count_without_the_second_pass(name):
    """local_collisions_before takes a rank's share from its own result where that already determines it (owner, no global
    winner, nothing colliding before the local winner); it must equal the direct count in every case."""
    from commonroad_rp_amd.distributed import local_collisions_before
    g = Golden(name)
    C, n = g.inputs.n_candidates, g.inputs.params.N + 1
    full = oracle.plan(g.inputs, g.oracle_tables()).out

    class Counting:
        calls = 0

        def __init__(self, run, base):
            self.run, self.base = run, base

        def count_collisions_before(self, cost, index):
            Counting.calls += 1
            return oracle.count_collisions_before(self.run.status, self.run.cost, self.base, cost, index)

    for world in (2, 3, 5, 8):
        runs = [oracle.plan(g.inputs, g.oracle_tables(), *shard_range(C, r, world)) for r in range(world)]
        glob, owner = combine_results(np.stack([pack_result(r.out, n) for r in runs]), n)
        total = 0
        for r, run in enumerate(runs):
            base = shard_range(C, r, world)[0]
            direct = oracle.count_collisions_before(run.status, run.cost, base, glob.best_cost if glob.best_index >= 0 else 0.0,
                                                    glob.best_index)
            got = local_collisions_before(Counting(run, base), run.out, glob, r == owner)
            assert got == direct, (world, r)
            total += got
        assert total == full.n_collision_before_best == int(g["n_infeasible_collision"])


def test_mailbox_timeout_names_the_stalled_rank():
    """A peer that never posts: the wait ends after the configured budget with RP_ESTATE and the library names the rank
    it was waiting for (host-only entry points of librp_amd.so: no GPU involved)."""
    import ctypes as C
    import time
    from commonroad_rp_amd import _capi
    lib = _capi.load_library()
    n, world = 21, 3
    size = int(lib.rp_mailbox_bytes(world, n))
    region = (C.c_char * size)()
    assert lib.rp_mailbox_set_timeout(-1.0) != 0
    assert lib.rp_mailbox_set_timeout(0.15) == 0
    try:
        loc, glob = _capi.RpResult(), _capi.RpResult()
        loc.best_index = -1
        owner = C.c_int32(-5)
        # ranks 0 and 1 post, rank 2 never does: whoever waits is told "rank 2"
        st = np.zeros((14, n))
        t0 = time.perf_counter()
        rc0 = lib.rp_mailbox_exchange(C.addressof(region), world, 1, 1, n, C.byref(loc), _capi.dptr(st), C.byref(glob), _capi.dptr(st), C.byref(owner))
        assert rc0 != 0 and lib.rp_mailbox_stalled_rank() == 0          # nobody else there yet: rank 0 is the first it waits for
        rc1 = lib.rp_mailbox_exchange(C.addressof(region), world, 0, 1, n, C.byref(loc), _capi.dptr(st), C.byref(glob), _capi.dptr(st), C.byref(owner))
        assert rc1 != 0 and lib.rp_mailbox_stalled_rank() == 2
        assert 0.25 < time.perf_counter() - t0 < 5.0
        total = C.c_int64(0)
        assert lib.rp_mailbox_sum(C.addressof(region), 1, 0, 1, n, 7, C.byref(total)) == 0 and total.value == 7   # world of one: no wait
        assert lib.rp_mailbox_stalled_rank() == -1
    finally:
        lib.rp_mailbox_set_timeout(30.0)


def test_exchange_posts_only_the_result_it_was_given():
    """MailboxExchange may post the context's raw C result instead of re-packing `out` only when `out` IS that result (same
    call serial) -- a later select() / plan_coeffs() on the context, or an `out` from elsewhere, must not be confused with it
    even when winner index and candidate count happen to agree."""
    from commonroad_rp_amd._capi import PlanOutput, RpContext, RpResult
    ctx = object.__new__(RpContext)
    ctx._res, ctx._last_best, ctx._serial = RpResult(), None, 0
    r = ctx._res
    r.best_index, r.n_candidates, r.best_cost = 4, 10, 1.5
    a = ctx._output(r, np.zeros((14, 3)))
    assert a.serial == 1 == ctx._serial
    r.best_cost = 9.0                       # a select() with external costs rewrites the same C struct ...
    b = ctx._output(r, np.ones((14, 3)))
    assert b.serial == 2 and a.serial != ctx._serial and b.best_index == a.best_index and b.n_candidates == a.n_candidates
    foreign = PlanOutput.from_c(r, None)
    assert foreign.serial == 0              # ... and a result that did not come from this context never matches


PLAN_CASES = ["plan_arc_hv_obs", "plan_all_collide", "plan_standstill", "plan_scurve_lv", "plan_arc_swept_hit", "cfg4_slice"]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_plan_matches_single_rank(tmp_path, world):
    """Row N1: sharding lives inside ``ReactivePlanner.plan()`` -- the reference's only parallel backend also sits inside
    ``_get_optimal_trajectory`` (reactive_planner.py:1084-1111).  Every rank of a gloo group runs ``plan()`` with the planner's
    process group set; trajectory, counters and reasons must equal a single-rank run on every rank: several sampling levels
    (plan_all_collide visits them all), the standstill branch with its cost range over all shards (plan_standstill), low-velocity
    mode, the continuous collision check of the winner (plan_arc_swept_hit), and a 32 768-candidate slice of cfg4 that
    crosses the planner's own saturation threshold while a small level stays whole on every rank."""
    worker = os.path.join(HERE, "_dist_plan_worker.py")
    base_env = dict(os.environ, OMP_NUM_THREADS="1")
    single = subprocess.Popen([sys.executable, worker, str(tmp_path)] + PLAN_CASES, env=dict(base_env, RANK="0", WORLD_SIZE="1"))
    env = dict(base_env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29730 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path)] + PLAN_CASES, env=dict(env, RANK=str(r))) for r in range(world)]
    assert single.wait(timeout=600) == 0
    for p in procs:
        assert p.wait(timeout=600) == 0
    ref = json.load(open(tmp_path / "rank0_of1.json"))
    for r in range(world):
        got = json.load(open(tmp_path / f"rank{r}_of{world}.json"))
        assert got.pop("_small_level_sharded_calls") == 0
        for name in PLAN_CASES:
            a, b = dict(got[name]), dict(ref[name])
            assert b.pop("sharded_calls") == 0
            assert a.pop("sharded_calls") >= 1, name          # the level(s) really went through the exchange
            assert a == b, (name, r)
