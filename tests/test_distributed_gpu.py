"""Sharding inside ``ReactivePlanner.plan()`` (row N1) on the REAL device context: ranks as separate processes that share GPU 0
(the arrangement of RP_BENCH_REHEARSE), started by tests/conftest.py at session start -- before this process touches the GPU -- through
tests/_dist_plan_worker.py:

  ref_   one process, no group                                   -- what every sharded run must reproduce
  mb_    two ranks, gloo group, shared-memory mailbox (the intra-node default): HIP context, pinned result block and mailbox meet
         inside plan(); min_candidates = 1 on the plan_* goldens, the default threshold on a 32 768-candidate slice of cfg4
  nccl_  one rank, RCCL group, `shard_single_rank`: the device-side exchange (rp_result_device -> all_gather_into_tensor ->
         rp_combine_kernel) inside plan()

and ``bench.py --gpus 2`` with both ranks on GPU 0 (functional: the driver's 8-GPU run is not the first N > 1 run of the bench)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = ["plan_arc_hv_obs", "plan_all_collide", "plan_standstill", "plan_scurve_lv", "plan_arc_swept_hit", "cfg4_slice"]


def _load(d, name):
    path = os.path.join(d, name)
    assert os.path.exists(path), (name, open(os.path.join(d, "status.json")).read(), _logs(d))
    return json.load(open(path))


def _logs(d):
    out = []
    for f in sorted(os.listdir(d)):
        if f.endswith(".log"):
            out.append(f"--- {f}\n" + open(os.path.join(d, f)).read()[-1500:])
    return "\n".join(out)


def _same(a, b, name):
    a, b = dict(a), dict(b)
    for k in ("planned", "n_kin", "n_coll", "reasons"):
        assert a[k] == b[k], (name, k, a[k], b[k])
    if a["planned"]:
        # (a shard is a smaller batch and may take another launch path than the whole grid: same winner, states to the contract)
        np.testing.assert_allclose(np.array(a["cart"]), np.array(b["cart"]), rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array(a["lon"]), np.array(b["lon"]), rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array(a["lat"]), np.array(b["lat"]), rtol=0, atol=1e-6)


def test_children_ran(dist_gpu_dir):
    st = json.load(open(os.path.join(dist_gpu_dir, "status.json")))
    assert st["ref_"] == [0] and st["mb_"] == [0, 0] and st["nccl_"] == [0], (st, _logs(dist_gpu_dir))


@pytest.mark.parametrize("rank", [0, 1])
def test_two_ranks_on_one_gpu_mailbox(dist_gpu_dir, rank):
    ref = _load(dist_gpu_dir, "ref_rank0_of1.json")
    got = _load(dist_gpu_dir, f"mb_rank{rank}_of2.json")
    assert got.pop("_small_level_sharded_calls") == 0          # a level below the threshold runs whole on every rank
    for name in CASES:
        assert ref[name]["sharded_calls"] == 0
        assert got[name]["sharded_calls"] >= 1, name             # the level(s) really went through the exchange
        _same(got[name], ref[name], name)
    assert ref["cfg4_slice"]["planned"] and ref["plan_arc_hv_obs"]["planned"] and not ref["plan_all_collide"]["planned"]


def test_single_rank_rccl_collective(dist_gpu_dir):
    ref = _load(dist_gpu_dir, "ref_rank0_of1.json")
    got = _load(dist_gpu_dir, "nccl_rank0_of1.json")
    assert got.pop("_small_level_sharded_calls") == 0
    for name in CASES:
        assert got[name]["sharded_calls"] >= 1, name
        _same(got[name], ref[name], name)


def test_bench_two_ranks_rehearsal(dist_gpu_dir):
    st = json.load(open(os.path.join(dist_gpu_dir, "status.json")))
    log = open(os.path.join(dist_gpu_dir, "bench2.log")).read()[-3000:]
    assert st["bench2"] == [0], log
    line = open(os.path.join(dist_gpu_dir, "bench2.json")).read().strip().splitlines()[-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["config"]["parallelism"] != "one GPU"
    # the compact line (what a SCALE record of the driver holds): how the winner messages travelled, the other transport beside it,
    # the ranks RCCL itself saw (0: a rehearsal runs on a gloo group), rank 0 alone on one shard's grid in the same run
    assert len(line) < 4096, len(line)
    for key in ("exchange", "exchange_ms_per_step", "other_transport", "ranks_seen_by_rccl", "one_gpu_same_grid", "strong_scaling",
                "roofline", "wait_mode", "detail"):
        assert key in r, (key, list(r))
    assert r["exchange"] == r["config"]["exchange"] and r["rehearsal"] is True and r["ranks_seen_by_rccl"] == 0
    assert r["one_gpu_same_grid"]["value"] > 0 and r["other_transport"].get("exchange") != r["exchange"]
    full = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), r["detail"])))
    for key in ("weak", "strong"):
        assert key in full, list(full)
