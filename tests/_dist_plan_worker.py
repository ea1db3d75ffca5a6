"""Worker of tests/test_distributed_gloo.py::test_sharded_plan_* and tests/test_distributed_gpu.py: one rank of a process group
running ``ReactivePlanner.plan()`` with the planner's process group set.  Environment:
  RP_TEST_BACKEND    oracle (default: the oracle-backed context stands in for the device, no GPU in the CPU suite) | gpu (RpContext
                     on device 0 -- several ranks share ONE GPU, the arrangement of RP_BENCH_REHEARSE)
  RP_TEST_GROUP      gloo (default) | nccl (RCCL; a world of one rank on the one-GPU box)
  RP_TEST_TRANSPORT  auto (default: the shared-memory mailbox on one host) | mailbox | collective
  RP_TEST_SINGLE     1: a group of ONE rank still shards (planner.shard_single_rank): the whole exchange path with one rank"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def summary(rp, res, n_calls):
    out = dict(planned=res is not None, n_kin=int(rp.infeasible_count_kinematics), n_coll=int(rp.infeasible_count_collision),
               reasons={k: int(v) for k, v in rp.infeasible_reason_dict.items()}, sharded_calls=n_calls)
    if res is not None:
        cart, cvln, lon, lat = res
        out["cart"] = [[float(st.position[0]), float(st.position[1]), float(st.orientation), float(st.velocity)] for st in cart.state_list]
        out["lon"], out["lat"] = lon, lat
    return out


def cfg4_slice_planner(backend):
    """a slice of BASELINE.json configs[3]: 16 of cfg4's 63 time samples x 32 velocity samples x its 64 lateral samples = 32 768
    candidates -- past the planner's default sharding threshold"""
    import dataclasses
    from commonroad_rp_amd import workloads as W
    from commonroad_rp_amd._capi import PlanInputs
    w = W.cfg4()
    i = w.inputs
    T = i.T[::4][:16]
    w2 = dataclasses.replace(w, inputs=PlanInputs(i.params, i.cost, T, W.traj_len_of(T, 0.1), np.linspace(i.L.min(), i.L.max(), 32), i.D))
    return W.make_planner(w2, backend_factory=backend)


def main():
    import torch
    import torch.distributed as dist
    from _golden import build_planner_from_plan_golden
    from commonroad_rp_amd import distributed as D

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend, group = os.environ.get("RP_TEST_BACKEND", "oracle"), os.environ.get("RP_TEST_GROUP", "gloo")
    transport, single = os.environ.get("RP_TEST_TRANSPORT", "auto"), os.environ.get("RP_TEST_SINGLE") == "1"
    if backend == "gpu":
        from commonroad_rp_amd._capi import RpContext as OracleContext   # (the real device context under the name used below)
    else:
        from _oracle_ctx import OracleContext
    dev = torch.device("cuda", 0) if group == "nccl" else torch.device("cpu")
    grouped = world > 1 or single
    if grouped:
        if group == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    calls = [0]
    real = D.exchange_winner

    def counting(*a, **k):
        calls[0] += 1
        return real(*a, **k)
    D.exchange_winner = counting
    results = {}
    for name in sys.argv[2:]:
        calls[0] = 0
        if name == "cfg4_slice":
            rp = cfg4_slice_planner(OracleContext)
            if grouped:
                rp.shard_single_rank = single
                rp.set_process_group(dist, dev, transport=transport)                 # default threshold: 32 768 >= 16 384 shards
        else:
            rp, _ = build_planner_from_plan_golden(name, OracleContext)
            if grouped:
                rp.shard_single_rank = single
                rp.set_process_group(dist, dev, transport=transport, min_candidates=1)   # (the fixtures are small: force it)
        res = rp.plan()
        results[name] = summary(rp, res, calls[0])
        if hasattr(rp, "close"):
            rp.close()
        if name == "cfg4_slice" and grouped:   # a level below the threshold runs whole on every rank
            rp2, _ = build_planner_from_plan_golden("plan_arc_hv_obs", OracleContext)
            rp2.shard_single_rank = single
            rp2.set_process_group(dist, dev, transport=transport)
            calls[0] = 0
            rp2.plan()
            results["_small_level_sharded_calls"] = calls[0]
            rp2.close()
    if grouped:
        D.close_exchanges()
        dist.barrier()
    tag = os.environ.get("RP_TEST_TAG", "")
    with open(os.path.join(sys.argv[1], f"{tag}rank{rank}_of{world}.json"), "w") as f:
        json.dump(results, f)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
