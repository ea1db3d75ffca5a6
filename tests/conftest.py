import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- GPU session: the multi-process tests of tests/test_distributed_gpu.py -------------------------------------------------------
# Their ranks are CHILD PROCESSES that use the GPU.  A process that has initialised the GPU must not start another program (the GPU
# boxes refuse such an exec), so the children are started here -- before any test has touched the device -- and run to completion;
# the tests only read what they left behind.  CPU sessions (-m "not gpu") and boxes without a GPU skip this.
def _gpu_session(config) -> bool:
    expr = config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("RP_TEST_NO_DIST_GPU"):
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0          # (does not initialise the device, unlike is_available())
    except Exception:
        return False


def pytest_sessionstart(session):
    config = session.config
    config._rp_dist_gpu_dir = None
    if not _gpu_session(config):
        return
    import subprocess
    import tempfile
    out = tempfile.mkdtemp(prefix="rp_dist_gpu_")
    worker = os.path.join(REPO, "tests", "_dist_plan_worker.py")
    cases = ["plan_arc_hv_obs", "plan_all_collide", "plan_standstill", "plan_scurve_lv", "plan_arc_swept_hit", "cfg4_slice"]
    base = dict(os.environ, OMP_NUM_THREADS="1", RP_TEST_BACKEND="gpu", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    jobs = []

    def start(tag, env, log):
        f = open(os.path.join(out, log), "w")
        jobs.append((tag, subprocess.Popen([sys.executable, worker, out] + cases, env=dict(base, RP_TEST_TAG=tag, **env), stdout=f,
                                           stderr=subprocess.STDOUT), f))
    # reference: one process, no group | two ranks on GPU 0, gloo group + shared-memory mailbox | one rank, RCCL collectives
    start("ref_", dict(RANK="0", WORLD_SIZE="1"), "ref.log")
    for r in range(2):
        start("mb_", dict(RANK=str(r), WORLD_SIZE="2", MASTER_PORT="29761", RP_TEST_GROUP="gloo"), f"mb{r}.log")
    start("nccl_", dict(RANK="0", WORLD_SIZE="1", MASTER_PORT="29762", RP_TEST_GROUP="nccl", RP_TEST_TRANSPORT="collective",
                        RP_TEST_SINGLE="1"), "nccl.log")
    status = {}
    for tag, p, f in jobs:
        try:
            status.setdefault(tag, []).append(p.wait(timeout=900))
        except subprocess.TimeoutExpired:
            p.kill()
            status.setdefault(tag, []).append("timeout")
        f.close()
    # bench.py --gpus 2 with both ranks on GPU 0 (functional rehearsal of the N > 1 bench: the driver's 8-GPU run is not its first)
    with open(os.path.join(out, "bench2.log"), "w") as f, open(os.path.join(out, "bench2.json"), "w") as g:
        try:
            rc = subprocess.call([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                                  "--min-seconds", "0.02", "--sequence", "4", "--no-cpu-baseline"],
                                 env=dict(os.environ, RP_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1"),
                                 stdout=g, stderr=f, timeout=900)
        except subprocess.TimeoutExpired:
            rc = "timeout"
    status["bench2"] = [rc]
    import json
    with open(os.path.join(out, "status.json"), "w") as f:
        json.dump(status, f)
    config._rp_dist_gpu_dir = out


@pytest.fixture(scope="session")
def dist_gpu_dir(request):
    d = getattr(request.config, "_rp_dist_gpu_dir", None)
    if d is None:
        pytest.skip("no multi-process GPU results (not a GPU session, or RP_TEST_NO_DIST_GPU set)")
    return d
