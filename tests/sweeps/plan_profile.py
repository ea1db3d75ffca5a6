"""Python time of ReactivePlanner.plan() around the device call: the planner of a workload with a canned backend (the first
oracle result replayed), so that what is measured is the host glue alone.  usage: python tests/sweeps/plan_profile.py [cfg2] [--profile]"""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")]
import numpy as np  # noqa: E402
from commonroad_rp_amd import workloads as W  # noqa: E402
from _oracle_ctx import OracleContext  # noqa: E402


class Canned(OracleContext):
    cache = None

    def plan(self, inp, *a, **k):
        if Canned.cache is None:
            Canned.cache = super().plan(inp, *a, **k)
        self._N = inp.params.N
        return Canned.cache

    packed = None

    def plan_packed(self, params, cost, T, traj_len, L, D):   # (the planner's one-call-per-level path: _plan_fast)
        if Canned.packed is None:
            Canned.packed = super().plan_packed(params, cost, T, traj_len, L, D)
        self._N = params.N
        return Canned.packed

    def cost_range(self):
        return (0.0, 1.0, 1)


w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"]()
rp = W.make_planner(w, backend_factory=Canned)
rp.plan()
lat = []
for _ in range(400):
    rp.set_desired_velocity(current_speed=rp.x_0.velocity)
    t = time.perf_counter()
    rp.plan()
    lat.append(time.perf_counter() - t)
print("plan() host glue, p50 / p90 us: %.1f %.1f" % (np.percentile(lat, 50) * 1e6, np.percentile(lat, 90) * 1e6))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        rp.plan()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
