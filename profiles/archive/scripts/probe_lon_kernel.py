"""rp_lon_kernel alone: cfg3 / cfg4 in production mode with and without the obstacle tables (run under rocprofv3 --kernel-trace and read
the rp_lon_kernel rows).  usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python3 profiles/probe_lon_kernel.py cfg3 [noobs]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
from commonroad_rp_amd.collision import ObstacleTables
w = W.WORKLOADS[sys.argv[1]]()
ctx = RpContext(0)
w.setup(ctx)
if len(sys.argv) > 2:
    ctx.set_obstacles(ObstacleTables())
for i in range(60):
    ctx.plan(w.inputs, want_best_states=False)
ctx.close()
