"""In-kernel stamps of one cfg3 draw-mode plan (diagnostic library build with -DRP_STAMPS -DRP_TIMELINE).
usage (GPU box): RP_AMD_LIBRARY=.../librp_amd_stamps.so RP_AMD_PRINT_STAMPS=1 [RP_AMD_TIMELINE_LON=1] python profiles/probe_stamps_cfg3.py"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]()
w.inputs.params.flags = FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
ctx = RpContext(0); w.setup(ctx)
for _ in range(4):
    out = ctx.plan(w.inputs)
print("winner", out.best_index)
