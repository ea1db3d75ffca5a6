"""In-kernel stamps of one plan of a workload (diagnostic library build with -DRP_STAMPS): evaluation kernel, profile kernel, selection
epilogue.   usage (GPU box): RP_AMD_LIBRARY=.../librp_amd_st.so RP_AMD_PRINT_STAMPS=1 python profiles/probe_stamps.py <workload>[rb] [draw]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]()
if "draw" in sys.argv[2:]:
    w.inputs.params.flags = FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL
ctx = RpContext(0); w.setup(ctx)
for _ in range(4):
    out = ctx.plan(w.inputs)
print(name, "winner", out.best_index, "colliding", out.n_collision, "before the winner", out.n_collision_before_best, file=sys.stderr)
