"""One production-mode plan of a workload with the cost-ordered stage forced and its rounds traced (RP_AMD_LAZY_TRACE).
usage (GPU box): python profiles/probe_lazy_trace.py <workload>[rb]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "commonroad-reactive-planner_amd")]
os.environ["RP_AMD_LAZY"] = "1"
os.environ["RP_AMD_LAZY_TRACE"] = "1"
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import RpContext
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4rb"
w = W.WORKLOADS[name[:-2]](road_boundary=True) if name.endswith("rb") else W.WORKLOADS[name]()
ctx = RpContext(0); w.setup(ctx)
for k in range(3):
    t = time.perf_counter()
    out = ctx.plan(w.inputs)
    print(f"{name}: plan {k}: {(time.perf_counter() - t) * 1e6:.0f} us, path {ctx.last_path()}, winner {out.best_index}, before the winner {out.n_collision_before_best}", file=sys.stderr)
