"""One corridor level through rp_plan_coeffs, a few times (for a rocprofv3 --kernel-trace --memory-copy-trace timeline).
usage (GPU box): rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 profiles/probe_coeffs_timeline.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "commonroad-reactive-planner_amd")]
import bench
from commonroad_rp_amd import workloads as W
base = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]()
print(bench.corridor_sampling_cost(base, 0, reps=6))
