import sys, os, cProfile, pstats, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
rp = W.make_planner(W.cfg2())
for _ in range(20): rp.plan()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): rp.plan()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3800])
