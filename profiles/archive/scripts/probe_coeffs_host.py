"""Host-side phases of rp_plan_coeffs on a corridor level (RP_AMD_TIMING=1: sums printed by rp_destroy).
usage (GPU box): RP_AMD_TIMING=1 python3 profiles/probe_coeffs_host.py [cfg3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "commonroad-reactive-planner_amd")]
from commonroad_rp_amd import workloads as W
from commonroad_rp_amd.corridor import ReachBox
from commonroad_rp_amd.sampling import CorridorSampling
base = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]()
rp = W.make_planner(base, device=0)
p = base.inputs.params
sp = CorridorSampling(rp.config)
sp.samples_t._dict_level_to_sample_set[1] = set(float(t) for t in base.inputs.T)
sp._dict_level_to_num_samples = {k: 21 for k in range(rp.config.sampling.num_sampling_levels)}
s0, v0 = p.x0_lon[0], max(p.x0_lon[1], 1.0)
cor = {}
for q in range(p.N + 2):
    k = p.time_step0 + q; t = q * p.dt
    cor[k] = [ReachBox(s0 - 1.0, s0 + 1.6 * v0 * t + 4.0, -2.8, -0.3, 0.5 * v0, 1.3 * v0 + 1.0),
              ReachBox(s0 + 0.3 * v0 * t, s0 + 1.8 * v0 * t + 6.0, -0.5, 1.4, 0.6 * v0, 1.4 * v0 + 1.0),
              ReachBox(s0 + 0.8 * v0 * t + 2.0, s0 + 2.0 * v0 * t + 8.0, 2.0, 3.0, 0.9 * v0, 1.5 * v0 + 1.0)]
sp.driving_corridor = cor
rp.set_sampling_space(sp)
ctx = rp._gpu_ctx()
sp.__dict__["_native_buffers"] = {"alloc": ctx.coeffs_arena}
sp.__dict__["_arena_owner"] = rp
x0_lon, x0_lat = rp.x_0_cl
mode = rp.config.sampling.longitudinal_mode
blon, blat, bT, btl, _, _ = sp.coeffs_at_level(1, x0_lon, x0_lat, mode, False)
params, cost = rp._gpu_params(x0_lon, x0_lat, 0), rp._gpu_cost()
groups = None if os.environ.get("PROBE_NO_GROUPS") else sp._last_groups[1]
print("groups", groups[0] if groups else None)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); out = ctx.plan_coeffs(params, cost, blon, blat, bT, btl, groups=groups); ts.append(time.perf_counter() - t0)
print("plan_coeffs p50 us", np.median(ts) * 1e6, "candidates", len(bT), "winner", out.best_index, "path", ctx.last_path())
rp.close()
