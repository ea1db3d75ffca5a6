"""One seed of the fuzz comparison (tests/_fuzz.py) in detail: status / cost of given candidates on both sides, state rows.
usage (GPU box): python profiles/debug_fuzz_seed.py <seed> [candidate ...]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")]
from commonroad_rp_amd._capi import RpContext, PlanInputs, copy_params, FLAG_MATERIALIZE_ALL
from _fuzz import random_case
from oracle import oracle
np.set_printoptions(precision=17, linewidth=220)
seed = int(sys.argv[1]); cands = [int(v) for v in sys.argv[2:]]
inp, co, obs, info = random_case(seed)
print("seed", seed, info)
p = inp.params
print("x0_lon", list(p.x0_lon), "x0_lat", list(p.x0_lat), "low_vel", p.low_vel_mode, "lon_mode", p.lon_mode, "flags", p.flags, "mask", p.constraint_mask, "dt", p.dt, "N", p.N)
print("T", inp.T, "traj_len", inp.traj_len, "L", inp.L, "D", inp.D)
ctx = RpContext(0)
ctx.set_coordinate_system(co); ctx.set_obstacles(obs)
tb = oracle.OracleTables.from_coordinate_system(co, obs)
ref = oracle.plan(inp, tb, want_states=True)
p2 = copy_params(p); p2.flags |= FLAG_MATERIALIZE_ALL
out = ctx.plan(PlanInputs(p2, inp.cost, inp.T, inp.traj_len, inp.L, inp.D))
st, cs = ctx.fetch_status()
states = ctx.fetch_states()
print("GPU winner", out.best_index, out.best_cost.hex() if out.best_index >= 0 else None, "oracle winner", ref.out.best_index, float(ref.out.best_cost).hex() if ref.out.best_index >= 0 else None)
print("counters GPU", out.n_feasible, out.n_collision, out.n_collision_before_best, "oracle", ref.out.n_feasible, ref.out.n_collision, ref.out.n_collision_before_best)
diff = np.flatnonzero((st & 0xff) != (ref.status & 0xff))
print("status words differing (low byte):", diff[:20], [(hex(int(st[i])), hex(int(ref.status[i]))) for i in diff[:10]])
for c in cands or list(diff[:4]):
    print(f"--- candidate {c}: status GPU {int(st[c]):#x} oracle {int(ref.status[c]):#x}; cost GPU {cs[c]!r} ({float(cs[c]).hex()}) oracle {ref.cost[c]!r} ({float(ref.cost[c]).hex()})")
    names = ["x", "y", "theta", "v", "a", "kappa", "kappa_dot", "s", "d", "s_dot", "s_ddot", "d_dot", "d_ddot", "theta_cl"]
    for r in range(14):
        g, o = states[c, r], ref.states[c, r]
        print(f"   {names[r]:9s} GPU {g}  oracle {o}  max diff {np.nanmax(np.abs(g - o)):.3g}")
