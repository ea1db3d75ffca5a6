"""Detail of one seed of the randomised differential test: which rows / steps of the winner's state block deviate.
usage (GPU box): python tests/sweeps/probe_fuzz_seed.py 28615"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")]
from commonroad_rp_amd._capi import RpContext, ARRAY_NAMES
from _fuzz import random_case
from oracle import oracle
np.set_printoptions(precision=6, linewidth=200)
for seed in [int(v) for v in sys.argv[1:]]:
    inp, co, obs, info = random_case(seed)
    ctx = RpContext(0)
    ctx.set_coordinate_system(co); ctx.set_obstacles(obs)
    tb = oracle.OracleTables.from_coordinate_system(co, obs)
    ref = oracle.plan(inp, tb, want_states=True)
    out = ctx.plan(inp)
    print(seed, info, "winner", out.best_index, ref.out.best_index, "x0_lon", list(inp.params.x0_lon), "x0_lat", list(inp.params.x0_lat), "low_vel", inp.params.low_vel_mode)
    dev = np.abs(out.best_states - ref.out.best_states)
    for r in range(14):
        if dev[r].max() > 1e-9:
            i = int(np.argmax(dev[r]))
            print("  row", ARRAY_NAMES[r], "max dev %.3g at step %d: got %r want %r" % (dev[r].max(), i, out.best_states[r, i], ref.out.best_states[r, i]))
    print("  v:", out.best_states[3]); print("  want v:", ref.out.best_states[3])
    print("  theta:", out.best_states[2]); print("  want theta:", ref.out.best_states[2])
    print("  s_dot:", out.best_states[10])
