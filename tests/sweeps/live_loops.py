"""Closed loops on random scenes, the REFERENCE planner run live beside this build's (oracle-backed context: the host logic of
plan() / reset() / set_desired_velocity and the restated arithmetic): state sequences, low-velocity / standstill flags and the counters
of every plan() call must agree.  Only where the reference is mounted (the build container); ~0.5 s per reference cycle.
usage: python tests/sweeps/live_loops.py [n_loops=12] [first_seed=9000]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")]
if not os.path.isdir("/root/reference/commonroad_rp"):
    sys.exit("reference not mounted")
import make_golden as mg   # (installs the third-party stand-ins, imports the reference)
from _golden import build_planner_from_plan_golden, record_plan_stats
from _oracle_ctx import OracleContext
from commonroad_rp_amd.harness import run_closed_loop

n_loops = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
bad = done = 0
for k, case in enumerate(mg.random_cases(n=3 * n_loops, seed0=seed0)):
    if case.get("lon_mode") == "stopping" or done >= n_loops:
        continue
    rng = np.random.default_rng(seed0 + 500 + k)
    case = dict(case, draw=False, cost_kind=0, steps=int(rng.integers(12, 26)), replanning_frequency=int(rng.choice([1, 1, 2, 3])))
    case.pop("constraints", None)
    v0 = case["x0_velocity"]
    v_a, v_b, switch = float(max(0.0, v0 + rng.normal(0, 2))), float(max(0.0, v0 + rng.normal(0, 4))), int(rng.integers(3, 10))
    if k % 4 == 0:   # on a lateral sample of level 1
        case["x0_lat"] = [float(np.linspace(-3, 3, 5)[int(rng.integers(1, 4))]), case["x0_lat"][1], case["x0_lat"][2]]
    case["schedule"] = lambda c, a=v_a, b=v_b, s=switch: a if c < s else b
    t0 = time.time()
    try:
        ref = mg.run_loop_case(dict(case))
    except ValueError as e:   # (the reference itself gives up on some made-up configurations: a time sample one step beyond the horizon)
        print(f"{case['name']:18s} the reference raises {type(e).__name__}: {str(e)[:80]} -- skipped", flush=True)
        continue
    ref["flags"], ref["plan_stats"] = ref["flags"].reshape(-1, 4), ref["plan_stats"].reshape(-1, 10)   # (no cycle completed: empty arrays)
    t1 = time.time()
    rp, _ = build_planner_from_plan_golden(dict(ref, continuous=0, via_scenario=0, planned=1), OracleContext)
    flags = []
    stats = record_plan_stats(rp)

    def on_step(c, planner, optimal):
        flags.append((int(planner._low_vel_mode), int(len(optimal[0].state_list) == planner.N)))
    res = run_closed_loop(rp, max_steps=case["steps"], replanning_frequency=case["replanning_frequency"], desired_velocity=case["schedule"], on_step=on_step)
    trace = np.array([[s.time_step, s.position[0], s.position[1], s.orientation, s.velocity, s.acceleration or 0.0, s.steering_angle or 0.0]
                      for s in res.states], dtype=float)
    ok = (int(res.completed) == int(ref["completed"]) and res.n_replans == int(ref["n_replans"]) and trace.shape == ref["trace"].shape
          and np.allclose(trace, ref["trace"], rtol=0, atol=1e-8) and np.array_equal(np.array(stats, dtype=np.int64).reshape(-1, 10) if len(stats) else np.zeros((0, 10)), ref["plan_stats"].reshape(-1, 10))
          and np.array_equal(np.array(flags, dtype=np.int64).reshape(-1, 2), ref["flags"][:, :2].reshape(-1, 2)))
    bad += not ok
    done += 1
    fl = ref["flags"]
    print(f"{case['name']:18s} steps {case['steps']:2d} every {case['replanning_frequency']} | reference: completed {int(ref['completed'])}, {int(ref['n_replans'])} replans, "
          f"low-velocity cycles {int(fl[:, 0].sum()) if len(fl) else 0}, standstill {int(fl[:, 1].sum()) if len(fl) else 0}, rejected by collisions {int(ref['plan_stats'][:, 3].sum()) if len(ref['plan_stats']) else 0} "
          f"| max |state difference| {np.abs(trace - ref['trace']).max() if trace.shape == ref['trace'].shape else float('nan'):.2e} : {'OK' if ok else 'MISMATCH'} "
          f"(reference {t1 - t0:.1f} s)", flush=True)
print("all live loops agree" if not bad else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
