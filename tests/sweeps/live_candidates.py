"""The oracle against the REFERENCE run live on further random cases (the committed fixtures rand_00 .. rand_23 are draws 1000 .. 1023 of
the same generator): every candidate's coefficients, label, first-failure reason and cost, the winner, the counters and a sample of
state blocks -- the assertions of tests/test_oracle_golden.py on reference output that is made here and not kept.  Only where the
reference is mounted.   usage: python tests/sweeps/live_candidates.py [n=60] [first_seed=5000]"""
import os, sys, tempfile, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")]
if not os.path.isdir("/root/reference/commonroad_rp"):
    sys.exit("reference not mounted")
import make_golden as mg
import _golden
import test_oracle_golden as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
tmp = tempfile.mkdtemp(prefix="live_candidates_")
_golden.GOLDEN_DIR = tmp
bad = total = winners = 0
for k, case in enumerate(mg.random_cases(n=n, seed0=seed0)):
    if k % 4 == 0:   # the vehicle exactly on a lateral sample of the case's level
        lvl_n = {0: 3, 1: 5, 2: 9, 3: 17}[case["level"]]
        case["x0_lat"] = [float(np.linspace(-3, 3, lvl_n)[1 + k % (lvl_n - 2)]), case["x0_lat"][1], case["x0_lat"][2]]
    t0 = time.time()
    try:
        out = mg.run_case(dict(case))
    except ValueError as e:
        print(f"{case['name']:18s} the reference raises {type(e).__name__}: {str(e)[:70]} -- skipped", flush=True)
        continue
    np.savez(os.path.join(tmp, case["name"] + ".npz"), **out)
    try:
        T.test_oracle_matches_reference(case["name"])
        ok = True
    except AssertionError as e:
        ok = False
        print(str(e)[:400])
    os.remove(os.path.join(tmp, case["name"] + ".npz"))
    lab = out["label"]
    total += len(lab); winners += int(out["winner"]) >= 0; bad += not ok
    print(f"{case['name']:18s} level {case['level']} C={len(lab):5d} feasible={int((lab == 1).sum()):5d} kin={int((lab == 2).sum()):5d} coll={int((lab == 3).sum()):4d} "
          f"winner={int(out['winner']):5d} draw={int(bool(case.get('draw')))}: {'OK' if ok else 'MISMATCH'} (reference {time.time() - t0:.1f} s)", flush=True)
os.rmdir(tmp)
print(f"{total} candidates, {winners} cases with a winner: " + ("all live candidate checks passed" if not bad else f"{bad} MISMATCHES"))
sys.exit(1 if bad else 0)
