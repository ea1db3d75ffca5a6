"""Sweep of the randomised differential test (tests/_fuzz.py) over many seeds.
usage (GPU box): python tests/sweeps/fuzz_parity.py [first_seed] [n_cases] [seed,seed,...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests")]
from commonroad_rp_amd._capi import RpContext
from _fuzz import compare


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ctx = RpContext(0)
    bad, ties, total_c, winners, t_start = 0, 0, 0, 0, time.time()
    seeds = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(first, first + n)
    for seed in seeds:
        try:
            info, C, problems, ro = compare(ctx, seed)
        except Exception as e:   # a case the generator cannot build is not a parity failure
            print(f"seed {seed}: generator/setup error {type(e).__name__}: {e}")
            continue
        total_c += C
        winners += ro.best_index >= 0
        if problems:
            only_ties = all("tie within rounding" in p for p in problems)
            ties += only_ties
            bad += not only_ties
            print(f"seed {seed} {info} C={C}:")
            for p in problems:
                print("   ", p)
        if (seed - first) % 50 == 49:
            print(f"... {seed - first + 1} cases, {total_c} candidates, {winners} with a winner, {bad} with differences, {time.time() - t_start:.0f} s", flush=True)
    print(f"fuzz: {n} cases, {total_c} candidates, {winners} with a winner, {bad} cases with differences" +
          (f", {ties} with two candidates tied within rounding and the other one returned" if ties else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
