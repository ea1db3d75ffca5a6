"""Randomised differential test: the HIP path (through the C ABI) against the CPU oracle on seeded random scenarios
(tests/_fuzz.py).  Sweeps over 10 000 seeds (default launch path) and 3 000 seeds (each other path) are recorded in profiles/r01_fuzz_parity.txt."""
import numpy as np
import pytest

from _fuzz import compare, random_case


def test_generator_is_seeded_and_varied():
    a, _, _, ia = random_case(7)
    b, _, _, ib = random_case(7)
    assert ia == ib and np.array_equal(a.T, b.T) and np.array_equal(a.L, b.L) and np.array_equal(a.D, b.D)
    infos = [random_case(s)[3] for s in range(40)]
    assert len({i["mode"] for i in infos}) >= 4 and any(i["draw"] for i in infos) and any(i["n_dyn"] > 63 for i in infos)
    assert any(i["N"] > 64 for i in infos) and any(i["factor"] > 1 for i in infos) and any(i["mask"] != 31 for i in infos)


@pytest.mark.gpu
@pytest.mark.parametrize("path,first", [("single_launch", 0), ("single_launch", 100), ("two_kernel", 200), ("g32", 300), ("g64", 400), ("lazy", 500),
                                        ("lazy", 600)])
def test_random_scenarios_match_oracle(path, first):
    from _paths import launch_path_env
    from commonroad_rp_amd._capi import RpContext
    with launch_path_env(path):
        ctx = RpContext(0)
        winners = candidates = 0
        for seed in range(first, first + 100):
            info, C, problems, ro = compare(ctx, seed)
            assert not problems, (path, seed, info, problems)
            winners += ro.best_index >= 0
            candidates += C
        assert winners >= 15 and candidates > 10000    # the cases are not degenerate
        ctx.close()
