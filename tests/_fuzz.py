"""Randomised differential cases for the HIP path against the CPU oracle (used by tests/test_fuzz_parity.py and by the
sweep script tests/sweeps/fuzz_parity.py).  Random smooth reference paths, initial states (moving, slow, standstill), horizons,
grids in arbitrary order, obstacle sets of every kind, modes (velocity keeping / stopping, low velocity, draw), constraint
subsets, planning.factor, sharded ranges.  Per case: labels / reasons / failing step / winner / counters exact, costs 1e-8
relative (sums of squares of whatever a relaxed constraint mask lets through), states 1e-6."""
import math

import numpy as np

from _lazy import lazy_relaxed

from commonroad_rp_amd import workloads as W
from commonroad_rp_amd._capi import (PlanInputs, make_params, make_cost, copy_params, FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL,
                                     COST_DEFAULT, COST_FAILSAFE, LON_STOPPING, LON_VELOCITY_KEEPING)
from commonroad_rp_amd.collision import ObstacleTables
from commonroad_rp_amd.coordinate_system import CoordinateSystem
from oracle import oracle


def random_path(rng):
    n = int(rng.integers(60, 420))
    step = 1.0
    kind = rng.integers(0, 4)
    s = np.arange(n) * step
    if kind == 0:
        kappa = np.zeros(n)
    elif kind == 1:
        kappa = np.full(n, rng.choice([-1, 1]) / rng.uniform(25.0, 400.0))
    else:
        kappa = sum(rng.uniform(0.0, 0.02) * np.sin(2 * np.pi * s / rng.uniform(30.0, 200.0) + rng.uniform(0, 6.28)) for _ in range(3))
    th = rng.uniform(-math.pi, math.pi) + np.cumsum(kappa * step)
    x = rng.uniform(-500, 500) + np.cumsum(np.cos(th) * step)
    y = rng.uniform(-500, 500) + np.cumsum(np.sin(th) * step)
    return np.stack((x, y), 1)


def random_case(seed):
    rng = np.random.default_rng(seed)
    co = CoordinateSystem(random_path(rng))
    s_max = co.ref_pos[-1]
    dt = float(rng.choice([0.1, 0.1, 0.2]))
    N = int(rng.choice([5, 12, 15, 16, 20, 30, 31, 32, 40, 60, 63, 64, 70, 100, 110]))
    horizon = N * dt
    mode = rng.choice(["hv", "hv", "lv", "still", "stop"])
    v0 = {"hv": rng.uniform(4.5, 25.0), "lv": rng.uniform(0.3, 3.5), "still": rng.choice([0.0, 0.02]), "stop": rng.uniform(3.0, 12.0)}[mode]
    low_vel = v0 < 4.0
    s0 = rng.uniform(5.0, max(6.0, min(0.5 * s_max, s_max - v0 * horizon * 1.3 - 10.0)))
    x0_lon = [float(s0), float(v0), float(rng.normal(0, 0.8))]
    x0_lat = [float(rng.normal(0, 0.8)), float(rng.normal(0, 0.15 if not low_vel else 0.03)), float(rng.normal(0, 0.05))]
    theta0 = float(np.interp(s0, co.ref_pos, np.unwrap(co.ref_theta)) + (math.atan(x0_lat[1]) if low_vel else math.asin(np.clip(x0_lat[1] / max(v0, 0.5), -0.5, 0.5))))
    k_min = max(2, int(rng.integers(2, max(3, N // 2))))
    ks = np.unique(rng.choice(np.arange(k_min, N + 1), size=min(int(rng.integers(1, 9)), N + 1 - k_min), replace=False))
    rng.shuffle(ks)                                   # (reference order = set iteration order: arbitrary)
    T = np.array([dt * k for k in ks])
    if rng.random() < 0.3:
        T = np.array([float(np.arange(0, round(t + dt, 2), dt)[-1]) for t in T])   # not clean multiples of dt (sampling.py:113-118)
    tl = W.traj_len_of(T, dt)
    nL, nD = int(rng.integers(1, 12)), int(rng.integers(1, 20))
    stopping = mode == "stop"
    if stopping:
        L = s0 + rng.uniform(-4.0, 0.6 * v0 * horizon) + np.linspace(-3, 8, nL) * rng.uniform(0.1, 1.0)
    else:
        lo, hi = W.velocity_range(v0, horizon)
        L = np.linspace(lo, hi, nL) if rng.random() < 0.7 else rng.uniform(max(0, v0 - 6), v0 + 6, nL)
    D = np.linspace(-3, 3, nD) if rng.random() < 0.6 else rng.uniform(-4, 4, nD)
    if rng.random() < 0.7:
        D = W._with_d0(D, x0_lat[0])
    rng.shuffle(L); rng.shuffle(D)
    draw = rng.random() < 0.25
    mask = 31 if rng.random() < 0.6 else int(rng.integers(0, 32))
    factor = int(rng.choice([1, 1, 1, 2, 3]))
    t0 = int(rng.integers(0, 30))
    veh = dict(W.VEHICLE2)
    params = make_params(dt=dt, N=N, factor=factor, time_step0=t0, low_vel_mode=low_vel, lon_mode=LON_STOPPING if stopping else LON_VELOCITY_KEEPING,
                         constraint_mask=mask, flags=FLAG_DRAW_ALL if draw else 0, x0_lon=x0_lon, x0_lat=x0_lat, x0_orientation=theta0, **veh)
    if rng.random() < 0.15:
        cost = make_cost(COST_FAILSAFE)
    else:
        cost = make_cost(COST_DEFAULT, w_a=float(rng.choice([1, 5])), desired_speed=None if stopping else float(v0 + rng.normal(0, 2)),
                         desired_d=float(rng.choice([0.0, 0.5])), desired_s=float(s0 + 20) if stopping else None)
    # obstacles near the corridor
    def at(s, d):
        return co.convert_to_cartesian_coords(float(np.clip(s, 1.0, s_max - 1.0)), float(d))
    n_dyn = int(rng.choice([0, 0, 1, 3, 8, 70]))
    n_steps = t0 + N * factor + int(rng.integers(-10, 12))
    n_steps = max(1, n_steps)
    dyn = np.full((n_dyn, n_steps, 5), np.nan)
    for j in range(n_dyn):
        sj, dj, vj = s0 + rng.uniform(-5, v0 * horizon + 25), rng.uniform(-4, 4), rng.uniform(0, 15)
        a, b = sorted(rng.integers(0, n_steps + 1, 2))
        for q in range(a, b):
            p = at(sj + vj * dt * q / max(factor, 1), dj + 0.02 * q)
            if p is not None:
                dyn[j, q] = (p[0], p[1], rng.uniform(-3.2, 3.2), rng.uniform(0.3, 3.0), rng.uniform(0.3, 1.2))
    sobb, tri, circ = [], [], []
    for _ in range(int(rng.choice([0, 0, 2, 6, 40, 130]))):
        p = at(s0 + rng.uniform(-5, v0 * horizon + 30), rng.normal(0, 3.5) + rng.choice([-6, 0, 6]))
        if p is None:
            continue
        k = rng.integers(0, 3)
        if k == 0:
            sobb.append([p[0], p[1], rng.uniform(-3.2, 3.2), rng.uniform(0.2, 8.0), rng.uniform(0.05, 1.2)])
        elif k == 1:
            tri.append([p[0], p[1], p[0] + rng.uniform(-3, 3), p[1] + rng.uniform(-3, 3), p[0] + rng.uniform(-3, 3), p[1] + rng.uniform(-3, 3)])
        else:
            circ.append([p[0], p[1], rng.uniform(0.1, 2.0)])
    obs = ObstacleTables(static_obb=sobb, static_tri=tri, static_circ=circ, dyn_obb=dyn, dyn_t0=int(rng.integers(0, 4)))
    return PlanInputs(params, cost, T, tl, L, D), co, obs, dict(mode=mode, N=N, dt=dt, draw=draw, mask=mask, factor=factor, nT=len(T), nL=nL,
                                                                 nD=len(D), n_dyn=n_dyn, n_static=len(sobb) + len(tri) + len(circ))


def compare(ctx, seed):
    inp, co, obs, info = random_case(seed)
    ctx.set_coordinate_system(co)
    ctx.set_obstacles(obs)
    tb = oracle.OracleTables.from_coordinate_system(co, obs)
    ref = oracle.plan(inp, tb, want_states=True)
    C = inp.n_candidates
    problems = []
    for variant in ("fused", "materialize", "shard"):
        if variant == "shard":
            lo, hi = C // 3, max(C // 3, (2 * C) // 3)
            out = ctx.plan(inp, lo, hi)
            refv = oracle.plan(inp, tb, lo, hi, want_states=False)
            st, cs = ctx.fetch_status()
            rs, rc, ro = refv.status, refv.cost, refv.out
        else:
            p2 = copy_params(inp.params)
            if variant == "materialize":
                p2.flags |= FLAG_MATERIALIZE_ALL
            out = ctx.plan(PlanInputs(p2, inp.cost, inp.T, inp.traj_len, inp.L, inp.D))
            st, cs = ctx.fetch_status()
            rs, rc, ro = ref.status, ref.cost, ref.out
        # label and first-failure reason of every candidate, failing step of the kinematically infeasible ones (the step a
        # collision is found at is not part of the reference's result)
        key = lambda w: np.where((w & 3) == 2, w, w & 0xff)
        # (plans that ran the cost-ordered collision stage label the colliding candidates they looked at: _lazy.lazy_relaxed)
        path = getattr(ctx, "last_path", lambda: 0)()
        lazy_run = path in (1, 3)   # (list rounds | bounded sweep)
        unlabelled = 0
        if lazy_run:
            try:
                st, unlabelled = lazy_relaxed(st, cs, refv if variant == "shard" else ref, ctx, out, lo if variant == "shard" else 0)
            except AssertionError:
                problems.append(f"{variant}: a colliding candidate the cost-ordered stage left unlabelled sorts before the winner")
        if not np.array_equal(key(st), key(rs)):
            bad = np.flatnonzero(key(st) != key(rs))
            problems.append(f"{variant}: {len(bad)} status words differ, first {bad[0]}: got {st[bad[0]]:#x} want {rs[bad[0]]:#x}")
        both = ~np.isnan(rc)
        if not np.array_equal(np.isnan(cs), np.isnan(rc)):
            problems.append(f"{variant}: cost NaN pattern differs")
        elif both.any() and np.max(np.abs(cs[both] - rc[both]) / np.maximum(1.0, np.abs(rc[both]))) > 1e-8:
            problems.append(f"{variant}: cost deviation {np.max(np.abs(cs[both] - rc[both]) / np.maximum(1.0, np.abs(rc[both]))):.3g}")
        if out.best_index != ro.best_index:
            # two FEASIBLE candidates whose costs agree to the last digits on BOTH sides: which of them sorts first is decided by
            # the order of the cost sum's additions (sixteen partial sums on the device, NumPy's pairwise sum per term in the
            # reference and the oracle) -- reported as what it is (seed 351029 of the round-5 soak: 1e-14 relative apart)
            a, b = int(out.best_index), int(ro.best_index)
            lo_ = lo if variant == "shard" else 0
            tie = False
            if a >= 0 and b >= 0:
                ia, ib = a - lo_, b - lo_
                if 0 <= ia < len(cs) and 0 <= ib < len(cs) and (st[ia] & 3) == (st[ib] & 3) == (rs[ia] & 3) == (rs[ib] & 3) == 1:
                    tie = abs(cs[ia] - cs[ib]) <= 1e-12 * max(1.0, abs(cs[ia])) and abs(rc[ia] - rc[ib]) <= 1e-12 * max(1.0, abs(rc[ia]))
            problems.append(f"{variant}: " + ("tie within rounding: " if tie else "") + f"winner {out.best_index} want {ro.best_index}" +
                            (f" (costs {cs[ia]!r} / {cs[ib]!r} here, {rc[ia]!r} / {rc[ib]!r} in the oracle)" if tie else ""))
        # (after a bounded sweep "collisions found" is the count in front of the winner: include/rp_amd.h, RP_PATH_SWEEP)
        found = ro.n_collision if (path == 3 and out.n_collision == out.n_collision_before_best) else out.n_collision + unlabelled
        if (out.n_feasible, found, out.n_collision_before_best) != (ro.n_feasible, ro.n_collision, ro.n_collision_before_best):
            problems.append(f"{variant}: counters {(out.n_feasible, out.n_collision, out.n_collision_before_best)} want "
                            f"{(ro.n_feasible, ro.n_collision, ro.n_collision_before_best)}")
        if not np.array_equal(out.reason_counts[:7], ro.reason_counts[:7]):
            problems.append(f"{variant}: reason counters {out.reason_counts[:7]} want {ro.reason_counts[:7]}")
        if out.best_index >= 0 and ro.best_index == out.best_index and out.best_states is not None:
            # 1e-6 absolute (the contract), relative above magnitude 1 -- as for the materialised blocks below: with the checks
            # that would reject them masked out, a winner can carry a curvature of -3.5e3 1/m or a d'' of -6.6e5 over a speed of
            # 1e-4 m/s, where 1e-5 absolute is 1e-9 .. 1e-11 relative (seeds 28063, 28615: profiles/r02_fuzz_parity.txt)
            dev = np.max(np.abs(out.best_states - ro.best_states) / np.maximum(1.0, np.abs(ro.best_states)))
            if not dev <= 1e-6:
                problems.append(f"{variant}: winner states deviate {dev:.3g} (relative above 1)")
        if variant == "materialize":
            states = ctx.fetch_states()
            have = ((rs & 3) == 1) | ((rs & 3) == 3) | (bool(info["draw"]) & ((rs & 3) == 2))
            if have.any():
                idx = np.flatnonzero(have)
                want = ref.states[idx]
                # 1e-6 absolute (the contract), relative above magnitude 1: candidates kept only for drawing carry values
                # of any size (curvature over a vanishing speed)
                err = np.abs(states[idx] - want) / np.maximum(1.0, np.abs(want))
                err = np.where(np.isnan(states[idx]) & np.isnan(want), 0.0, err)   # NaN on both sides
                # ... and from its first failing step on such a candidate is arithmetic on blown-up values (a velocity that
                # is the rounding residue of the polynomial at T divided by cos(theta) ~ 1e-5 decides whether the horizon
                # extension moves at all): compared before that step only.  (The draw-mode fixtures of tests/golden pin
                # complete state blocks of infeasible candidates against the reference.)
                infeasible = ((rs[idx] & 3) == 2)[:, None, None]
                from_failure = np.arange(want.shape[2])[None, None, :] >= (rs[idx] >> 8)[:, None, None]
                err = np.where(infeasible & from_failure, 0.0, err)
                # ... and so is a value beyond 1e12 of an infeasible candidate ahead of its failing step when the check that
                # would have caught it is masked out: an acceleration of -1.6e28 m/s^2 is s'^2 / cos(theta_cl) with the cosine a
                # rounding residue (theta_cl = +-pi/2 at a standstill start); seeds 44106 and 58603 of a 90 000-seed sweep
                err = np.where(infeasible & (np.abs(want) > 1e12), 0.0, err)
                if not np.nanmax(err) <= 1e-6 or np.isnan(err).any():
                    c, r, i = np.unravel_index(np.nanargmax(np.where(np.isnan(err), np.inf, err)), err.shape)
                    problems.append(f"materialize: states deviate {err[c, r, i]:.3g} (relative above 1; candidate {idx[c]} label {rs[idx[c]] & 3} status {rs[idx[c]]:#x} "
                                    f"array {r} step {i}: got {states[idx[c], r, i]!r} want {ref.states[idx[c], r, i]!r}; "
                                    f"{int((np.nanmax(err, axis=(1, 2)) > 1e-6).sum())} candidates, labels {sorted(set((rs[idx[np.nanmax(err, axis=(1, 2)) > 1e-6]] & 3).tolist()))})")
    return info, C, problems, ref.out
