#!/usr/bin/env python3
"""Golden vectors of the reference ON BASELINE.json's OWN CONFIGURATIONS: the workloads of
``commonroad_rp_amd/workloads.py`` -- cfg1 (ZAM_Over-1_1, the reference's sampling levels 1-3), cfg2 (ZAM_Tjunction 15x15x31),
cfg3 (DEU_Test 31x31x63 + 50 obstacles), cfg4 (ZAM_Tjunction 63x63x127; every 8th time sample: 65 024 candidates) -- each with
and without the road boundary, cfg1-cfg3 also with ``draw_traj_set`` -- run through the reference's own
``generate_trajectories_at_level -> _check_kinematics -> TrajectoryBundle.sort -> _check_collisions``
(/root/reference/commonroad_rp, unmodified, under the third-party stand-ins of ``_ref_shims.py``; planner set-up shared with
``make_golden.py``).

Runs ONLY in the build container.  Output: ``tests/golden/cfg{1,2,3,4}_ref*.npz`` -- data only:
  inputs  : as the fixtures of make_golden.py (tables, grids in the reference's set-iteration order, POD parameters, obstacles)
  outputs : per candidate label (u8), first-failure reason (u8), cost (f64, NaN where the reference has none);
            winner, counters, reason counts; 14-row states of a stratified subset of <= 64 candidates (+ their polynomial
            coefficients); the reference's own collision verdict for a sample of <= 2 000 feasible candidates (``collide_index``
            / ``collide``: the lazy walk only labels up to the winner).

    python tests/golden/make_baseline_golden.py                 # everything (about 25 min on 8 cores: --jobs 6)
    python tests/golden/make_baseline_golden.py --only cfg3_ref
    python tests/golden/make_baseline_golden.py --list
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg   # noqa: E402  (installs the stand-ins, imports the reference)

from commonroad_rp_amd import workloads as W   # noqa: E402
from commonroad_rp.trajectories import FeasibilityStatus, TrajectoryBundle   # noqa: E402  (reference)

CFG4_T_STRIDE = 8


def baseline_cases():
    cs = []
    for rb in (False, True):
        for draw in (False, True):
            sfx = ("_rb" if rb else "") + ("_draw" if draw else "")
            for level in (1, 2, 3):
                cs.append(dict(name=f"cfg1_ref_l{level}{sfx}", workload="cfg1", kw=dict(level=level, road_boundary=rb), draw=draw,
                               native_level=level))
            cs.append(dict(name=f"cfg2_ref{sfx}", workload="cfg2", kw=dict(road_boundary=rb), draw=draw))
            cs.append(dict(name=f"cfg3_ref{sfx}", workload="cfg3", kw=dict(road_boundary=rb), draw=draw))
        cs.append(dict(name="cfg4_ref" + ("_rb" if rb else ""), workload="cfg4", kw=dict(road_boundary=rb), draw=False,
                       t_stride=CFG4_T_STRIDE))
    return cs


def case_of(bc):
    """make_golden.py case (what make_planner reads) of a BASELINE workload."""
    w = W.WORKLOADS[bc["workload"]](**bc["kw"])
    p, c = w.inputs.params, w.inputs.cost
    v0 = float(c.desired_speed)            # (_scenario_workload: desired speed = the scenario's initial speed)
    low = bool(p.low_vel_mode)
    case = dict(name=bc["name"], dt=float(p.dt), N=int(p.N), factor=int(p.factor), coordinate_system=w.coordinate_system,
                ref_path=w.coordinate_system.reference, x0_lon=list(p.x0_lon), x0_lat=list(p.x0_lat),
                x0_orientation=float(p.x0_orientation), x0_velocity=v0, desired_speed=v0, time_step0=int(p.time_step0),
                obstacles=w.obstacles, low_vel_mode_threshold=(v0 + 1.0) if low else 0.5 * v0, draw=bool(bc["draw"]),
                max_states=64)
    T, L, D = w.inputs.T, w.inputs.L, w.inputs.D
    if "native_level" in bc:   # the reference's own sampling levels (configurations/ZAM_Over-1_1.yaml: t_min 0.2, d in [-3, 3])
        case.update(level=bc["native_level"], t_min=0.2, d_min=-3, d_max=3)
    else:                      # explicit grids (denser than the level scheme), installed as the level-1 sample sets
        Tq = T[::bc.get("t_stride", 1)]
        Dq = D[:-1] if (len(D) > 1 and D[-1] == p.x0_lat[0] and not np.any(D[:-1] == p.x0_lat[0])) else D   # (the reference appends d0 itself)
        case.update(level=1, t_min=float(max(2 * p.dt, Tq.min())), custom_T=[float(t) for t in Tq], custom_L=[float(v) for v in L],
                    custom_D=[float(d) for d in Dq])
    return case, w


def run_baseline_case(bc, progress=None):
    case, w = case_of(bc)
    t_start = time.time()
    rp = mg.make_planner(case)
    N, dt = rp.N, rp.dt
    # ---- pass 1: the real call path -> labels, costs, winner, counters
    bundle = mg.sample(rp, case)
    all_trajs = list(bundle.trajectories)
    C = len(all_trajs)
    opt = rp._get_optimal_trajectory(bundle)
    winner = all_trajs.index(opt) if opt is not None else -1
    counts = dict(rp._infeasible_reason_dict)
    n_kin, n_col = rp._infeasible_count_kinematics, rp._infeasible_count_collision
    if progress:
        progress(f"pass 1 done ({C} candidates, {time.time() - t_start:.0f} s)")

    # ---- grids in the reference's iteration order, from the list itself
    Ts, Ls, Ds = [], [], []
    seenT, seenL, seenD = set(), set(), set()
    for t in all_trajs:
        T, L, D = t.trajectory_long.delta_tau, t.trajectory_long.x_d[0], t.trajectory_lat.x_d[0]
        if T not in seenT:
            seenT.add(T); Ts.append(T)
        if L not in seenL:
            seenL.add(L); Ls.append(L)
        if D not in seenD:
            seenD.add(D); Ds.append(D)
    nT, nL, nD = len(Ts), len(Ls), len(Ds)
    assert nT * nL * nD == C, (nT, nL, nD, C)
    for i in range(0, C, max(1, C // 5000)):   # the list is the T-major / L / D-minor product (spot check: every k-th entry)
        t = all_trajs[i]
        iT, r = divmod(i, nL * nD)
        iL, iD = divmod(r, nD)
        assert t.trajectory_long.delta_tau == Ts[iT] and t.trajectory_long.x_d[0] == Ls[iL] and t.trajectory_lat.x_d[0] == Ds[iD]
    # the workload's grids are these sets (its ORDER is its own: sorted; the fixture keeps the reference's)
    wT = w.inputs.T[::bc.get("t_stride", 1)]
    assert sorted(Ts) == sorted(float(x) for x in wT), "time samples differ from the workload's"
    assert np.allclose(sorted(Ls), sorted(w.inputs.L), rtol=0, atol=1e-12) and sorted(Ds) == sorted(float(x) for x in w.inputs.D)
    traj_len = [len(np.arange(0, np.round(T + dt, 5), dt)) for T in Ts]

    label = np.array([mg.LABEL[t.feasibility_label] for t in all_trajs], dtype=np.uint8)
    has_cost = (label == 1) | (label == 3)
    cost = np.array([t.cost if h else np.nan for t, h in zip(all_trajs, has_cost)])

    # ---- states of a stratified subset: winner + candidates spread over the index range (and over the labels that have states)
    have = np.array([i for i, t in enumerate(all_trajs) if t.cartesian is not None], dtype=np.int64)
    rng = np.random.default_rng(0)
    keep = set()
    if winner >= 0:
        keep.add(winner)
    if len(have):
        strata = np.array_split(have, min(len(have), case["max_states"] - 1))
        for s_ in strata:
            keep.add(int(rng.choice(s_)))
    keep = sorted(keep)[:case["max_states"]]
    if winner >= 0 and winner not in keep:
        keep[-1] = winner
        keep = sorted(keep)
    states = np.zeros((len(keep), 14, N + 1))
    for k, i in enumerate(keep):
        t = all_trajs[i]
        for j, nme in enumerate(mg.CART):
            states[k, j] = getattr(t.cartesian, nme)
        for j, nme in enumerate(mg.CURV):
            states[k, 7 + j] = getattr(t.curvilinear, nme)
    lon_c = np.array([all_trajs[i].trajectory_long.coeffs for i in keep]).reshape(len(keep), -1)
    lat_c = np.array([all_trajs[i].trajectory_lat.coeffs for i in keep]).reshape(len(keep), -1)

    # ---- the reference's own collision verdict for a sample of the feasible candidates (the lazy walk labels up to the winner only)
    feas = np.flatnonzero(has_cost)
    sample_n = min(len(feas), 2000)
    coll_idx = np.sort(rng.choice(feas, size=sample_n, replace=False)) if sample_n else np.zeros(0, dtype=np.int64)
    coll = np.zeros(len(coll_idx), dtype=np.uint8)
    rp3 = mg.make_planner(case)
    for k, i in enumerate(coll_idx):
        t = all_trajs[i]
        b1 = TrajectoryBundle([t], cost_function=rp3.cost_function)
        b1._is_sorted = True
        lab = t.feasibility_label
        coll[k] = int(rp3._check_collisions(b1) is None)
        t.feasibility_label = lab
    assert np.all(coll[label[coll_idx] == 3] == 1)
    if progress:
        progress(f"collision sample done ({time.time() - t_start:.0f} s)")
    del all_trajs, bundle, opt
    for t in (rp.stored_trajectories or []):
        t._cartesian = t._curvilinear = None

    # ---- pass 2: one candidate at a time -> per-candidate first-failure reason (arrays dropped as we go)
    rp2 = mg.make_planner(case)
    bundle2 = mg.sample(rp2, case)
    reason = np.zeros(C, dtype=np.uint8)
    for c in mg.REASONS:
        rp2._infeasible_reason_dict[c] = 0
    trajs2 = bundle2.trajectories
    assert len(trajs2) == C
    for i in range(C):
        t = trajs2[i]
        before = dict(rp2._infeasible_reason_dict)
        feas_l, _ = rp2._check_kinematics([t])
        diff = [k for k in mg.REASONS if rp2._infeasible_reason_dict[k] != before[k]]
        assert len(diff) <= 1
        if diff:
            reason[i] = mg.REASONS.index(diff[0]) + 1
        elif not feas_l:
            reason[i] = 6   # rejected without a reason counter: outside the projection domain
        t._cartesian = t._curvilinear = None   # (the setters insist on sample objects)
        trajs2[i] = None
        if progress and i % 20000 == 19999:
            progress(f"pass 2: {i + 1} / {C} ({time.time() - t_start:.0f} s)")
    # the one-at-a-time pass agrees with the list pass
    assert [int((reason == k + 1).sum()) for k in range(5)] == [counts.get(k, 0) for k in mg.REASONS], "reason counts differ between the passes"
    if not case["draw"]:
        assert int((reason == 0).sum()) == int(has_cost.sum())

    cf = rp.cost_function
    tb = case["obstacles"]
    return dict(
        # inputs
        workload=bc["workload"], road_boundary=int(bool(bc["kw"].get("road_boundary"))), t_stride=int(bc.get("t_stride", 1)),
        ref_path=rp._co.reference, ref_pos=rp._co.ref_pos, ref_theta=rp._co.ref_theta, ref_curv=rp._co.ref_curv,
        ref_curv_d=rp._co.ref_curv_d, proj_d_limit=rp._co.proj_domain_d_limit,
        T=np.array(Ts), traj_len=np.array(traj_len, dtype=np.int32), L=np.array(Ls), D=np.array(Ds),
        dt=dt, N=N, factor=rp.config.planning.factor, time_step0=rp.x_0.time_step, low_vel_mode=int(rp._low_vel_mode), lon_mode=0,
        draw=int(rp._draw_traj_set), constraint_mask=31, x0_lon=np.array(rp.x_0_cl[0], dtype=float),
        x0_lat=np.array(rp.x_0_cl[1], dtype=float), x0_orientation=rp.x_0.orientation, x0_velocity=rp.x_0.velocity,
        vehicle=np.array([mg.VEH[k] for k in ("wheelbase", "wb_rear_axle", "length", "width", "a_max", "v_switch", "delta_max", "v_delta_max")]),
        cost_kind=0, w_a=float(cf.w_a), desired_speed=cf.desired_speed, desired_d=cf.desired_d, desired_s=np.nan,
        static_obb=tb.static_obb, static_tri=tb.static_tri, static_circ=tb.static_circ, dyn_obb=tb.dyn_obb, dyn_t0=tb.dyn_t0,
        # outputs
        label=label, reason=reason, cost=cost, state_index=np.array(keep, dtype=np.int64), states=states, lon_coeffs=lon_c, lat_coeffs=lat_c,
        collide_index=coll_idx.astype(np.int64), collide=coll,
        winner=winner, winner_cost=(cost[winner] if winner >= 0 else np.nan),
        n_infeasible_kinematics=n_kin, n_infeasible_collision=n_col,
        reason_counts=np.array([counts.get(k, 0) for k in mg.REASONS], dtype=np.int64))


# ----------------------------------------------------------------------------------------------
# Closed loops over the scenarios BASELINE.json's configurations name, driven by the reference planner itself through the loop of
# run_planner.py:61-107 (commonroad_rp_amd.harness.run_closed_loop: the reference's public methods only) with the settings of the
# shipped YAML files (configurations/*.yaml: N = 20, t_min, low_vel_mode_threshold, draw_traj_set and (show|save)_plots), the
# scenario's own obstacles and the road boundary (set_collision_checker(scenario=...) always adds it, reactive_planner.py:246-250).
# ----------------------------------------------------------------------------------------------
LOOPS = {
    # name: (scenario fixture, t_min, low_vel_mode_threshold, draw, replanning frequency, simulated steps)
    # ZAM_Over: 20 m/s on a 200-m route -- 60 steps; replanning every step (the YAML's 3 would leave 20 cycles)
    "loop_zam_over": ("ZAM_Over-1_1", 0.2, 4.0, False, 1, 60),
    # ZAM_Tjunction: the YAML's frequency 3 (draw_traj_set without plots: _draw_traj_set stays off, reactive_planner.py:97)
    "loop_zam_tjunction": ("ZAM_Tjunction-1_42_T-1", 1.0, 2.0, False, 3, 96),
    # DEU_Test: save_plots + draw_traj_set -> the whole set is evaluated and stored every cycle (no pre-filter, no early exit)
    "loop_deu_test": ("DEU_Test-1_1_T-1", 0.4, 4.0, True, 2, 70),
}


def loop_case(name):
    scen, t_min, low_thr, draw, freq, steps = LOOPS[name]
    w = W._scenario_workload(name, scen, 20, [1.0], 3, 3, low_vel_threshold=low_thr, road_boundary=True)
    p, c = w.inputs.params, w.inputs.cost
    v0 = float(c.desired_speed)
    return dict(name=name, dt=float(p.dt), N=20, t_min=t_min, coordinate_system=w.coordinate_system, ref_path=w.coordinate_system.reference,
                level=1, x0_lon=list(p.x0_lon), x0_lat=list(p.x0_lat), x0_orientation=float(p.x0_orientation), x0_velocity=v0,
                desired_speed=v0, time_step0=int(p.time_step0), obstacles=w.obstacles, low_vel_mode_threshold=low_thr, draw=draw,
                steps=steps, replanning_frequency=freq)


def run_scenario_loop(case):
    from commonroad_rp_amd.harness import run_closed_loop
    from commonroad_rp_amd.coordinate_system import CoordinateSystem
    rp = mg.make_planner(case)
    rp.record_state_and_input = lambda state: rp._record_state_list.append(state)   # (InputState is a stand-in here)
    inputs = mg.case_inputs(rp, case)
    co2 = CoordinateSystem(inputs["ref_path"], float(inputs["proj_d_limit"]))   # what the tests rebuild from the fixture: the same tables
    for k in ("ref_pos", "ref_theta", "ref_curv", "ref_curv_d", "reference"):
        assert np.array_equal(getattr(co2, k), getattr(rp._co, k)), k
    from _golden import record_plan_stats
    stats = record_plan_stats(rp)   # counters of every plan() call (reset() clears them before the loop's callback runs)
    res = run_closed_loop(rp, max_steps=case["steps"], replanning_frequency=case["replanning_frequency"])
    trace = np.array([[s.time_step, s.position[0], s.position[1], s.orientation, s.velocity, s.acceleration or 0.0, s.steering_angle or 0.0]
                      for s in res.states], dtype=float)
    return dict(inputs, completed=int(res.completed), n_replans=res.n_replans, trace=trace, plan_stats=np.array(stats, dtype=np.int64),
                steps=case["steps"], replanning_frequency=case["replanning_frequency"], draw=int(case["draw"]))


def _one_loop(name):
    t0 = time.time()
    out = run_scenario_loop(loop_case(name))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    fl = out["plan_stats"]
    return (f"{name:22s} completed={out['completed']} steps={len(out['trace']) - 1} replans={out['n_replans']} "
            f"collisions before the winner per plan: {int(fl[:, 3].min())}..{int(fl[:, 3].max())}, kinematic rejects: {int(fl[:, 2].min())}..{int(fl[:, 2].max())}, "
            f"stored: {int(fl[:, 9].max())} "
            f"final speed={out['trace'][-1, 4]:.3f} m/s at ({out['trace'][-1, 1]:.2f}, {out['trace'][-1, 2]:.2f}) time={time.time() - t0:.0f}s")


def _one(name):
    if name in LOOPS:
        return _one_loop(name)
    bc = next(c for c in baseline_cases() if c["name"] == name)
    t0 = time.time()
    out = run_baseline_case(bc, progress=lambda m: print(f"  [{name}] {m}", flush=True))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    lab = out["label"]
    return (f"{name:22s} C={len(lab):6d} feasible={int((lab == 1).sum()):6d} kin={int((lab == 2).sum()):6d} coll={int((lab == 3).sum()):6d} "
            f"none={int((lab == 0).sum()):6d} reasons={out['reason_counts'].tolist()} ood={int((out['reason'] == 6).sum())} "
            f"winner={out['winner']} cost={out['winner_cost']:.6g} ncol={out['n_infeasible_collision']} "
            f"sample colliding={int(out['collide'].sum())}/{len(out['collide'])} states={out['states'].shape[0]} "
            f"size={os.path.getsize(path) / 1024:.0f}KB time={time.time() - t0:.0f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated case names or prefixes")
    ap.add_argument("--jobs", type=int, default=1)
    args = ap.parse_args()
    names = [c["name"] for c in baseline_cases()] + list(LOOPS)
    if args.only:
        want = args.only.split(",")
        names = [n for n in names if any(n == q or n.startswith(q + "_") for q in want)]
    if args.list:
        print("\n".join(names))
        return
    if args.jobs > 1:
        import multiprocessing as mp
        big_first = sorted(names, key=lambda n: (not n.startswith(("cfg4", "cfg3")), n))
        with mp.get_context("spawn").Pool(args.jobs, maxtasksperchild=1) as pool:
            for line in pool.imap_unordered(_one, big_first):
                print(line, flush=True)
    else:
        for n in names:
            print(_one(n), flush=True)


if __name__ == "__main__":
    main()
