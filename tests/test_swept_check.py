"""Continuous collision check of one trajectory (reference: reactive_planner.py:1049-1058).

The rectangle around two consecutive ego rectangles belongs to commonroad-dc (absent here); its definition (planar
form of FCL's OBB sum) lives in oracle/rp_oracle.c, tests/golden/_ref_shims.py (what the reference was run with for
the plan_arc_swept_* fixtures) and the HIP kernel.  CPU tests: known answers of the definition, enclosure as a
property, the time-index rule.  GPU tests: rp_check_swept against the oracle -- rectangles to 1e-9, hit segment exact."""
import math

import numpy as np
import pytest

from commonroad_rp_amd._capi import make_params
from commonroad_rp_amd.collision import ObstacleTables

HL, HW, WB = 2.254, 0.805, 1.4227


def _params(time_step0=0, factor=1, n=21):
    return make_params(dt=0.1, N=n - 1, factor=factor, time_step0=time_step0, low_vel_mode=False, lon_mode=0, constraint_mask=0,
                       flags=0, x0_lon=[0, 0, 0], x0_lat=[0, 0, 0], x0_orientation=0.0, wheelbase=2.5789, wb_rear_axle=WB,
                       length=2 * HL, width=2 * HW, a_max=11.5, v_switch=7.319, delta_max=1.066, v_delta_max=0.4)


def _oracle_tables(obstacles=None):
    from oracle.oracle import OracleTables
    s = np.arange(0.0, 50.0, 1.0)
    return OracleTables(s, np.zeros_like(s), np.zeros_like(s), np.zeros_like(s), np.stack((s, np.zeros_like(s)), 1), 20.0,
                        obstacles)


def _corners(cx, cy, ux, uy, hl, hw):
    return [(cx + a * hl * ux - b * hw * uy, cy + a * hl * uy + b * hw * ux) for a in (-1, 1) for b in (-1, 1)]


def _inside(box, pt, tol=1e-9):
    cx, cy, ux, uy, hl, hw = box
    dx, dy = pt[0] - cx, pt[1] - cy
    return abs(dx * ux + dy * uy) <= hl + tol and abs(-dx * uy + dy * ux) <= hw + tol


def _boxes(x, y, th, obstacles=None, **kw):
    from oracle import oracle
    return oracle.check_swept(_params(n=len(x), **kw), _oracle_tables(obstacles), x, y, th, want_boxes=True)


def test_known_answers():
    # same pose twice: the ego rectangle itself (centre = rear axle moved by wb_rear_axle along the heading)
    _, b = _boxes([1.0, 1.0], [2.0, 2.0], [0.3, 0.3])
    np.testing.assert_allclose(b[0], [1.0 + WB * math.cos(0.3), 2.0 + WB * math.sin(0.3), math.cos(0.3), math.sin(0.3), HL, HW],
                               atol=1e-12)
    # straight ahead by 1 m: half length grows by half the distance, centre in the middle
    _, b = _boxes([0.0, 1.0], [0.0, 0.0], [0.0, 0.0])
    np.testing.assert_allclose(b[0], [0.5 + WB, 0.0, 1.0, 0.0, HL + 0.5, HW], atol=1e-12)
    # sideways by 1 m (no heading change): half width grows
    _, b = _boxes([0.0, 0.0], [0.0, 1.0], [0.0, 0.0])
    np.testing.assert_allclose(b[0], [WB, 0.5, 1.0, 0.0, HL, HW + 0.5], atol=1e-12)
    # headings +-a about the x axis, same centre: bisector orientation, symmetric extents
    a = 0.2
    x = [-WB * math.cos(a), -WB * math.cos(a)]
    y = [-WB * math.sin(a), WB * math.sin(a)]
    _, b = _boxes(x, y, [a, -a])
    np.testing.assert_allclose(b[0], [0.0, 0.0, 1.0, 0.0, HL * math.cos(a) + HW * math.sin(a), HL * math.sin(a) + HW * math.cos(a)],
                               atol=1e-12)
    # headings pointing apart (pi - a vs a): the same rectangles as a flipped pair -> axis along y
    _, b = _boxes([0.0, 0.0], [0.0, 0.0], [math.pi / 2 + 0.1, -math.pi / 2 + 0.1 + 0.2])
    assert abs(abs(b[0][2] * math.cos(math.pi / 2 + 0.2) + b[0][3] * math.sin(math.pi / 2 + 0.2)) - 1.0) < 1e-12
    # far apart (> 2 * (HL + HL)): axis along the line through the centres, whatever the headings
    _, b = _boxes([0.0, 30.0], [0.0, 40.0], [1.0, 1.0])
    np.testing.assert_allclose(b[0][2:4], [0.6, 0.8], atol=1e-12)


def test_encloses_both_rectangles_and_is_tight():
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = 6
        th = rng.uniform(-math.pi, math.pi) + np.cumsum(rng.normal(0, 0.3, n))
        step = rng.uniform(0.0, 25.0)
        x = np.cumsum(rng.uniform(0, step, n) * np.cos(th))
        y = np.cumsum(rng.uniform(0, step, n) * np.sin(th))
        _, b = _boxes(x, y, th)
        for i in range(n - 1):
            pts = []
            for k in (i, i + 1):
                c, s = math.cos(th[k]), math.sin(th[k])
                pts += _corners(x[k] + WB * c, y[k] + WB * s, c, s, HL, HW)
            assert all(_inside(b[i], p) for p in pts)
            # tight: every side of the box is touched by some corner
            cx, cy, ux, uy, hl, hw = b[i]
            lx = [(p[0] - cx) * ux + (p[1] - cy) * uy for p in pts]
            ly = [-(p[0] - cx) * uy + (p[1] - cy) * ux for p in pts]
            assert abs(max(lx) - hl) < 1e-9 and abs(min(lx) + hl) < 1e-9 and abs(max(ly) - hw) < 1e-9 and abs(min(ly) + hw) < 1e-9
            assert abs(ux * ux + uy * uy - 1.0) < 1e-12


def test_time_index_rule_and_gap_detection():
    """An obstacle that exists at ONE time index, between the rectangles of poses 4 and 5: no pose touches it, segment 4
    does -- at scenario index time_step0 + 4, independent of planning.factor (:1050 starts the ego object at
    x_0.time_step and appends one shape per index)."""
    from oracle import oracle
    n = 12
    x = 12.0 * np.arange(n)          # 12 m per step: consecutive rectangles (4.5 m long) leave gaps
    y, th = np.zeros(n), np.zeros(n)
    gap_x = 0.5 * (x[4] + x[5]) + WB
    for t0, factor in ((0, 1), (7, 1), (7, 3)):
        for k_obs, expect in ((t0 + 4, 4), (t0 + 5, -1), (t0 + 4 * factor if factor > 1 else t0 + 3, -1)):
            dyn = np.full((1, 40, 5), np.nan)
            dyn[0, k_obs] = (gap_x, 0.0, 0.0, 0.3, 0.3)
            tb = _oracle_tables(ObstacleTables(dyn_obb=dyn, dyn_t0=0))
            first, _ = oracle.check_swept(_params(time_step0=t0, factor=factor, n=n), tb, x, y, th)
            assert first == expect, (t0, factor, k_obs, first)
    # static shapes of every kind are seen by every segment; the first one in time order is reported
    tb = _oracle_tables(ObstacleTables(static_circ=[[0.5 * (x[7] + x[8]) + WB, 0.0, 0.2]],
                                       static_tri=[[gap_x, -0.1, gap_x + 0.2, 0.0, gap_x, 0.1]]))
    assert oracle.check_swept(_params(n=n), tb, x, y, th)[0] == 4
    assert oracle.check_swept(_params(n=n), _oracle_tables(), x, y, th)[0] == -1
    assert oracle.check_swept(_params(n=1), tb, x[:1], y[:1], th[:1])[0] == -1


def _random_case(rng, n, n_dyn, n_static):
    th0 = rng.uniform(-math.pi, math.pi)
    th = th0 + np.cumsum(rng.normal(0, 0.05, n))
    v = rng.uniform(0.0, 30.0)
    x = 50.0 + np.cumsum(v * 0.1 * np.cos(th))
    y = -20.0 + np.cumsum(v * 0.1 * np.sin(th))
    t0 = int(rng.integers(0, 5))
    dyn = np.full((n_dyn, n + 8, 5), np.nan)
    for j in range(n_dyn):
        k = rng.integers(0, n)
        off = rng.normal(0, 4.0, 2)
        for q in range(dyn.shape[1]):
            if rng.random() < 0.8:
                dyn[j, q] = (x[k] + off[0] + 0.3 * q, y[k] + off[1], rng.uniform(-3, 3), rng.uniform(0.2, 2.5), rng.uniform(0.2, 1.2))
    sobb, tri, circ = [], [], []
    for j in range(n_static):
        k = rng.integers(0, n)
        px, py = x[k] + rng.normal(0, 6.0), y[k] + rng.normal(0, 6.0)
        kind = j % 3
        if kind == 0:
            sobb.append([px, py, rng.uniform(-3, 3), rng.uniform(0.2, 6.0), rng.uniform(0.05, 1.0)])
        elif kind == 1:
            tri.append([px, py, px + rng.uniform(0.2, 2), py + rng.uniform(-1, 1), px + rng.uniform(-1, 1), py + rng.uniform(0.2, 2)])
        else:
            circ.append([px, py, rng.uniform(0.1, 1.5)])
    return x, y, th, t0, ObstacleTables(static_obb=sobb, static_tri=tri, static_circ=circ, dyn_obb=dyn, dyn_t0=0)


@pytest.mark.gpu
def test_gpu_matches_oracle():
    from commonroad_rp_amd._capi import RpContext
    from oracle import oracle
    rng = np.random.default_rng(11)
    ctx = RpContext(0)
    tb0 = _oracle_tables()
    ctx.set_reference(tb0.ref_pos, tb0.ref_theta, tb0.ref_curv, tb0.ref_curv_d, np.stack((tb0.ref_x, tb0.ref_y), 1), 20.0)
    hits = 0
    for trial in range(120):
        n = int(rng.choice([2, 3, 21, 31, 64, 65, 66, 101, 130, 256, 257, 258, 1000, 4095]))   # (256 segments per pass of the kernel)
        x, y, th, t0, obs = _random_case(rng, n, n_dyn=int(rng.integers(0, 70)) if trial % 4 == 0 else int(rng.integers(0, 6)),
                                         n_static=int(rng.integers(0, 12)) if trial % 3 else 0)
        p = _params(time_step0=t0, n=n)
        ctx.set_obstacles(obs)
        want_first, want_boxes = oracle.check_swept(p, _oracle_tables(obs), x, y, th, want_boxes=True)
        got_first, got_boxes = ctx.check_swept(p, x, y, th, want_boxes=True)
        np.testing.assert_allclose(got_boxes, want_boxes, rtol=0, atol=1e-9)
        assert got_first == want_first, (trial, n)
        assert ctx.check_swept(p, x, y, th) == want_first
        hits += want_first >= 0
    assert 20 < hits < 110, hits      # both outcomes are exercised
    assert ctx.check_swept(_params(n=1), [0.0], [0.0], [0.0]) == -1
    ctx.set_obstacles(ObstacleTables())
    assert ctx.check_swept(_params(n=21), np.arange(21.0), np.zeros(21), np.zeros(21)) == -1
    with pytest.raises(ValueError):
        ctx.check_swept(_params(n=3), [0.0, 1.0, 2.0], [0.0, 1.0], [0.0, 0.0, 0.0])
    ctx.close()


# ---- obstacle side: planning.continuous_collision_check replaces every dynamic obstacle by its OBB sum
#      (reference: reactive_planner.py:238-245) -------------------------------------------------------------------
def test_obstacle_obb_sum_rows_against_oracle_and_rules():
    from commonroad_rp_amd.collision import obb_sum_rows
    from oracle import oracle
    rng = np.random.default_rng(5)
    dyn = np.full((6, 12, 5), np.nan)
    for j in range(6):
        a, b = sorted(rng.integers(0, 13, 2))
        for k in range(a, b):
            dyn[j, k] = (rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(-3.2, 3.2), rng.uniform(0.2, 3.0), rng.uniform(0.2, 1.5))
    dyn[0, :, :] = np.nan                      # never present
    dyn[1, 3:9] = dyn[1, 3]                    # standing still: the sum is the rectangle itself
    dyn[1, 3:9, 0] = 4.0
    dyn[2, 2:8, 2] = np.linspace(3.0, 3.3, 6)  # headings across +-pi
    mine, ref = obb_sum_rows(dyn), oracle.obb_sum_rows(dyn)
    np.testing.assert_array_equal(np.isnan(mine[..., 0]), np.isnan(ref[..., 0]))
    ok = ~np.isnan(ref[..., 0])
    np.testing.assert_allclose(mine[ok][:, [0, 1, 3, 4]], ref[ok][:, [0, 1, 3, 4]], rtol=0, atol=1e-12)
    dth = np.angle(np.exp(1j * (mine[ok][:, 2] - ref[ok][:, 2])))
    np.testing.assert_allclose(dth, 0.0, atol=1e-12)
    # one shape shorter: a step has a shape only if the obstacle exists at that step AND the next one
    present = ~np.isnan(dyn[..., 0])
    expect = np.zeros_like(present)
    expect[:, :-1] = present[:, :-1] & present[:, 1:]
    np.testing.assert_array_equal(ok, expect)
    np.testing.assert_allclose(mine[1, 3:8, [0, 1, 3, 4]], dyn[1, 3:8, [0, 1, 3, 4]], atol=1e-12)
    # enclosure: all eight corners of both rectangles lie inside their sum
    for j, k in zip(*np.nonzero(ok)):
        m = mine[j, k]
        box = (m[0], m[1], math.cos(m[2]), math.sin(m[2]), m[3], m[4])
        for r in (dyn[j, k], dyn[j, k + 1]):
            for c in _corners(r[0], r[1], math.cos(r[2]), math.sin(r[2]), r[3], r[4]):
                assert _inside(box, c, 1e-9)


def test_from_scenario_continuous_flag_reaches_the_tables():
    """ObstacleTables.from_scenario(continuous=True) == OBB sums of the per-step rectangles; a crossing obstacle that misses
    a parked ego at both of its poses hits it in between."""
    from _golden import scenario_from_tables
    from commonroad_rp_amd.collision import obb_sum_rows
    dyn = np.full((1, 6, 5), np.nan)
    dyn[0, 2] = (10.0, 3.0, -0.5 * math.pi, 0.4, 0.2)
    dyn[0, 3] = (10.0, -3.0, -0.5 * math.pi, 0.4, 0.2)
    tb = ObstacleTables(dyn_obb=dyn, dyn_t0=4)
    sc = scenario_from_tables(tb)
    plain = ObstacleTables.from_scenario(sc)
    swept = ObstacleTables.from_scenario(sc, continuous=True)
    assert plain.dyn_t0 == swept.dyn_t0 == 6         # tables start at the obstacle's first time step
    np.testing.assert_allclose(plain.dyn_obb[0, :2], dyn[0, 2:4])
    np.testing.assert_allclose(swept.dyn_obb, obb_sum_rows(plain.dyn_obb))
    m = swept.dyn_obb[0, 0]
    np.testing.assert_allclose([m[0], m[1], m[3], m[4]], [10.0, 0.0, 3.4, 0.2], atol=1e-12)
    assert np.isnan(swept.dyn_obb[0, 1, 0])
    # oracle: a parked ego at x = 10 - WB (centre at 10, 0) collides only with the swept table, and only at time index 6
    from oracle import oracle
    x, y, th = np.full(3, 10.0 - WB), np.zeros(3), np.zeros(3)
    for tables, t0, hit in ((plain, 5, -1), (swept, 5, 1), (swept, 6, 0), (swept, 7, -1)):
        first = oracle.check_swept(_params(time_step0=t0, n=3), _oracle_tables(tables), x, y, th)[0]
        assert first == hit, (t0, first, hit)
