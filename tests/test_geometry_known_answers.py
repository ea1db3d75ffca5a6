"""Analytic known answers for the two geometric definitions this build makes itself (last row): the
overlap predicate of the collision query (pycrcc's source is not under /root/reference) and the (s, d) -> (x, y)
transform (CCosy's is not either).

Overlap: CLOSED sets -- exactly touching shapes collide.  Rectangles touching / separated / contained, rotated by
0, pi/4 and pi/2; rectangle against triangle and circle.  Exact touching is asked only where the touching
configuration is exactly representable (axis-parallel, dyadic numbers); rotated cases straddle the contact by 1e-9.
Transform: a straight line in general position (exact) and a circle of known radius (to the polyline's sagitta).

CPU: the oracle (the checker of the HIP path must itself give the analytic answers).  GPU (`-m gpu`): the HIP path
through the C ABI -- `rp_check_swept` with two identical poses IS the per-pose predicate (the rectangle around a
rectangle and itself is that rectangle), and `rp_plan` for labels and state rows.
"""
import math

import numpy as np
import pytest

from commonroad_rp_amd._capi import (FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL, PlanInputs, make_cost, make_params)
from commonroad_rp_amd.collision import ObstacleTables
from commonroad_rp_amd.coordinate_system import CoordinateSystem

# an "ego" with dyadic half extents and the reference point in its centre: contact configurations are exact
HL, HW = 2.0, 0.75
S2 = math.sqrt(0.5)


def _params(n=2, wb_rear_axle=0.0, **kw):
    base = dict(dt=0.125, N=n - 1, factor=1, time_step0=0, low_vel_mode=False, lon_mode=0, constraint_mask=0, flags=0,
                x0_lon=[0, 0, 0], x0_lat=[0, 0, 0], x0_orientation=0.0, wheelbase=2.5, wb_rear_axle=wb_rear_axle,
                length=2 * HL, width=2 * HW, a_max=11.5, v_switch=7.319, delta_max=1.066, v_delta_max=0.4)
    base.update(kw)
    return make_params(**base)


class _OracleProbe:
    """ego rectangle (centre x, y, heading th) against obstacle tables -> bool, through the oracle"""

    def __call__(self, tables, x, y, th):
        from oracle import oracle
        s = np.arange(0.0, 8.0, 1.0)
        tb = oracle.OracleTables(s, 0 * s, 0 * s, 0 * s, np.stack((s, 0 * s), 1), 20.0, tables)
        return oracle.check_swept(_params(), tb, [x, x], [y, y], [th, th])[0] == 0


class _GpuProbe:
    def __init__(self, ctx):
        self.ctx = ctx

    def __call__(self, tables, x, y, th):
        self.ctx.set_obstacles(tables)
        return self.ctx.check_swept(_params(), [x, x], [y, y], [th, th]) == 0


def _rect(cx, cy, th, hl, hw, as_dynamic=False):
    if as_dynamic:
        return ObstacleTables(dyn_obb=np.array([[[cx, cy, th, hl, hw]]]), dyn_t0=0)
    return ObstacleTables(static_obb=[[cx, cy, th, hl, hw]])


def _overlap_known_answers(hit):
    eps = 1e-9
    for dyn in (False, True):
        R = lambda *a: _rect(*a, as_dynamic=dyn)   # noqa: E731
        # -- axis-parallel: touching along an edge, at a corner; separated by one ulp-scale step; contained both ways
        assert hit(R(4.0 + 1.0, 0.0, 0.0, 1.0, 0.5), 2.0, 0.0, 0.0)                  # edges touch: x = 4
        assert not hit(R(4.0 + 1.0 + eps, 0.0, 0.0, 1.0, 0.5), 2.0, 0.0, 0.0)
        assert hit(R(0.0, 0.75 + 0.5, 0.0, 1.0, 0.5), 0.0, 0.0, 0.0)                 # edges touch: y = 0.75
        assert not hit(R(0.0, 0.75 + 0.5 + eps, 0.0, 1.0, 0.5), 0.0, 0.0, 0.0)
        assert hit(R(3.0, 1.25, 0.0, 1.0, 0.5), 0.0, 0.0, 0.0)                       # corners touch: (2, 0.75)
        assert not hit(R(3.0 + eps, 1.25 + eps, 0.0, 1.0, 0.5), 0.0, 0.0, 0.0)
        assert hit(R(0.25, 0.125, 0.0, 0.5, 0.25), 0.0, 0.0, 0.0)                    # obstacle inside the ego
        assert hit(R(0.0, 0.0, 0.0, 8.0, 8.0), 1.0, -1.0, 0.0)                       # ego inside the obstacle
        assert not hit(R(0.0, 4.0, 0.0, 9.0, 0.5), 0.0, 0.0, 0.0)                    # long strip beside the ego
        # -- obstacle rotated by pi/2: a 1 x 0.5 rectangle stands upright; touching distance along x = 2 + 0.5
        assert hit(R(2.5 - eps, 0.0, 0.5 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.0)
        assert not hit(R(2.5 + eps, 0.0, 0.5 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.0)
        assert hit(R(0.0, 0.75 + 1.0 - eps, 0.5 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.0)
        assert not hit(R(0.0, 0.75 + 1.0 + eps, 0.5 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.0)
        # -- obstacle rotated by pi/4: a square of half side 1 is a diamond of half diagonal sqrt 2;
        #    its tip touches the ego's right edge at x = 2, its edge passes the ego's corner (2, 0.75)
        r2 = math.sqrt(2.0)
        assert hit(R(2.0 + r2 - eps, 0.0, 0.25 * math.pi, 1.0, 1.0), 0.0, 0.0, 0.0)
        assert not hit(R(2.0 + r2 + eps, 0.0, 0.25 * math.pi, 1.0, 1.0), 0.0, 0.0, 0.0)
        #    diamond centred at (2 + a, 0.75 + a): its lower-left edge x + y = cx + cy - sqrt 2 passes the corner when
        #    2 a = sqrt 2 -- neither rectangle's own axes separate there, only the diamond's edge normal does
        a = 0.5 * r2
        assert hit(R(2.0 + a - eps, 0.75 + a - eps, 0.25 * math.pi, 1.0, 1.0), 0.0, 0.0, 0.0)
        assert not hit(R(2.0 + a + eps, 0.75 + a + eps, 0.25 * math.pi, 1.0, 1.0), 0.0, 0.0, 0.0)
        # -- ego rotated by pi/4 as well (both rotated: parallel again), and ego rotated by pi/2 against an upright strip
        assert hit(R((2.0 + 1.0 - eps) * S2, (2.0 + 1.0 - eps) * S2, 0.25 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.25 * math.pi)
        assert not hit(R((2.0 + 1.0 + eps) * S2, (2.0 + 1.0 + eps) * S2, 0.25 * math.pi, 1.0, 0.5), 0.0, 0.0, 0.25 * math.pi)
        assert hit(R(0.75 + 0.5 - eps, 0.0, 0.0, 0.5, 3.0), 0.0, 0.0, 0.5 * math.pi)
        assert not hit(R(0.75 + 0.5 + eps, 0.0, 0.0, 0.5, 3.0), 0.0, 0.0, 0.5 * math.pi)
    # -- circle: touching an edge, touching a corner (distance measured to the corner), centre inside
    C = lambda cx, cy, r: ObstacleTables(static_circ=[[cx, cy, r]])   # noqa: E731
    assert hit(C(3.0, 0.0, 1.0), 0.0, 0.0, 0.0)                                      # touches the edge x = 2 exactly
    assert not hit(C(3.0 + eps, 0.0, 1.0), 0.0, 0.0, 0.0)
    assert hit(C(0.0, -1.25, 0.5), 0.0, 0.0, 0.0)
    assert not hit(C(0.0, -1.25 - eps, 0.5), 0.0, 0.0, 0.0)
    assert hit(C(2.0 + 3.0, 0.75 + 4.0, 5.0), 0.0, 0.0, 0.0)                         # 3-4-5 from the corner (2, 0.75): touches
    assert not hit(C(2.0 + 3.0 + 1e-6, 0.75 + 4.0, 5.0), 0.0, 0.0, 0.0)
    assert not hit(C(2.0 + 0.8, 0.75 + 0.8, 1.0), 0.0, 0.0, 0.0)                     # inside the corner's bounding square only
    assert hit(C(0.5, 0.25, 0.01), 0.0, 0.0, 0.0)
    assert hit(C(2.0 + S2 * (1.0 - 1e-9), 0.0, 1.0), 0.0, 0.0, 0.0) and hit(C(S2 * 3.0 - eps, S2 * 3.0 - eps, 1.0), 0.0, 0.0, 0.25 * math.pi)
    assert not hit(C(S2 * 3.0 + eps, S2 * 3.0 + eps, 1.0), 0.0, 0.0, 0.25 * math.pi)  # rotated ego: its front edge is 2 away
    # -- triangle: vertex on an edge, edge through a corner, ego inside a big triangle, triangle inside the ego
    T = lambda *v: ObstacleTables(static_tri=[list(v)])   # noqa: E731
    assert hit(T(2.0, 0.0, 4.0, 1.0, 4.0, -1.0), 0.0, 0.0, 0.0)                      # vertex touches the edge x = 2
    assert not hit(T(2.0 + eps, 0.0, 4.0, 1.0, 4.0, -1.0), 0.0, 0.0, 0.0)
    assert hit(T(2.0 + 1.0, 0.75 - 1.0, 2.0 - 1.0, 0.75 + 1.0, 4.0, 3.0), 0.0, 0.0, 0.0)   # edge x + y = 2.75 passes the corner
    assert not hit(T(2.0 + 1.0 + eps, 0.75 - 1.0 + eps, 2.0 - 1.0 + eps, 0.75 + 1.0 + eps, 4.0, 3.0), 0.0, 0.0, 0.0)
    assert hit(T(-20.0, -20.0, 20.0, -20.0, 0.0, 30.0), 0.0, 0.0, 0.3)
    assert hit(T(-0.5, -0.25, 0.5, -0.25, 0.0, 0.5), 0.0, 0.0, 0.0)
    assert not hit(T(-9.0, 3.0, 9.0, 3.0, 0.0, 4.0), 0.0, 0.0, 0.0)
    # -- nothing there
    assert not hit(ObstacleTables(), 0.0, 0.0, 0.0)


def test_overlap_predicate_known_answers_oracle():
    _overlap_known_answers(_OracleProbe())


def test_ego_rectangle_sits_ahead_of_the_rear_axle():
    """The planner state is the rear axle; the rectangle is centred wb_rear_axle ahead of it (reactive_planner.py:1033-1035)."""
    from oracle import oracle
    s = np.arange(0.0, 8.0, 1.0)
    for th in (0.0, 0.5 * math.pi):
        front = (1.0 + 2.0) * np.array([math.cos(th), math.sin(th)])     # centre 1 m ahead of the axle: the front is 3 m ahead
        for dist, want in ((3.5 - 1e-9, True), (3.5 + 1e-9, False)):           # a 1 m square touches it with its centre at 3.5 m
            p = dist * np.array([math.cos(th), math.sin(th)])
            tb = oracle.OracleTables(s, 0 * s, 0 * s, 0 * s, np.stack((s, 0 * s), 1), 20.0, _rect(p[0], p[1], th, 0.5, 0.5))
            got = oracle.check_swept(_params(wb_rear_axle=1.0), tb, [0.0, 0.0], [0.0, 0.0], [th, th])[0] == 0
            assert got == want, (th, dist, front)


# ---- (s, d) -> (x, y) -------------------------------------------------------------------------------------------
def _line_case(theta=0.6435011087932844, x0=-3.0, y0=7.0):   # atan(3/4): cos = 0.8, sin = 0.6
    s = np.arange(0.0, 121.0, 1.0)
    path = np.stack((x0 + 0.8 * s, y0 + 0.6 * s), axis=1)
    return CoordinateSystem(path), theta, (x0, y0)


def _circle_case(R=100.0, length=150.0, step=1.0):
    s = np.arange(0.0, length + 1e-9, step)
    # left turn, centre (0, R): position = R (sin(s/R), 1 - cos(s/R))
    return CoordinateSystem(np.stack((R * np.sin(s / R), R * (1.0 - np.cos(s / R))), axis=1)), R


def _drive(co, d0, n=31, v=8.0):
    """PlanInputs for candidates that keep the speed v and move from lateral offset d0 to the offsets D."""
    dt = 0.125
    T = np.array([dt * (n - 1)])
    p = make_params(dt=dt, N=n - 1, factor=1, time_step0=0, low_vel_mode=False, lon_mode=0, constraint_mask=0,
                    flags=FLAG_DRAW_ALL | FLAG_MATERIALIZE_ALL, x0_lon=[10.0, v, 0.0], x0_lat=[d0, 0.0, 0.0], x0_orientation=0.0,
                    wheelbase=2.5, wb_rear_axle=0.0, length=2 * HL, width=2 * HW, a_max=11.5, v_switch=7.319, delta_max=1.066,
                    v_delta_max=0.4)
    return PlanInputs(p, make_cost(desired_speed=v), T, np.array([n], dtype=np.int32), np.array([v]), np.array([-2.0, d0, 1.5]))


def _transform_known_answers(states_of):
    # straight line in general position: x = x0 + s cos - d sin, y = y0 + s sin + d cos, exactly
    co, th, (x0, y0) = _line_case()
    st = states_of(co, _drive(co, 0.5))
    for blk in st:
        x, y, s, d = blk[0], blk[1], blk[7], blk[8]
        np.testing.assert_allclose(x, x0 + 0.8 * s - 0.6 * d, rtol=0, atol=1e-9)
        np.testing.assert_allclose(y, y0 + 0.6 * s + 0.8 * d, rtol=0, atol=1e-9)
        np.testing.assert_allclose(blk[2] - blk[9], th, rtol=0, atol=1e-12)   # theta - theta_cl = heading of the line
        assert s[-1] - s[0] > 25.0 and abs(d[-1] - d[0]) <= 2.5 + 1e-9
    # circle of radius R, left turn: (s, d) lies at distance R - d from the centre (0, R), at polar angle s / R.
    # The reference path is a polyline with 1 m chords: sagitta h^2 / (8 R) = 1.25 mm bounds the radial error
    co, R = _circle_case()
    st = states_of(co, _drive(co, -0.5))
    for blk in st:
        x, y, s, d = blk[0], blk[1], blk[7], blk[8]
        rad = np.hypot(x, y - R)
        np.testing.assert_allclose(rad, R - d, rtol=0, atol=1.5e-3)
        ang = np.arctan2(x, R - y)                      # angle travelled around the centre
        # arc length is measured along the chords, which are shorter than the arc by a factor 1 - h^2 / (24 R^2)
        np.testing.assert_allclose(ang, s / R, rtol=0, atol=2e-5)
        assert np.all(rad[np.abs(d) > 0.1] != R)


def test_transform_known_answers_oracle():
    from oracle import oracle

    def states_of(co, inp):
        run = oracle.plan(inp, oracle.OracleTables.from_coordinate_system(co))
        assert np.all((run.status & 3) != 0)
        return run.states
    _transform_known_answers(states_of)
    # and the host-side definition the tables come from (commonroad_rp_amd.coordinate_system) agrees with both
    co, th, (x0, y0) = _line_case()
    np.testing.assert_allclose(co.convert_to_cartesian_coords(37.25, -1.5), [x0 + 0.8 * 37.25 + 0.6 * 1.5, y0 + 0.6 * 37.25 - 0.8 * 1.5],
                               atol=1e-12)
    np.testing.assert_allclose(co.convert_to_curvilinear_coords(x0 + 0.8 * 37.25 + 0.6 * 1.5, y0 + 0.6 * 37.25 - 0.8 * 1.5),
                               [37.25, -1.5], atol=1e-9)
    assert co.convert_to_cartesian_coords(37.25, 20.5) is None and co.convert_to_cartesian_coords(-0.1, 0.0) is None


def test_touching_trajectory_is_labelled_colliding_oracle():
    _touching_trajectory(lambda co, tables, inp: __import__("oracle.oracle", fromlist=["x"]).plan(
        inp, __import__("oracle.oracle", fromlist=["x"]).OracleTables.from_coordinate_system(co, tables)).status)


def _touching_trajectory(status_of):
    """A wall parallel to a straight route: the ego (d = 0 all the way) passes with its side exactly ON the wall's face --
    closed sets, so the trajectory collides; one nanometre farther out it does not."""
    s = np.arange(0.0, 121.0, 1.0)
    co = CoordinateSystem(np.stack((s, 0 * s), axis=1))
    inp = _drive(co, 0.0)
    inp = PlanInputs(inp.params, inp.cost, inp.T, inp.traj_len, inp.L, np.array([0.0]))
    inp.params.flags = 0
    wall = lambda y: ObstacleTables(static_obb=[[30.0, y, 0.0, 4.0, 0.5]])   # noqa: E731
    assert (status_of(co, wall(0.75 + 0.5), inp)[0] & 3) == 3
    assert (status_of(co, wall(-(0.75 + 0.5)), inp)[0] & 3) == 3
    assert (status_of(co, wall(0.75 + 0.5 + 1e-9), inp)[0] & 3) == 1
    # a wall across the road that ends exactly where the trajectory's last pose ends (s = 10 + 8 * 3.75 = 40, front at 42)
    assert (status_of(co, ObstacleTables(static_obb=[[42.0 + 0.25, 0.0, 0.0, 0.25, 3.0]]), inp)[0] & 3) == 3
    assert (status_of(co, ObstacleTables(static_obb=[[42.0 + 0.25 + 1e-6, 0.0, 0.0, 0.25, 3.0]]), inp)[0] & 3) == 1


# ---- the same through the C ABI on the GPU ----------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_ctx():
    from commonroad_rp_amd._capi import RpContext
    c = RpContext(0)
    s = np.arange(0.0, 8.0, 1.0)
    c.set_reference(s, 0 * s, 0 * s, 0 * s, np.stack((s, 0 * s), 1), 20.0)
    yield c
    c.close()


@pytest.mark.gpu
def test_overlap_predicate_known_answers_gpu(gpu_ctx):
    _overlap_known_answers(_GpuProbe(gpu_ctx))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["single_launch", "two_kernel", "g64"])
def test_transform_and_touching_known_answers_gpu(path):
    from _paths import launch_path_env
    from commonroad_rp_amd._capi import RpContext
    with launch_path_env(path):
        ctx = RpContext(0)

        def states_of(co, inp):
            ctx.set_coordinate_system(co)
            ctx.set_obstacles(ObstacleTables())
            ctx.plan(inp)
            assert np.all((ctx.fetch_status()[0] & 3) != 0)
            return ctx.fetch_states()

        def status_of(co, tables, inp):
            ctx.set_coordinate_system(co)
            ctx.set_obstacles(tables)
            ctx.plan(inp)
            return ctx.fetch_status()[0]
        _transform_known_answers(states_of)
        _touching_trajectory(status_of)
        ctx.close()
