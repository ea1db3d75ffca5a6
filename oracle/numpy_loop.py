"""Loop-faithful NumPy restatement of the reference's hot path (TEST INFRASTRUCTURE: imported by tests/ and by
bench.py's ``cpu_baseline`` leg only, never by the product).

SURVEY.md 8(d)(i): the reference is pure Python with NumPy -- one Python iteration per candidate, NumPy array
expressions per trajectory, a Python loop with scalar arithmetic over the time steps, NumPy again for the horizon
extension and the cost, then a sort and a lazy collision walk.  ``oracle/rp_oracle.c`` restates the same algorithm in C,
which is a far stronger CPU baseline than the reference itself; this module keeps the reference's *execution model* so
that the bench can also quote the rate of "the reference's kind of program" on the GPU box's host, where the reference
cannot travel.  Paths relative to /root/reference/commonroad_rp:

  sampling (A3)                      sampling.py:202-242
  quartic / quintic solve (A4, A5)   polynomial_trajectory.py:292-320, 341-360  (np.linalg.solve per polynomial)
  per-trajectory evaluation (A8)     reactive_planner.py:731-960
  constraints (A9)                   reactive_planner.py:971-1017
  horizon extension (A10, A11)       trajectories.py:168-197, 302-332
  cost (A13)                         cost_function.py:51-71, 82-92
  sort + lazy collision walk (A14/15) trajectories.py:502-510, reactive_planner.py:1031-1063

The collision predicate and (s, d) -> (x, y) are this build's own definitions (third-party boundary)
and are the scalar formulas of oracle/rp_oracle.c.  Parity status: pinned through tests/test_numpy_loop.py against the C
oracle (which the golden fixtures pin) -- labels, costs, winner and state rows on the fixtures' inputs.
"""
from __future__ import annotations

import math

import numpy as np

EPS = 1e-5   # reactive_planner.py: _EPS
# rows of a state block (include/rp_amd.h)
X, Y, THETA, V, A, KAPPA, KAPPA_DOT, S, D, THETA_CL, S_DOT, S_DDOT, D_DOT, D_DDOT = range(14)
LABEL_NONE, LABEL_FEASIBLE, LABEL_KINEMATIC, LABEL_COLLISION = 0, 1, 2, 3
REASON = {"none": 0, "velocity": 1, "acceleration": 2, "kappa": 3, "kappa_dot": 4, "yaw_rate": 5, "domain": 6}
CHECK_VELOCITY, CHECK_ACCELERATION, CHECK_KAPPA, CHECK_KAPPA_DOT, CHECK_YAW_RATE = 1, 2, 4, 8, 16
TWO_PI = 2.0 * math.pi


# ---- polynomials ----------------------------------------------------------------------------------------------------------
def quintic(x0, xd, T):
    """polynomial_trajectory.py:292-320"""
    t2 = T * T
    t3, t4 = t2 * T, t2 * t2
    t5 = t4 * T
    a = np.array([[t3, t4, t5], [3. * t2, 4. * t3, 5. * t4], [6. * T, 12. * t2, 20. * t3]])
    b = np.array([xd[0] - (x0[0] + x0[1] * T + .5 * x0[2] * t2), xd[1] - (x0[1] + x0[2] * T), xd[2] - x0[2]])
    try:
        h = np.linalg.solve(a, b)
    except np.linalg.LinAlgError:
        h = np.full(3, np.nan)
    return np.array([x0[0], x0[1], .5 * x0[2], h[0], h[1], h[2]])


def quartic(x0, vd, T):
    """polynomial_trajectory.py:341-360"""
    t2 = T * T
    t3 = t2 * T
    a = np.array([[3. * t2, 4. * t3], [6. * T, 12. * t2]])
    b = np.array([vd - x0[1] - x0[2] * T, -x0[2]])
    try:
        h = np.linalg.solve(a, b)
    except np.linalg.LinAlgError:
        h = np.full(2, np.nan)
    return np.array([x0[0], x0[1], .5 * x0[2], h[0], h[1], 0.])


def _pos(c, t, t2, t3, t4, t5):    # polynomial_trajectory.py:258-271
    return c[0] + c[1] * t + c[2] * t2 + c[3] * t3 + c[4] * t4 + c[5] * t5


def _vel(c, t, t2, t3, t4):        # :252-256
    return c[1] + 2. * c[2] * t + 3. * c[3] * t2 + 4. * c[4] * t3 + 5. * c[5] * t4


def _acc(c, t, t2, t3):            # :240-250
    return 2 * c[2] + 6 * c[3] * t + 12 * c[4] * t2 + 20 * c[5] * t3


# ---- reference-path helpers -----------------------------------------------------------------------------------------------
def _valid_orientation(a):
    a = a % TWO_PI
    if math.pi <= a <= TWO_PI:
        a -= TWO_PI
    return a


def _interp_angle(x, x1, x2, y1, y2):   # utils_coordinate_system.py:25-43
    return _valid_orientation((y2 - y1) * (x - x1) / (x2 - x1) + y1)


class _Env:
    """launch-uniform data of one plan call"""

    def __init__(self, params, cost, tables):
        self.p, self.c, self.tb = params, cost, tables
        self.n = params.N + 1
        self.draw = bool(params.flags & 1)
        self.skip_collision = bool(params.flags & 4)
        self.low = bool(params.low_vel_mode)
        self.kappa_max = math.tan(params.delta_max) / params.wheelbase    # reactive_planner.py:985
        from oracle.frontend import compute_vertex_tangents
        self.tan = compute_vertex_tangents(np.stack((tables.ref_x, tables.ref_y), axis=1))
        ob = tables.obstacles
        self.sobb = [(o[0], o[1], math.cos(o[2]), math.sin(o[2]), o[3], o[4]) for o in np.asarray(ob.static_obb).reshape(-1, 5)]
        self.tri = np.asarray(ob.static_tri).reshape(-1, 6)
        self.circ = np.asarray(ob.static_circ).reshape(-1, 3)
        self.dyn = np.asarray(ob.dyn_obb)
        self.dyn_t0 = int(ob.dyn_t0)


def _to_cartesian(env, s, d):
    tb = env.tb
    rp = tb.ref_pos
    n = len(rp)
    if not (rp[0] <= s <= rp[-1] and abs(d) <= tb.proj_d_limit):
        return None
    k = int(np.argmax(rp > s)) - 1
    if k < 0 or k > n - 2:
        k = n - 2
    lam = (s - rp[k]) / (rp[k + 1] - rp[k])
    px = tb.ref_x[k] + lam * (tb.ref_x[k + 1] - tb.ref_x[k])
    py = tb.ref_y[k] + lam * (tb.ref_y[k + 1] - tb.ref_y[k])
    ax = env.tan[k, 0] + lam * (env.tan[k + 1, 0] - env.tan[k, 0])
    ay = env.tan[k, 1] + lam * (env.tan[k + 1, 1] - env.tan[k, 1])
    tn = math.sqrt(ax * ax + ay * ay)
    return px - d * (ay / tn), py + d * (ax / tn)


# ---- collision primitives (closed sets), scalar formulas of oracle/rp_oracle.c ------------------------------------------
def _obb_obb(a, b):
    acx, acy, aux, auy, ahl, ahw = a
    bcx, bcy, bux, buy, bhl, bhw = b
    avx, avy, bvx, bvy = -auy, aux, -buy, bux
    tx, ty = bcx - acx, bcy - acy
    uu, uv = aux * bux + auy * buy, aux * bvx + auy * bvy
    vu, vv = avx * bux + avy * buy, avx * bvx + avy * bvy
    if abs(tx * aux + ty * auy) > ahl + (bhl * abs(uu) + bhw * abs(uv)):
        return False
    if abs(tx * avx + ty * avy) > ahw + (bhl * abs(vu) + bhw * abs(vv)):
        return False
    if abs(tx * bux + ty * buy) > bhl + (ahl * abs(uu) + ahw * abs(vu)):
        return False
    if abs(tx * bvx + ty * bvy) > bhw + (ahl * abs(uv) + ahw * abs(vv)):
        return False
    return True


def _obb_tri(a, t):
    cx, cy, ux, uy, hl, hw = a
    vx, vy = -uy, ux
    lx, ly = [], []
    for k in range(3):
        px, py = t[2 * k] - cx, t[2 * k + 1] - cy
        lx.append(px * ux + py * uy)
        ly.append(px * vx + py * vy)
    if min(lx) > hl or max(lx) < -hl or min(ly) > hw or max(ly) < -hw:
        return False
    for k in range(3):
        k2 = (k + 1) % 3
        nx, ny = -(ly[k2] - ly[k]), lx[k2] - lx[k]
        pr = [lx[q] * nx + ly[q] * ny for q in range(3)]
        r = hl * abs(nx) + hw * abs(ny)
        if min(pr) > r or max(pr) < -r:
            return False
    return True


def _obb_circ(a, c):
    cx, cy, ux, uy, hl, hw = a
    px, py = c[0] - cx, c[1] - cy
    lx, ly = px * ux + py * uy, -px * uy + py * ux
    dx, dy = max(abs(lx) - hl, 0.0), max(abs(ly) - hw, 0.0)
    return dx * dx + dy * dy <= c[2] * c[2]


def _pose_collides(env, ego, t):
    """cc.collide(ego) for one pose at scenario time index t (reactive_planner.py:1040-1042)"""
    for b in env.sobb:
        if _obb_obb(ego, b):
            return True
    for tr in env.tri:
        if _obb_tri(ego, tr):
            return True
    for c in env.circ:
        if _obb_circ(ego, c):
            return True
    k = t - env.dyn_t0
    if env.dyn.size and 0 <= k < env.dyn.shape[1]:
        for o in env.dyn[:, k, :]:
            if o[0] != o[0]:
                continue
            if _obb_obb(ego, (o[0], o[1], math.cos(o[2]), math.sin(o[2]), o[3], o[4])):
                return True
    return False


# ---- one candidate -------------------------------------------------------------------------------------------------------
def _constraints(env, v, kappa, theta, a, i):
    """reactive_planner.py:971-1017: first failing reason or None"""
    p = env.p
    m = p.constraint_mask
    if m & CHECK_VELOCITY and v[i] < -EPS:
        return "velocity"
    if m & CHECK_KAPPA and abs(kappa[i]) > env.kappa_max:
        return "kappa"
    if m & CHECK_YAW_RATE:
        yaw_rate = (theta[i] - theta[i - 1]) / p.dt if i > 0 else 0.
        if abs(np.round(yaw_rate, 5)) > env.kappa_max * v[i]:
            return "yaw_rate"
    if m & CHECK_KAPPA_DOT:
        steering = np.arctan2(p.wheelbase * kappa[i], 1.0)
        kappa_dot_max = p.v_delta_max / (p.wheelbase * math.cos(steering) ** 2)
        kappa_dot = (kappa[i] - kappa[i - 1]) / p.dt if i > 0 else 0.
        if abs(kappa_dot) > kappa_dot_max:
            return "kappa_dot"
    if m & CHECK_ACCELERATION:
        a_max = p.a_max * p.v_switch / v[i] if v[i] > p.v_switch else p.a_max
        if not (-p.a_max <= a[i] <= a_max):
            return "acceleration"
    return None


def _enlarge(st, L, dt):
    """trajectories.py:168-197 (Cartesian) and :302-332 (curvilinear); st = (14, N + 1) with L valid steps"""
    n = st.shape[1]
    last, steps = L - 1, n - L
    t = np.arange(1, steps + 1, 1) * dt
    st[A, L:] = np.repeat(st[A, last], steps)
    v_temp = st[V, last] + t * st[A, -1]
    v_temp = v_temp * np.greater_equal(v_temp, 0)
    st[V, L:] = v_temp
    st[THETA, L:] = np.repeat(st[THETA, last], steps)
    st[KAPPA, L:] = np.repeat(st[KAPPA, last], steps)
    st[KAPPA_DOT, L:] = np.repeat(st[KAPPA_DOT, last], steps)
    st[X, L:] = st[X, last] + np.cumsum(dt * v_temp * math.cos(st[THETA, last]))
    st[Y, L:] = st[Y, last] + np.cumsum(dt * v_temp * math.sin(st[THETA, last]))
    # s_ddot[-1] / d_ddot[-1] are read before the arrays are extended: the zero padding (trajectories.py:312-316)
    s_dot_temp = st[S_DOT, last] + t * st[S_DDOT, -1]
    s_dot_temp = s_dot_temp * np.greater_equal(s_dot_temp, 0)
    d_dot_temp = st[D_DOT, last] + t * st[D_DDOT, -1]
    st[S_DOT, L:] = s_dot_temp
    st[D_DOT, L:] = d_dot_temp
    st[S_DDOT, L:] = np.repeat(st[S_DDOT, last], steps)
    st[D_DDOT, L:] = np.repeat(st[D_DDOT, last], steps)
    st[THETA_CL, L:] = np.repeat(st[THETA_CL, last], steps)
    st[S, L:] = st[S, last] + t * st[S_DOT, last]
    st[D, L:] = st[D, last] + t * st[D_DOT, last]


def _cost(c, st):
    """cost_function.py:51-71 (default), :82-92 (fail-safe)"""
    a, v, s, d, th = st[A], st[V], st[S], st[D], st[THETA_CL]
    if c.kind == 1:
        costs = np.sum((1 * a) ** 2)
        costs += np.sum((0.25 * d) ** 2) + (20 * d[-1]) ** 2
        costs += np.sum((0.25 * np.abs(th)) ** 2) + (5 * (np.abs(th[-1]))) ** 2
        return float(costs)
    costs = 0.0
    costs += np.sum((c.w_a * a) ** 2)
    if not math.isnan(c.desired_speed):
        vd = c.desired_speed
        costs += np.sum((5 * (v - vd)) ** 2) + (50 * (v[-1] - vd) ** 2) + (100 * (v[int(len(v) / 2)] - vd) ** 2)
    if not math.isnan(c.desired_s):
        costs += np.sum((0.25 * (c.desired_s - s)) ** 2) + (20 * (c.desired_s - s[-1])) ** 2
    costs += np.sum((0.25 * (c.desired_d - d)) ** 2) + (20 * (c.desired_d - d[-1])) ** 2
    costs += np.sum((0.25 * np.abs(th)) ** 2) + (5 * (np.abs(th[-1]))) ** 2
    return float(costs)


def evaluate(env, lon, lat, traj_len):
    """One iteration of the loop body of _check_kinematics (reactive_planner.py:731-960).
    Returns (status word, cost or NaN, state block or None)."""
    p = env.p
    n, dt = env.n, p.dt
    traj_len = min(int(traj_len), n)
    t = np.arange(traj_len) * dt                                  # :733
    t2 = np.square(t)
    t3 = t2 * t
    t4 = np.square(t2)
    t5 = t4 * t
    st = np.zeros((14, n))
    s, s_vel, s_acc, d, d_vel, d_acc = st[S], st[S_DOT], st[S_DDOT], st[D], st[D_DOT], st[D_DDOT]
    s[:traj_len] = _pos(lon, t, t2, t3, t4, t5)
    s_vel[:traj_len] = _vel(lon, t, t2, t3, t4)
    s_acc[:traj_len] = _acc(lon, t, t2, t3)
    if not env.low:
        d[:traj_len] = _pos(lat, t, t2, t3, t4, t5)
        d_vel[:traj_len] = _vel(lat, t, t2, t3, t4)
        d_acc[:traj_len] = _acc(lat, t, t2, t3)
    else:                                                         # :759-772
        s1 = s[:traj_len] - s[0]
        s2 = np.square(s1)
        s3 = s2 * s1
        s4 = np.square(s2)
        s5 = s4 * s1
        d[:traj_len] = _pos(lat, s1, s2, s3, s4, s5)
        d_vel[:traj_len] = _vel(lat, s1, s2, s3, s4)
        d_acc[:traj_len] = _acc(lat, s1, s2, s3)
    s_vel[np.abs(s_vel) < EPS] = 0.0                              # :776-777
    d_vel[np.abs(d_vel) < EPS] = 0.0
    if not env.draw:                                              # :796-805
        if np.any(np.abs(s_acc) > p.a_max):
            return LABEL_NONE | (REASON["acceleration"] << 4), math.nan, None
        if np.any(s_vel < -EPS):
            return LABEL_NONE | (REASON["velocity"] << 4), math.nan, None

    x, y, theta_gl, v, a = st[X], st[Y], st[THETA], st[V], st[A]
    theta_cl, kappa_gl = st[THETA_CL], st[KAPPA]
    tb = env.tb
    ref_pos, ref_theta, ref_curv, ref_curv_d = tb.ref_pos, tb.ref_theta, tb.ref_curv, tb.ref_curv_d
    feasible, reason, bad_step = True, "none", 0
    for i in range(traj_len):                                     # :807-904
        if not env.low:
            if s_vel[i] > 0.001:
                dp = d_vel[i] / s_vel[i]
            else:
                dp = 0.
            ddot = d_acc[i] - dp * s_acc[i]
            if s_vel[i] > 0.001:
                dpp = ddot / (s_vel[i] ** 2)
            else:
                dpp = 0.
        else:
            dp, dpp = d_vel[i], d_acc[i]
        s_idx = np.argmax(ref_pos > s[i]) - 1                     # :835 (a negative index wraps, as in the reference)
        s_lambda = (s[i] - ref_pos[s_idx]) / (ref_pos[s_idx + 1] - ref_pos[s_idx])
        if s_vel[i] > 0.001 or env.low:
            theta_cl[i] = np.arctan2(dp, 1.0)
            theta_gl[i] = theta_cl[i] + _interp_angle(s[i], ref_pos[s_idx], ref_pos[s_idx + 1], ref_theta[s_idx],
                                                      ref_theta[s_idx + 1])
        else:
            theta_gl[i] = p.x0_orientation if i == 0 else theta_gl[i - 1]
            theta_cl[i] = theta_gl[i] - _interp_angle(s[i], ref_pos[s_idx], ref_pos[s_idx + 1], ref_theta[s_idx],
                                                      ref_theta[s_idx + 1])
        k_r = (ref_curv[s_idx + 1] - ref_curv[s_idx]) * s_lambda + ref_curv[s_idx]
        k_r_d = (ref_curv_d[s_idx + 1] - ref_curv_d[s_idx]) * s_lambda + ref_curv_d[s_idx]
        one_krd = 1 - k_r * d[i]
        cos_t = math.cos(theta_cl[i])
        tan_t = np.tan(theta_cl[i])
        kappa_gl[i] = (dpp + (k_r * dp + k_r_d * d[i]) * tan_t) * cos_t * (cos_t / one_krd) ** 2 + (cos_t / one_krd) * k_r
        v[i] = s_vel[i] * (one_krd / cos_t)
        a[i] = s_acc[i] * one_krd / cos_t + ((s_vel[i] ** 2) / cos_t) * (
            one_krd * tan_t * (kappa_gl[i] * one_krd / cos_t - k_r) - (k_r_d * d[i] + k_r * dp))
        if feasible:
            r = _constraints(env, v, kappa_gl, theta_gl, a, i)
            if r is not None:
                feasible, reason, bad_step = False, r, i
        if not feasible and not env.draw:
            break
    if not feasible and not env.draw:
        return LABEL_KINEMATIC | (REASON[reason] << 4) | (bad_step << 8), math.nan, None

    for i in range(traj_len):                                     # :907-917
        pos = _to_cartesian(env, s[i], d[i])
        if pos is None:
            if feasible:
                reason, bad_step = "domain", i
            feasible = False
            break
        x[i], y[i] = pos
    if not feasible and not env.draw:
        return LABEL_NONE | (REASON[reason] << 4) | (bad_step << 8), math.nan, None

    st[KAPPA_DOT, 1:] = np.diff(kappa_gl)                         # :919-921 (over the padded array)
    if n > traj_len:
        _enlarge(st, traj_len, dt)
    if not feasible:
        return LABEL_KINEMATIC | (REASON[reason] << 4) | (bad_step << 8), math.nan, st
    return LABEL_FEASIBLE, _cost(env.c, st), st


def _collision_step(env, st):
    """reactive_planner.py:1031-1046 for one trajectory: first colliding step or -1"""
    p = env.p
    hl, hw = 0.5 * p.length, 0.5 * p.width
    pos1 = st[X] + p.wb_rear_axle * np.cos(st[THETA])
    pos2 = st[Y] + p.wb_rear_axle * np.sin(st[THETA])
    theta = st[THETA]
    for i in range(len(pos1)):
        ego = (pos1[i], pos2[i], math.cos(theta[i]), math.sin(theta[i]), hl, hw)
        if _pose_collides(env, ego, p.time_step0 + i * p.factor):
            return i
    return -1


def sample(params, grids, idx):
    """sampling.py:202-242 for candidate ``idx`` of the (T, L, D) triple loop -> (lon, lat, traj_len); ``grids`` has the
    arrays T, traj_len, L, D (a PlanInputs)"""
    nL, nD = len(grids.L), len(grids.D)
    iT, rem = divmod(int(idx), nL * nD)
    iL, iD = divmod(rem, nD)
    T = float(grids.T[iT])
    x0_lon, x0_lat = list(params.x0_lon), list(params.x0_lat)
    if params.lon_mode == 1:
        lon = quintic(x0_lon, (grids.L[iL], 0.0, 0.0), T)
    else:
        lon = quartic(x0_lon, grids.L[iL], T)
    tau = T
    if params.low_vel_mode:
        t2 = T * T
        t3, t4 = t2 * T, t2 * t2
        s_goal = _pos(lon, T, t2, t3, t4, t3 * t2) - x0_lon[0]
        tau = s_goal if s_goal > 0 else T
    lat = quintic(x0_lat, (grids.D[iD], 0.0, 0.0), tau)
    return lon, lat, int(grids.traj_len[iT])


def plan(inp, tables, cand_begin: int = 0, cand_end: int = -1, eager_collision: bool = False):
    """One level of plan() over the candidates [cand_begin, cand_end): the reference's sequence -- create the samples,
    evaluate them one by one, sort by cost, walk the sorted list until the first collision-free one.

    Returns dict(best_index, best_cost, status (uint32 per candidate), cost, states {index: block}, n_collision_checked).
    ``eager_collision``: also label every costed candidate the way the device does (for parity tests)."""
    params, grids = inp.params, inp
    end = inp.n_candidates if cand_end < 0 else cand_end
    env = _Env(params, inp.cost, tables)
    cnt = end - cand_begin
    status = np.zeros(cnt, dtype=np.uint32)
    cost = np.full(cnt, np.nan)
    kept = {}
    for k in range(cnt):
        lon, lat, tl = sample(params, grids, cand_begin + k)
        status[k], cost[k], st = evaluate(env, lon, lat, tl)
        if st is not None:
            kept[k] = st
    feasible = [k for k in range(cnt) if (status[k] & 0xF) == LABEL_FEASIBLE]
    feasible.sort(key=lambda k: cost[k])                           # trajectories.py:502-510 (stable: ties keep list order)
    best, checked = -1, 0
    if env.skip_collision:
        best = feasible[0] if feasible else -1
    else:
        for k in feasible:
            checked += 1
            step = _collision_step(env, kept[k])
            if step >= 0:
                status[k] = LABEL_COLLISION | (step << 8)
            else:
                best = k
                break
        if eager_collision:
            for k in feasible:
                if (status[k] & 0xF) == LABEL_FEASIBLE and k != best:
                    step = _collision_step(env, kept[k])
                    if step >= 0:
                        status[k] = LABEL_COLLISION | (step << 8)
    return {"best_index": cand_begin + best if best >= 0 else -1, "best_cost": float(cost[best]) if best >= 0 else math.nan,
            "status": status, "cost": cost, "states": kept, "n_collision_checked": checked}
