"""CPU restatement of the reference-path front end (TEST INFRASTRUCTURE: imported by tests/ only, never by the product).

What it restates (paths relative to /root/reference), line by line in NumPy, with SciPy's ``splprep`` / ``splev`` called
exactly as the reference calls them (SciPy is the reference's own dependency, poetry.lock scipy 1.10.1; the product
re-implements the spline in C++, csrc/rp_frontend.h):

  CoordinateSystem.__init__                 commonroad_rp/utility/utils_coordinate_system.py:88-118
  smooth_ref_path                           commonroad_rp/utility/utils_coordinate_system.py:74-83
  ReactivePlanner._compute_initial_states   commonroad_rp/reactive_planner.py:446-512

Parity status: PINNED for the in-tree arithmetic by tests/golden/frontend_*.npz, which tests/golden/make_frontend_golden.py
produces by running the reference's own ``CoordinateSystem.__init__`` and ``_compute_initial_states`` (unmodified, imported
from /root/reference) under stand-ins for commonroad-drivability-checker.  UNPINNED at that third-party boundary
(``commonroad_dc.geometry.util.{compute_pathlength,compute_orientation,compute_curvature}_from_polyline``,
``resample_polyline``, ``pycrccosy.CurvilinearCoordinateSystem``: source not under /root/reference): the definitions below
are this build's own and are what the stand-ins hand to the reference.
"""
from __future__ import annotations

import math

import numpy as np

TWO_PI = 2.0 * np.pi


# ---- this build's definitions of the commonroad_dc.geometry.util functions the reference calls -------------------------
def compute_pathlength_from_polyline(polyline: np.ndarray) -> np.ndarray:
    """cumulative arc length, first entry 0 (used at utils_coordinate_system.py:114)"""
    polyline = np.asarray(polyline, dtype=np.float64)
    seg = np.sqrt(np.sum(np.diff(polyline, axis=0) ** 2, axis=1))
    return np.concatenate(([0.0], np.cumsum(seg)))


def compute_orientation_from_polyline(polyline: np.ndarray) -> np.ndarray:
    """segment headings; the last vertex repeats the last segment's heading (used at :116)"""
    polyline = np.asarray(polyline, dtype=np.float64)
    d = np.diff(polyline, axis=0)
    theta = np.arctan2(d[:, 1], d[:, 0])
    return np.concatenate((theta, theta[-1:]))


def compute_curvature_from_polyline(polyline: np.ndarray) -> np.ndarray:
    """signed curvature from second-order finite differences over arc length (used at :115)"""
    polyline = np.asarray(polyline, dtype=np.float64)
    s = compute_pathlength_from_polyline(polyline)
    x_d = np.gradient(polyline[:, 0], s)
    x_dd = np.gradient(x_d, s)
    y_d = np.gradient(polyline[:, 1], s)
    y_dd = np.gradient(y_d, s)
    return (x_d * y_dd - x_dd * y_d) / ((x_d ** 2 + y_d ** 2) ** 1.5)


def resample_polyline(polyline: np.ndarray, step: float = 1.0) -> np.ndarray:
    """equidistant resampling by linear interpolation over arc length (used at :82); the end point is kept when it lies
    more than 1e-9 m beyond the last multiple of ``step``"""
    polyline = np.asarray(polyline, dtype=np.float64)
    s = compute_pathlength_from_polyline(polyline)
    n = max(int(np.floor(s[-1] / step)) + 1, 2)
    s_new = np.arange(n) * step
    if s[-1] - s_new[-1] > 1e-9:
        s_new = np.append(s_new, s[-1])
    return np.stack((np.interp(s_new, s, polyline[:, 0]), np.interp(s_new, s, polyline[:, 1])), axis=1)


def compute_vertex_tangents(polyline: np.ndarray) -> np.ndarray:
    """unit tangent per vertex: normalised sum of the two adjacent unit segment directions"""
    p = np.asarray(polyline, dtype=np.float64)
    e = np.diff(p, axis=0)
    u = e / np.sqrt(e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1])[:, None]
    t = np.empty_like(p)
    t[0], t[-1] = u[0], u[-1]
    if len(p) > 2:
        ts = u[:-1] + u[1:]
        t[1:-1] = ts / np.sqrt(ts[:, 0] * ts[:, 0] + ts[:, 1] * ts[:, 1])[:, None]
    return t


def make_valid_orientation(angle: float) -> float:
    """commonroad-io 2024.1 ``make_valid_orientation`` (used by interpolate_angle, :43): into [-pi, pi)"""
    angle = angle % TWO_PI
    if np.pi <= angle <= TWO_PI:
        angle = angle - TWO_PI
    return angle


# ---- (x, y) <-> (s, d): this build's definition of what pycrccosy does ------------------------------------------------
def project(reference: np.ndarray, ref_pos: np.ndarray, x: float, y: float, d_limit: float = 20.0):
    """(s, d) of a point, or None outside the projection domain: per segment the foot point solves
    ((P - p0) - lam e) . ((1 - lam) t0 + lam t1) = 0; the admissible root with the smallest |d| wins."""
    ref = np.asarray(reference, dtype=np.float64)
    tan = compute_vertex_tangents(ref)
    P = np.array([x, y], dtype=np.float64)
    best = None
    for k in range(len(ref) - 1):
        e, q = ref[k + 1] - ref[k], P - ref[k]
        t0, dt = tan[k], tan[k + 1] - tan[k]
        a, b, c = -float(e @ dt), float(q @ dt) - float(e @ t0), float(q @ t0)
        if abs(a) < 1e-14:
            roots = [-c / b] if b != 0.0 else []
        else:
            disc = b * b - 4.0 * a * c
            if disc < 0.0:
                continue
            sq = math.sqrt(disc)
            # the two roots (-b + sq) / 2a, (-b - sq) / 2a, in that order, WITHOUT the cancellation of the textbook form: on a nearly
            # straight stretch a is tiny, the admissible root is -c / b (1 + O(a)), and (-b -+ sq) loses all but a few digits of it
            # (round 5: 5e-5 m in s on random polylines, live sweep against the C++ side, whose fma rounds the difference another way)
            q = -0.5 * (b + sq) if b >= 0.0 else -0.5 * (b - sq)
            if q == 0.0:
                roots = [0.0, 0.0]
            elif b >= 0.0:
                roots = [c / q, q / a]
            else:
                roots = [q / a, c / q]
        for lam in roots:
            if -1e-12 <= lam <= 1.0 + 1e-12:
                lam = min(max(lam, 0.0), 1.0)
                t = t0 + lam * dt
                foot = ref[k] + lam * e
                dd = (-(P[0] - foot[0]) * t[1] + (P[1] - foot[1]) * t[0]) / math.sqrt(t[0] * t[0] + t[1] * t[1])
                if abs(dd) <= d_limit and (best is None or abs(dd) < abs(best[1])):
                    best = (ref_pos[k] + lam * (ref_pos[k + 1] - ref_pos[k]), dd)
    return best


# ---- the reference's own lines -----------------------------------------------------------------------------------------
def unique_vertices(reference: np.ndarray) -> np.ndarray:
    """utils_coordinate_system.py:95-96 / :103-104"""
    _, idx = np.unique(reference, axis=0, return_index=True)
    return reference[np.sort(idx)]


def smooth_ref_path(ref_path: np.ndarray, smoothing_factor: float = 0.0, resample_step: float = 1.0) -> np.ndarray:
    """utils_coordinate_system.py:74-83"""
    from scipy.interpolate import splev, splprep
    tck, u = splprep(ref_path.T, u=None, k=3, s=smoothing_factor)        # :78
    u_new = np.linspace(u.min(), u.max(), 200)                           # :79
    x_new, y_new = splev(u_new, tck, der=0)                              # :80
    return resample_polyline(np.array([x_new, y_new]).transpose(), resample_step)   # :81-82


def build_reference(reference: np.ndarray, smooth: bool = True, resample_step: float = 1.0):
    """CoordinateSystem.__init__, utils_coordinate_system.py:88-118 -> (reference, ref_pos, ref_theta, ref_curv, ref_curv_d).
    (The CCosy object's ``reference_path()``, :128-129, is the identity in this build's definition.)"""
    reference = unique_vertices(np.asarray(reference, dtype=np.float64))                  # :95-96
    if smooth:
        reference = unique_vertices(smooth_ref_path(reference, resample_step=resample_step))   # :100-104
    ref_pos = compute_pathlength_from_polyline(reference)                                 # :114
    ref_curv = compute_curvature_from_polyline(reference)                                 # :115
    ref_theta = np.unwrap(compute_orientation_from_polyline(reference))                   # :116
    ref_curv_d = np.gradient(ref_curv, ref_pos)                                           # :117
    return reference, ref_pos, ref_theta, ref_curv, ref_curv_d


def initial_state(tables, x, y, orientation, velocity, acceleration=0.0, steering_angle=0.0, wheelbase=1.0,
                  low_vel_mode=False, d_limit=20.0):
    """ReactivePlanner._compute_initial_states, reactive_planner.py:446-512.  ``tables`` = build_reference's tuple.
    Returns (x_0_lon, x_0_lat); raises ValueError outside the projection domain (:459-461) and Exception for a negative
    longitudinal velocity (:492-494)."""
    reference, ref_pos, ref_theta, ref_curv, ref_curv_d = tables
    sd = project(reference, ref_pos, x, y, d_limit)                                       # :458
    if sd is None:
        raise ValueError("Initial state could not be transformed.")
    s, d = sd
    s_idx = int(np.argmax(ref_pos > s)) - 1                                               # :464
    s_lambda = (s - ref_pos[s_idx]) / (ref_pos[s_idx + 1] - ref_pos[s_idx])               # :465-466
    th = np.unwrap(ref_theta)                                                             # :469
    delta = th[s_idx + 1] - th[s_idx]                                                     # interpolate_angle, :37-43
    theta_cl = orientation - make_valid_orientation(delta * (s - ref_pos[s_idx]) / (ref_pos[s_idx + 1] - ref_pos[s_idx])
                                                    + th[s_idx])                          # :470-471
    kr = (ref_curv[s_idx + 1] - ref_curv[s_idx]) * s_lambda + ref_curv[s_idx]             # :474-475
    kr_d = (ref_curv_d[s_idx + 1] - ref_curv_d[s_idx]) * s_lambda + ref_curv_d[s_idx]     # :477
    kappa_0 = np.tan(steering_angle) / wheelbase                                          # :480
    d_p = (1 - kr * d) * np.tan(theta_cl)                                                 # :483
    d_pp = -(kr_d * d + kr * d_p) * np.tan(theta_cl) + ((1 - kr * d) / (math.cos(theta_cl) ** 2)) * (
        kappa_0 * (1 - kr * d) / math.cos(theta_cl) - kr)                                 # :484-485
    s_velocity = velocity * math.cos(theta_cl) / (1 - kr * d)                             # :488
    if s_velocity < 0:                                                                    # :489-491
        raise Exception("negative longitudinal velocity")
    s_acceleration = acceleration                                                         # :493
    s_acceleration -= (s_velocity ** 2 / math.cos(theta_cl)) * (
        (1 - kr * d) * np.tan(theta_cl) * (kappa_0 * (1 - kr * d) / (math.cos(theta_cl)) - kr) - (kr_d * d + kr * d_p))   # :494-496
    s_acceleration /= ((1 - kr * d) / (math.cos(theta_cl)))                               # :497
    if low_vel_mode:                                                                      # :500-503
        d_velocity, d_acceleration = d_p, d_pp
    else:                                                                                 # :504-507
        d_velocity = velocity * math.sin(theta_cl)
        d_acceleration = s_acceleration * d_p + s_velocity ** 2 * d_pp
    return [float(s), float(s_velocity), float(s_acceleration)], [float(d), float(d_velocity), float(d_acceleration)]
