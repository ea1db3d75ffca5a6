"""Reference-path tables and the (s, d) <-> (x, y) transform used by the planner hot path.

Mirrors the interface of ``commonroad_rp.utility.utils_coordinate_system.CoordinateSystem``
(reference: commonroad_rp/utility/utils_coordinate_system.py:86-178): the attributes
``reference, ref_pos, ref_curv, ref_theta, ref_curv_d`` and the two ``convert_to_*`` methods.
The work behind it is native: ``rp_build_reference`` (duplicate removal, cubic-spline smoothing,
resampling, tables) and ``rp_project`` ((x, y) -> (s, d)) of the C ABI (include/rp_amd.h,
csrc/rp_frontend.h; host-only entry points, no GPU needed).

The reference delegates the geometry to the un-vendored C++ package commonroad-drivability-checker
2024.1 (``pycrccosy.CurvilinearCoordinateSystem`` and ``commonroad_dc.geometry.util``), whose source
is not under /root/reference.  This build therefore *defines* the transform explicitly
(polyline segment + linearly interpolated vertex tangent, normal = tangent rotated by +90 deg,
normalised) -- the same definition is implemented by ``oracle/`` and by the HIP kernels
(``csrc/rp_device.h``), and it is what the golden fixtures were generated with.
Parity with the real CCosy is unpinned.

This is set-up time code (once per reference path); it is not on the per-replan hot path.
"""
from __future__ import annotations

import numpy as np

from . import _capi

__all__ = ["compute_vertex_tangents", "interpolate_angle", "make_valid_orientation", "CoordinateSystem"]

TWO_PI = 2.0 * np.pi


def make_valid_orientation(angle: float) -> float:
    """Wrap an angle into [-pi, pi) the way commonroad-io's ``make_valid_orientation`` does
    (third-party commonroad-io 2024.1, used at utils_coordinate_system.py:43; source not under
    /root/reference).  Restated from its published behaviour: ``angle % 2pi`` (Python float
    modulo, result in [0, 2pi]) and ``- 2pi`` when the result lies in [pi, 2pi]."""
    angle = angle % TWO_PI
    if np.pi <= angle <= TWO_PI:
        angle = angle - TWO_PI
    return angle


def interpolate_angle(x: float, x1: float, x2: float, y1: float, y2: float) -> float:
    """utils_coordinate_system.py:25-43 -- plain linear interpolation then wrap."""
    delta = y2 - y1
    return make_valid_orientation(delta * (x - x1) / (x2 - x1) + y1)


def compute_vertex_tangents(polyline: np.ndarray) -> np.ndarray:
    """Unit tangent per vertex: normalised sum of the two adjacent unit segment directions
    (end vertices: the single adjacent segment).  The operation order here is the contract that
    ``rp_vertex_tangents`` in oracle/rp_oracle.c and csrc/rp_host.hip reproduce bit for bit:
    e = p[i+1]-p[i]; u = e / sqrt(ex*ex + ey*ey); t = u[i-1] + u[i]; t / sqrt(tx*tx + ty*ty)."""
    p = np.asarray(polyline, dtype=np.float64)
    e = np.diff(p, axis=0)
    ln = np.sqrt(e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1])
    u = e / ln[:, None]
    t = np.empty_like(p)
    t[0] = u[0]
    t[-1] = u[-1]
    if len(p) > 2:
        ts = u[:-1] + u[1:]
        tn = np.sqrt(ts[:, 0] * ts[:, 0] + ts[:, 1] * ts[:, 1])
        t[1:-1] = ts / tn[:, None]
    return t


class CoordinateSystem:
    """Drop-in for the reference ``CoordinateSystem`` on the hot path's side of the boundary.

    :param reference: (n, 2) polyline; consecutive duplicates are removed.
    :param proj_domain_d_limit: |d| beyond which ``convert_to_cartesian_coords`` returns ``None``
        (CCosy's ``default_projection_domain_limit``; 20 m there).
    """

    def __init__(self, reference: np.ndarray, proj_domain_d_limit: float = 20.0, smooth_reference: bool = False,
                 resample_step: float = 1.0):
        reference = np.asarray(reference, dtype=np.float64)
        assert reference.ndim == 2 and reference.shape[1] == 2 and reference.shape[0] >= 2
        self.proj_domain_d_limit = float(proj_domain_d_limit)
        # utils_coordinate_system.py:88-118 in one native call: duplicates removed, smoothed and resampled when asked
        # (the reference smooths by default; here the caller says so: the fixtures and workloads hand in final polylines)
        (self._reference, self._ref_pos, self._ref_theta, self._ref_curv, self._ref_curv_d) = _capi.build_reference(
            reference, smooth=smooth_reference, resample_step=resample_step)
        self._tangent = compute_vertex_tangents(self._reference)

    # --- reference-compatible read-only views ------------------------------------------------
    @property
    def reference(self) -> np.ndarray:
        return self._reference

    @property
    def ref_pos(self) -> np.ndarray:
        return self._ref_pos

    @property
    def ref_curv(self) -> np.ndarray:
        return self._ref_curv

    @property
    def ref_curv_d(self) -> np.ndarray:
        return self._ref_curv_d

    @property
    def ref_theta(self) -> np.ndarray:
        return self._ref_theta

    @property
    def vertex_tangents(self) -> np.ndarray:
        return self._tangent

    # --- transforms --------------------------------------------------------------------------
    def convert_to_cartesian_coords(self, s: float, d: float):
        """(s, d) -> np.array([x, y]) or ``None`` outside the projection domain
        (reference: utils_coordinate_system.py:167-174 swallows CCosy's exception into None)."""
        rp = self._ref_pos
        n = len(rp)
        if not (s >= rp[0] and s <= rp[n - 1] and abs(d) <= self.proj_domain_d_limit):
            return None
        # k = last vertex with ref_pos[k] <= s, clamped to a valid segment
        k = int(np.searchsorted(rp, s, side="right")) - 1
        if k > n - 2:
            k = n - 2
        lam = (s - rp[k]) / (rp[k + 1] - rp[k])
        p0 = self._reference[k]
        p1 = self._reference[k + 1]
        t0 = self._tangent[k]
        t1 = self._tangent[k + 1]
        px = p0[0] + lam * (p1[0] - p0[0])
        py = p0[1] + lam * (p1[1] - p0[1])
        tx = t0[0] + lam * (t1[0] - t0[0])
        ty = t0[1] + lam * (t1[1] - t0[1])
        tn = np.sqrt(tx * tx + ty * ty)
        return np.array([px - d * (ty / tn), py + d * (tx / tn)])

    def convert_to_curvilinear_coords(self, x: float, y: float):
        """(x, y) -> (s, d); raises ValueError outside the projection domain
        (reference: utils_coordinate_system.py:176-178; used once per replan for the initial
        state at reactive_planner.py:458).  ``rp_project``: per segment the foot point solves
        ((P - p0) - lam*e) . ((1-lam) t0 + lam t1) = 0, a quadratic in lam."""
        return _capi.project(self._reference, self._ref_pos, x, y, self.proj_domain_d_limit)

    def initial_state(self, x: float, y: float, orientation: float, velocity: float, acceleration: float = 0.0,
                      steering_angle: float = 0.0, wheelbase: float = 1.0, low_vel_mode: bool = False):
        """Frenet state (x_0_lon, x_0_lat) of a Cartesian rear-axle state: ``rp_initial_state``, i.e.
        ``ReactivePlanner._compute_initial_states`` (reactive_planner.py:446-512)."""
        return _capi.initial_state(self._reference, self._ref_pos, self._ref_theta, self._ref_curv, self._ref_curv_d, x, y,
                                   orientation, velocity, acceleration, steering_angle, wheelbase, low_vel_mode,
                                   self.proj_domain_d_limit)
