"""ctypes wrapper of the CPU oracle ``librp_oracle.so`` (built from oracle/rp_oracle.c).

TEST INFRASTRUCTURE: only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg
may import this module.  It reuses the POD struct definitions of the product binding
(``commonroad_rp_amd._capi``) so both sides are fed identical bytes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
_PKG = os.path.join(REPO, "commonroad-reactive-planner_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from commonroad_rp_amd._capi import (RpParams, RpCost, RpGrids, RpResult, PlanInputs, PlanOutput, N_ARRAYS, f64,  # noqa: E402
                                     dptr)
from commonroad_rp_amd.collision import ObstacleTables  # noqa: E402

LIB = os.path.join(HERE, "librp_oracle.so")


class RpoTables(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("n_sobb", C.c_int32), ("n_tri", C.c_int32), ("n_circ", C.c_int32),
                ("n_dyn", C.c_int32), ("n_steps", C.c_int32), ("dyn_t0", C.c_int32), ("reserved_", C.c_int32),
                ("proj_d_limit", C.c_double),
                ("ref_pos", C.POINTER(C.c_double)), ("ref_theta", C.POINTER(C.c_double)),
                ("ref_curv", C.POINTER(C.c_double)), ("ref_curv_d", C.POINTER(C.c_double)),
                ("ref_x", C.POINTER(C.c_double)), ("ref_y", C.POINTER(C.c_double)),
                ("sobb", C.POINTER(C.c_double)), ("tri", C.POINTER(C.c_double)), ("circ", C.POINTER(C.c_double)),
                ("dyn", C.POINTER(C.c_double))]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "rp_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
        _lib.rp_oracle_plan.restype = C.c_int
        _lib.rp_oracle_plan.argtypes = [C.POINTER(RpParams), C.POINTER(RpCost), C.POINTER(RpGrids),
                                        C.POINTER(RpoTables), C.c_int64, C.c_int64, up, dp, dp, dp,
                                        C.POINTER(RpResult), dp, C.c_int]
        _lib.rp_oracle_plan_coeffs.restype = C.c_int
        _lib.rp_oracle_plan_coeffs.argtypes = [C.POINTER(RpParams), C.POINTER(RpCost), C.POINTER(RpoTables), C.c_int64,
                                               dp, dp, ip, up, dp, dp, C.POINTER(RpResult), dp, C.c_int]
        _lib.rp_oracle_count_collisions_before.restype = C.c_int64
        _lib.rp_oracle_count_collisions_before.argtypes = [C.c_int64, C.c_int64, up, dp, C.c_double, C.c_int64]
        _lib.rp_oracle_check_swept.restype = C.c_int
        _lib.rp_oracle_check_swept.argtypes = [C.POINTER(RpParams), C.POINTER(RpoTables), C.c_int, dp, dp, dp, dp]
        _lib.rp_oracle_obb_sum_rows.restype = None
        _lib.rp_oracle_obb_sum_rows.argtypes = [C.c_int, dp, dp]
        _lib.rpo_np_sum.restype = C.c_double
        _lib.rpo_np_sum.argtypes = [dp, C.c_long]
        _lib.rpo_quintic_coeffs.restype = None
        _lib.rpo_quintic_coeffs.argtypes = [C.c_double] * 7 + [dp]
        _lib.rpo_quartic_coeffs.restype = None
        _lib.rpo_quartic_coeffs.argtypes = [C.c_double] * 5 + [dp]
    return _lib


class OracleTables:
    """Reference-path + obstacle tables in the layout the oracle reads (host memory)."""

    def __init__(self, ref_pos, ref_theta, ref_curv, ref_curv_d, ref_xy, proj_d_limit=20.0, obstacles=None):
        self.ref_pos, self.ref_theta = f64(ref_pos), f64(ref_theta)
        self.ref_curv, self.ref_curv_d = f64(ref_curv), f64(ref_curv_d)
        xy = f64(ref_xy)
        self.ref_x, self.ref_y = f64(xy[:, 0]), f64(xy[:, 1])
        self.proj_d_limit = float(proj_d_limit)
        self.obstacles = obstacles if obstacles is not None else ObstacleTables()

    @classmethod
    def from_coordinate_system(cls, co, obstacles=None):
        return cls(co.ref_pos, co.ref_theta, co.ref_curv, co.ref_curv_d, co.reference,
                   getattr(co, "proj_domain_d_limit", 20.0), obstacles)

    def c_struct(self) -> RpoTables:
        ob = self.obstacles
        return RpoTables(len(self.ref_pos), len(ob.static_obb), len(ob.static_tri), len(ob.static_circ),
                         ob.dyn_obb.shape[0], ob.dyn_obb.shape[1], ob.dyn_t0, 0, self.proj_d_limit,
                         dptr(self.ref_pos), dptr(self.ref_theta), dptr(self.ref_curv), dptr(self.ref_curv_d),
                         dptr(self.ref_x), dptr(self.ref_y), dptr(ob.static_obb), dptr(ob.static_tri),
                         dptr(ob.static_circ), dptr(ob.dyn_obb))


class OracleRun:
    def __init__(self, out: PlanOutput, status, cost, coeffs, states):
        self.out, self.status, self.cost, self.coeffs, self.states = out, status, cost, coeffs, states


def plan(inp: PlanInputs, tables: OracleTables, cand_begin: int = 0, cand_end: int = -1, want_states: bool = True,
         nthreads: int = 1, scratch: dict = None) -> OracleRun:
    """``scratch``: a dict that keeps the output arrays between calls of the same size (timing loops: fresh arrays of hundreds
    of MB are page-faulted in by every call; rows the oracle does not write then keep what an earlier call left)."""
    total = inp.n_candidates
    end = total if cand_end < 0 else cand_end
    cnt = end - cand_begin
    n = inp.params.N + 1
    key = (cnt, n, bool(want_states))
    if scratch is not None and scratch.get("key") == key:
        status, cost, coeffs, states = scratch["arrays"]
        status[:] = 0
        cost[:] = np.nan
    else:
        status = np.zeros(cnt, dtype=np.uint32)
        cost = np.full(cnt, np.nan)
        coeffs = np.zeros((cnt, 13))
        states = np.zeros((cnt, N_ARRAYS, n)) if want_states else None
        if scratch is not None:
            scratch["key"], scratch["arrays"] = key, (status, cost, coeffs, states)
    best = np.zeros((N_ARRAYS, n))
    res = RpResult()
    tb = tables.c_struct()
    g = inp.grids()
    rc = lib().rp_oracle_plan(C.byref(inp.params), C.byref(inp.cost), C.byref(g), C.byref(tb), cand_begin, end,
                              status.ctypes.data_as(C.POINTER(C.c_uint32)), dptr(cost), dptr(coeffs), dptr(states),
                              C.byref(res), dptr(best), nthreads)
    if rc != 0:
        raise RuntimeError(f"rp_oracle_plan -> {rc}")
    return OracleRun(PlanOutput.from_c(res, best), status, cost, coeffs, states)


def plan_coeffs(params, cost_p, tables: OracleTables, lon_coeffs, lat_coeffs, traj_len, want_states=True, nthreads=1):
    lon_coeffs, lat_coeffs = f64(lon_coeffs), f64(lat_coeffs)
    traj_len = np.ascontiguousarray(traj_len, dtype=np.int32)
    cnt = len(traj_len)
    n = params.N + 1
    status = np.zeros(cnt, dtype=np.uint32)
    cost = np.full(cnt, np.nan)
    states = np.zeros((cnt, N_ARRAYS, n)) if want_states else None
    best = np.zeros((N_ARRAYS, n))
    res = RpResult()
    tb = tables.c_struct()
    rc = lib().rp_oracle_plan_coeffs(C.byref(params), C.byref(cost_p), C.byref(tb), cnt, dptr(lon_coeffs),
                                     dptr(lat_coeffs), traj_len.ctypes.data_as(C.POINTER(C.c_int32)),
                                     status.ctypes.data_as(C.POINTER(C.c_uint32)), dptr(cost), dptr(states),
                                     C.byref(res), dptr(best), nthreads)
    if rc != 0:
        raise RuntimeError(f"rp_oracle_plan_coeffs -> {rc}")
    return OracleRun(PlanOutput.from_c(res, best), status, cost, None, states)


def count_collisions_before(status, cost, base, wcost, windex) -> int:
    status = np.ascontiguousarray(status, dtype=np.uint32)
    cost = f64(cost)
    return int(lib().rp_oracle_count_collisions_before(len(status), base, status.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                       dptr(cost), float(wcost), int(windex)))


def check_swept(params, tables: OracleTables, x, y, theta, want_boxes: bool = False):
    """Continuous collision check of one trajectory: (first colliding segment or -1, boxes [n-1][6] or None)."""
    x, y, theta = f64(x), f64(y), f64(theta)
    n = len(x)
    boxes = np.zeros((max(n - 1, 0), 6)) if want_boxes else None
    tb = tables.c_struct()
    first = lib().rp_oracle_check_swept(C.byref(params), C.byref(tb), n, dptr(x), dptr(y), dptr(theta),
                                        dptr(boxes) if want_boxes else None)
    return int(first), boxes


def check_poses(params, tables: OracleTables, x, y, theta):
    """Per-pose collision verdicts of one trajectory (``cc.collide`` of the ego rectangle at scenario time index
    ``time_step0 + i * factor``, reactive_planner.py:1033-1046): (hit [n] bool, any)."""
    x, y, theta = f64(x), f64(y), f64(theta)
    n = len(x)
    hit = np.zeros(n, dtype=np.int32)
    tb = tables.c_struct()
    lib().rp_oracle_check_poses(C.byref(params), C.byref(tb), n, dptr(x), dptr(y), dptr(theta), hit.ctypes.data_as(C.POINTER(C.c_int32)))
    return hit.astype(bool), bool(hit.any())


def obb_sum_rows(dyn_obb) -> np.ndarray:
    """trajectory_preprocess_obb_sum of every dynamic obstacle of a (n_dyn, n_steps, 5) table
    (commonroad_rp/reactive_planner.py:238-245)."""
    dyn = np.ascontiguousarray(dyn_obb, dtype=np.float64)
    out = np.full_like(dyn, np.nan)
    for j in range(dyn.shape[0]):
        lib().rp_oracle_obb_sum_rows(dyn.shape[1], dyn[j].ctypes.data_as(C.POINTER(C.c_double)),
                                     out[j].ctypes.data_as(C.POINTER(C.c_double)))
    return out
