"""ctypes binding of the C ABI in ``include/rp_amd.h`` (``librp_amd.so``, built from ``csrc/``).

This is the whole Python<->HIP boundary: plain pointers and sizes, no torch types.  The library is
loaded from ``<package root>/lib/librp_amd.so`` (built in-tree by ``__graft_entry__.build()`` or
``make -C commonroad-reactive-planner_amd/csrc``).  There is NO CPU fallback: if the library is
missing, importing the binding works but creating a context raises ``RpLibraryMissing``.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import sys
from typing import Optional, Sequence

import numpy as np

N_ARRAYS = 14
ARRAY_NAMES = ("x", "y", "theta", "v", "a", "kappa", "kappa_dot",
               "s", "d", "theta_cl", "s_dot", "s_ddot", "d_dot", "d_ddot")

LABEL_NONE, LABEL_FEASIBLE, LABEL_INFEASIBLE_KINEMATIC, LABEL_INFEASIBLE_COLLISION = 0, 1, 2, 3
REASON_NAMES = ("none", "velocity", "acceleration", "kappa", "kappa_dot", "yaw_rate", "out_of_domain")
CHECK_BITS = {"velocity": 1, "acceleration": 2, "kappa": 4, "kappa_dot": 8, "yaw_rate": 16}
LON_VELOCITY_KEEPING, LON_STOPPING = 0, 1
COST_DEFAULT, COST_FAILSAFE, COST_EXTERNAL = 0, 1, 2
FLAG_DRAW_ALL, FLAG_MATERIALIZE_ALL, FLAG_SKIP_COLLISION = 1, 2, 4
COLLISION_AUTO, COLLISION_EAGER, COLLISION_COST_ORDERED, COLLISION_TIMED = 0, 1, 2, 3
WAIT_SPIN, WAIT_YIELD, WAIT_EVENT = 0, 1, 2
KERNEL_NAMES = ("rp_eval_kernel", "rp_cost_kernel", "rp_chunk_kernel")

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "librp_amd.so")


class RpLibraryMissing(RuntimeError):
    pass


class RpError(RuntimeError):
    pass


ABI_VERSION = 2
E_ABI = -7
PLAN_BEGIN, PLAN_ROWS, PLAN_PACKED = 1, 2, 4


class _Sized(C.Structure):
    """the structs of include/rp_amd.h start with their own size (checked by every entry: RP_EABI)"""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))


class RpParams(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32), ("N", C.c_int32), ("factor", C.c_int32), ("time_step0", C.c_int32),
        ("low_vel_mode", C.c_int32), ("lon_mode", C.c_int32), ("constraint_mask", C.c_uint32),
        ("flags", C.c_uint32), ("dt", C.c_double),
        ("x0_lon", C.c_double * 3), ("x0_lat", C.c_double * 3), ("x0_orientation", C.c_double),
        ("wheelbase", C.c_double), ("wb_rear_axle", C.c_double), ("length", C.c_double), ("width", C.c_double),
        ("a_max", C.c_double), ("v_switch", C.c_double), ("delta_max", C.c_double), ("v_delta_max", C.c_double),
    ]


class RpCost(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("w_a", C.c_double), ("desired_speed", C.c_double),
                ("desired_d", C.c_double), ("desired_s", C.c_double)]


class RpGrids(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("nT", C.c_int32), ("nL", C.c_int32), ("nD", C.c_int32),
                ("T", C.POINTER(C.c_double)), ("traj_len", C.POINTER(C.c_int32)),
                ("L", C.POINTER(C.c_double)), ("D", C.POINTER(C.c_double))]


class RpResult(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("reserved_", C.c_uint32), ("best_index", C.c_int64), ("best_cost", C.c_double), ("n_candidates", C.c_int64),
                ("n_feasible", C.c_int64), ("n_collision_before_best", C.c_int64), ("n_collision", C.c_int64),
                ("reason_counts", C.c_int64 * 8), ("best_lon_coeffs", C.c_double * 6),
                ("best_lat_coeffs", C.c_double * 6), ("best_lat_T", C.c_double), ("kernel_ms", C.c_double)]


assert C.sizeof(RpResult) == 29 * 8 and RpResult.best_index.offset == 8

# The cycle's one foreign call without ctypes (csrc/rp_pyfast.c -> _rpfast.so beside this file; RpContext.plan_packed_fast).  A binding
# accelerator only: it calls the same rp_plan of the same library; an extension built against another header is not used.
try:
    from . import _rpfast
    if (_rpfast.ABI_VERSION, _rpfast.SIZEOF_PARAMS, _rpfast.SIZEOF_RESULT) != (ABI_VERSION, C.sizeof(RpParams), C.sizeof(RpResult)):
        _rpfast = None
except ImportError:
    _rpfast = None


class RpCycle(_Sized):
    """``rp_cycle``: the inputs and by-products of one replanning cycle (``rp_plan_cycle``)"""
    _fields_ = [("struct_size", C.c_uint32), ("have_curvilinear", C.c_int32), ("prev_low_vel_mode", C.c_int32), ("low_vel_mode", C.c_int32),
                ("level", C.c_int32), ("reserved_", C.c_int32),
                ("x", C.c_double), ("y", C.c_double), ("orientation", C.c_double), ("velocity", C.c_double), ("acceleration", C.c_double),
                ("steering_angle", C.c_double), ("low_vel_mode_threshold", C.c_double), ("x0_lon", C.c_double * 3), ("x0_lat", C.c_double * 3)]
_DP = C.POINTER(C.c_double)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


@dataclasses.dataclass
class PlanInputs:
    """Everything one ``rp_plan`` call consumes besides the tables held by the context."""
    params: RpParams
    cost: RpCost
    T: np.ndarray
    traj_len: np.ndarray
    L: np.ndarray
    D: np.ndarray

    def __post_init__(self):
        self.T = f64(self.T)
        self.L = f64(self.L)
        self.D = f64(self.D)
        self.traj_len = np.ascontiguousarray(self.traj_len, dtype=np.int32)
        assert self.T.ndim == self.L.ndim == self.D.ndim == 1 and self.traj_len.shape == self.T.shape

    @classmethod
    def trusted(cls, params, cost, T, traj_len, L, D) -> "PlanInputs":
        """Without the conversions of ``__post_init__``: the caller hands C-contiguous float64 / int32 arrays (the planner's
        own grids, once per replanning cycle)."""
        self = object.__new__(cls)
        self.params, self.cost, self.T, self.traj_len, self.L, self.D = params, cost, T, traj_len, L, D
        return self

    @property
    def n_candidates(self) -> int:
        return len(self.T) * len(self.L) * len(self.D)

    def grids(self) -> RpGrids:
        g = getattr(self, "_grids", None)
        if g is None:
            g = self._grids = RpGrids(0, len(self.T), len(self.L), len(self.D), dptr(self.T),
                                      self.traj_len.ctypes.data_as(C.POINTER(C.c_int32)), dptr(self.L), dptr(self.D))
        return g


@dataclasses.dataclass
class PlanOutput:
    best_index: int
    best_cost: float
    n_candidates: int
    n_feasible: int
    n_collision_before_best: int
    n_collision: int
    reason_counts: np.ndarray          # [8], index = reason code
    best_lon_coeffs: np.ndarray
    best_lat_coeffs: np.ndarray
    best_lat_T: float
    kernel_ms: float
    best_states: Optional[np.ndarray]  # [14, N+1] or None
    serial: int = 0                    # RpContext: which call on the context produced this result (0: not from a context)

    @classmethod
    def from_c(cls, r: RpResult, best_states):
        # one copy of the 28 eight-byte words of rp_result, read as doubles and as integers
        f = np.frombuffer(r, dtype=np.float64, count=28, offset=8).copy()   # (behind the struct_size header)
        i = f.view(np.int64)
        bi = int(i[0])
        return cls(bi, float(f[1]), int(i[2]), int(i[3]), int(i[4]), int(i[5]), i[6:14], f[14:20], f[20:26], float(f[26]),
                   float(f[27]), best_states if bi >= 0 else None)

    @property
    def n_infeasible_kinematics(self) -> int:
        """``ReactivePlanner.infeasible_count_kinematics`` (reactive_planner.py:1119)."""
        return self.n_candidates - self.n_feasible


def make_params(*, dt: float, N: int, x0_lon: Sequence[float], x0_lat: Sequence[float], x0_orientation: float,
                wheelbase: float, wb_rear_axle: float, length: float, width: float, a_max: float, v_switch: float,
                delta_max: float, v_delta_max: float, factor: int = 1, time_step0: int = 0, low_vel_mode: bool = False,
                lon_mode: int = LON_VELOCITY_KEEPING, constraint_mask: int = 0x1F, flags: int = 0) -> RpParams:
    p = RpParams()
    p.dt, p.N, p.factor, p.time_step0 = float(dt), int(N), int(factor), int(time_step0)
    p.low_vel_mode, p.lon_mode = int(bool(low_vel_mode)), int(lon_mode)
    p.constraint_mask, p.flags = int(constraint_mask), int(flags)
    p.x0_lon[:] = [float(v) for v in x0_lon]
    p.x0_lat[:] = [float(v) for v in x0_lat]
    p.x0_orientation = float(x0_orientation)
    p.wheelbase, p.wb_rear_axle, p.length, p.width = float(wheelbase), float(wb_rear_axle), float(length), float(width)
    p.a_max, p.v_switch, p.delta_max, p.v_delta_max = float(a_max), float(v_switch), float(delta_max), float(v_delta_max)
    return p


def make_cost(kind: int = COST_DEFAULT, w_a: float = 5.0, desired_speed: Optional[float] = None,
              desired_d: float = 0.0, desired_s: Optional[float] = None) -> RpCost:
    nan = float("nan")
    return RpCost(0, int(kind), float(w_a), nan if desired_speed is None else float(desired_speed), float(desired_d),
                  nan if desired_s is None else float(desired_s))


def copy_params(p: RpParams) -> RpParams:
    q = RpParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(RpParams))
    return q


# ------------------------------------------------------------------------------------------------
_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load ``librp_amd.so`` and declare every entry point of ``include/rp_amd.h``."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("RP_AMD_LIBRARY", LIB_PATH)
    # One HIP runtime per process, and the same one whatever the import order: PyTorch-ROCm brings its own libamdhip64
    # (same SONAME as the system's), so whichever of the two libraries is loaded first decides which runtime both use.
    # With this library first, a process that created and destroyed a context and imported torch afterwards found
    # "No HIP GPUs are available" (observed on the GPU box, ROCm 7.2 system runtime under a torch built for 7.0); with
    # torch first -- the order bench.py has always had -- everything runs on torch's runtime.  So: torch first, when it
    # is installed and the process may use it (the multi-GPU exchange does).  RP_AMD_NO_TORCH_PRELOAD=1 skips this.
    if "torch" not in sys.modules and not os.environ.get("RP_AMD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise RpLibraryMissing(
            f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C commonroad-reactive-planner_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(path)
    ctx = C.c_void_p
    dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    sig = {
        "rp_abi_version": (C.c_int, []),
        "rp_source_hash": (C.c_char_p, []),
        "rp_create": (C.c_int, [C.POINTER(ctx), C.c_int]),
        "rp_destroy": (None, [ctx]),
        "rp_last_error": (C.c_char_p, [ctx]),
        "rp_set_profiling": (C.c_int, [ctx, C.c_int]),
        "rp_set_wait_mode": (C.c_int, [ctx, C.c_int]),
        "rp_get_wait_mode": (C.c_int, [ctx]),
        "rp_set_option": (C.c_int, [ctx, C.c_char_p, C.c_int64]),
        "rp_get_option": (C.c_int, [ctx, C.c_char_p, C.POINTER(C.c_int64)]),
        "rp_last_path": (C.c_int, [ctx]),
        "rp_last_kernel": (C.c_int, [ctx]),
        "rp_set_collision_path": (C.c_int, [ctx, C.c_int]),
        "rp_set_reference": (C.c_int, [ctx, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_double]),
        "rp_set_obstacles": (C.c_int, [ctx, C.c_int32, dp, C.c_int32, dp, C.c_int32, dp, C.c_int32, C.c_int32,
                                       C.c_int32, dp]),
        # (out: void pointer -- a NumPy buffer or a ctypes array of the packed calls)
        "rp_plan": (C.c_int, [ctx, C.POINTER(RpParams), C.POINTER(RpCost), C.POINTER(RpGrids), C.c_int64, C.c_int64, C.c_uint32,
                              C.POINTER(RpResult), C.c_void_p]),
        "rp_plan_wait": (C.c_int, [ctx, C.POINTER(RpResult), dp]),
        "rp_plan_levels": (C.c_int, [ctx, C.POINTER(RpParams), C.POINTER(RpCost), C.c_int32, C.POINTER(RpGrids), C.c_uint32, C.POINTER(RpResult),
                                     C.c_void_p, ip]),
        "rp_last_level": (C.c_int, [ctx]),
        "rp_plan_cycle": (C.c_int, [ctx, C.POINTER(RpCycle), C.POINTER(RpParams), C.POINTER(RpCost), C.c_int32, C.POINTER(RpGrids), C.POINTER(RpResult),
                                    C.c_void_p]),
        "rp_pack_trajectory": (C.c_int, [C.c_int32, dp, C.c_double, C.c_double, C.c_double, dp]),
        "rp_fast_buffer": (C.c_int, [ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
        "rp_plan_coeffs": (C.c_int, [ctx, C.POINTER(RpParams), C.POINTER(RpCost), C.c_int64, dp, dp, dp, ip, C.c_int64, ip, ip,
                                     C.POINTER(RpResult), dp]),
        "rp_coeffs_arena": (C.c_int, [ctx, C.c_int64, C.POINTER(dp), C.POINTER(dp), C.POINTER(ip)]),
        "rp_coeffs_arena_groups": (C.c_int, [ctx, C.POINTER(ip), C.POINTER(ip)]),
        "rp_fetch_status": (C.c_int, [ctx, C.c_int64, C.c_int64, up, dp]),
        "rp_fetch_states": (C.c_int, [ctx, C.c_int64, C.c_int64, dp]),
        "rp_eval_one": (C.c_int, [ctx, C.c_int64, dp, up, dp]),
        "rp_count_collisions_before": (C.c_int, [ctx, C.c_double, C.c_int64, C.POINTER(C.c_int64)]),
        "rp_select": (C.c_int, [ctx, dp, C.c_int64, C.POINTER(RpResult), dp]),
        "rp_cost_range": (C.c_int, [ctx, dp, dp, C.POINTER(C.c_int64)]),
        "rp_check_swept": (C.c_int, [ctx, C.POINTER(RpParams), C.c_int32, dp, dp, dp, ip, dp]),
        "rp_pyset_order": (C.c_int, [C.c_int32, dp, C.c_int32, dp, ip]),
        "rp_corridor_coeffs": (C.c_int, [C.c_int32, dp, ip, dp, dp, ip, dp, C.c_int32, dp, dp, C.c_int64, C.POINTER(C.c_int64), dp, dp, dp, ip, dp, dp]),
        "rp_corridor_coeffs_grouped": (C.c_int, [C.c_int32, dp, ip, dp, dp, ip, dp, C.c_int32, dp, dp, C.c_int64, C.POINTER(C.c_int64), dp, dp, dp, ip,
                                                 dp, dp, ip, ip, C.POINTER(C.c_int64)]),
        "rp_build_reference": (C.c_int, [C.c_int32, dp, C.c_int32, C.c_double, C.c_int32, ip, dp, dp, dp, dp, dp]),
        "rp_project": (C.c_int, [C.c_int32, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp]),
        "rp_initial_state": (C.c_int, [C.c_int32, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_double, C.c_int32, dp, dp]),
        "rp_result_device": (C.c_int, [ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), ip]),
        "rp_combine_results": (C.c_int, [ctx, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(RpResult), dp, ip, ip]),
        "rp_mailbox_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
        "rp_mailbox_exchange": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(RpResult), dp,
                                          C.POINTER(RpResult), dp, ip]),
        "rp_mailbox_sum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int64,
                                     C.POINTER(C.c_int64)]),
        "rp_mailbox_set_timeout": (C.c_int, [C.c_double]),
        "rp_mailbox_stalled_rank": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)   # AttributeError here = the library does not export the header's symbol
        except AttributeError:
            if path == LIB_PATH or name not in _OPTIONAL_IN_AB_BUILDS:
                raise
            continue                  # (an older build named by RP_AMD_LIBRARY for an A/B run: entries added since are absent)
        fn.restype, fn.argtypes = res, args
    if lib.rp_abi_version() != ABI_VERSION:
        raise RpError(f"librp_amd.so ABI version {lib.rp_abi_version()} != {ABI_VERSION}")
    if path == LIB_PATH or _lib is None:
        _lib = lib
    return lib


_OPTIONAL_IN_AB_BUILDS = ("rp_set_wait_mode", "rp_get_wait_mode", "rp_set_option", "rp_get_option", "rp_last_kernel", "rp_plan_levels", "rp_last_level", "rp_plan_cycle", "rp_set_collision_path", "rp_last_path", "rp_source_hash", "rp_plan_wait", "rp_pack_trajectory", "rp_pyset_order",
                          "rp_corridor_coeffs", "rp_fast_buffer", "rp_coeffs_arena", "rp_coeffs_arena_groups",
                          "rp_corridor_coeffs_grouped")
EXPORTED_SYMBOLS = ("rp_abi_version", "rp_source_hash", "rp_create", "rp_destroy", "rp_last_error", "rp_set_profiling", "rp_set_wait_mode", "rp_get_wait_mode", "rp_set_option", "rp_get_option", "rp_last_path", "rp_last_kernel", "rp_set_collision_path",
                    "rp_set_reference", "rp_set_obstacles", "rp_plan", "rp_plan_wait", "rp_plan_levels", "rp_last_level", "rp_plan_cycle", "rp_pack_trajectory", "rp_fast_buffer", "rp_plan_coeffs", "rp_coeffs_arena", "rp_coeffs_arena_groups", "rp_fetch_status",
                    "rp_fetch_states", "rp_eval_one", "rp_count_collisions_before", "rp_select",
                    "rp_cost_range", "rp_check_swept", "rp_pyset_order", "rp_corridor_coeffs", "rp_corridor_coeffs_grouped", "rp_build_reference", "rp_project", "rp_initial_state", "rp_result_device", "rp_combine_results", "rp_mailbox_bytes", "rp_mailbox_exchange", "rp_mailbox_sum",
                    "rp_mailbox_set_timeout", "rp_mailbox_stalled_rank")


def source_hash() -> str:
    """Hash of the sources the loaded library was built from ("unknown" for a build that does not carry one)."""
    fn = getattr(load_library(), "rp_source_hash", None)
    return (fn() or b"unknown").decode() if fn is not None else "unknown"


def pack_trajectory(states: np.ndarray, dt: float, wheelbase: float, x0_orientation: float) -> np.ndarray:
    """``rp_pack_trajectory``: [n, 13] = x, y | s, d | s, s', s'' | d, d', d'' | shifted orientation | steering angle | yaw rate of a
    state block [14, n] (what ``_compute_trajectory_pair`` packs, reactive_planner.py:514-568)."""
    n = states.shape[1]
    out = np.empty((n, 13))
    rc = load_library().rp_pack_trajectory(n, states.ctypes.data_as(_DP), dt, wheelbase, x0_orientation, out.ctypes.data_as(_DP))
    if rc != 0:
        raise RpError(f"rp_pack_trajectory -> {rc}")
    return out


def pyset_order(values, union_zero: bool = False) -> np.ndarray:
    """``rp_pyset_order``: the values of ``set(values)`` (``set(values).union({0})``) in the iteration order of a CPython set, as
    the library restates it."""
    v = f64(values)
    out = np.empty(len(v) + 1)
    n = C.c_int32(0)
    rc = load_library().rp_pyset_order(len(v), dptr(v), int(bool(union_zero)), dptr(out), C.byref(n))
    if rc != 0:
        raise RpError(f"rp_pyset_order -> {rc}")
    return out[:n.value].copy()


def corridor_coeffs(T, traj_len, v_low, v_up, box_off, boxes, n_samples: int, x0_lon, x0_lat, buffers: Optional[dict] = None):
    """``rp_corridor_coeffs``: (lon_coeffs [C, 6], lat_coeffs [C, 6], lon_T [C], traj_len [C], lon_end [C], lat_end [C]).
    ``buffers``: a dict that keeps the output arrays between calls (the result then consists of views into them, valid until the
    next call with the same dict: tens of thousands of candidates are a few MB, whose page faults cost more than the call);
    ``buffers["alloc"]``, if set: cap -> (lon [cap, 6], lat [cap, 6], traj_len [cap]) to write into instead of fresh arrays."""
    lib = load_library()
    T, v_low, v_up, boxes = f64(T), f64(v_low), f64(v_up), f64(boxes)
    traj_len = np.ascontiguousarray(traj_len, dtype=np.int32)
    box_off = np.ascontiguousarray(box_off, dtype=np.int32)
    x0_lon, x0_lat = f64(x0_lon), f64(x0_lat)
    ipt = C.POINTER(C.c_int32)
    keep = buffers if buffers is not None else {}
    cap = keep.get("cap", 16384)
    for _ in range(2):
        if keep.get("cap") != cap or "arrays" not in keep:
            keep["cap"] = cap
            alloc = keep.get("alloc")     # (a planner lets the candidates be written into its context's pinned arena: RpContext.coeffs_arena)
            lon_lat_tl = alloc(cap) if alloc is not None else (np.empty((cap, 6)), np.empty((cap, 6)), np.empty(cap, dtype=np.int32))
            keep["arrays"] = (lon_lat_tl[0], lon_lat_tl[1], np.empty(cap), lon_lat_tl[2], np.empty(cap), np.empty(cap))
            # (an arena also has room for the candidates' groups -- lateral samples of one (T, v) sample share their longitudinal
            #  polynomial: RpContext.coeffs_arena hands them out as items 3 and 4)
            keep["group_arrays"] = (lon_lat_tl[3], lon_lat_tl[4]) if len(lon_lat_tl) >= 5 else None
        lon, lat, lt, tl, le, de = keep["arrays"]
        cnt = C.c_int64(0)
        ga = keep.get("group_arrays")   # (group [cap], group_first [cap]) of the same arena, or None
        if ga is not None and getattr(lib, "rp_corridor_coeffs_grouped", None) is not None:
            ng = C.c_int64(0)
            rc = lib.rp_corridor_coeffs_grouped(len(T), dptr(T), traj_len.ctypes.data_as(ipt), dptr(v_low), dptr(v_up), box_off.ctypes.data_as(ipt),
                                                dptr(boxes), int(n_samples), dptr(x0_lon), dptr(x0_lat), cap, C.byref(cnt), dptr(lon), dptr(lat),
                                                dptr(lt), tl.ctypes.data_as(ipt), dptr(le), dptr(de), ga[0].ctypes.data_as(ipt),
                                                ga[1].ctypes.data_as(ipt), C.byref(ng))
            keep["groups"] = (int(ng.value), ga[0][:int(cnt.value)], ga[1][:int(ng.value)]) if rc == 0 else None
        else:
            keep["groups"] = None
            rc = lib.rp_corridor_coeffs(len(T), dptr(T), traj_len.ctypes.data_as(ipt), dptr(v_low), dptr(v_up), box_off.ctypes.data_as(ipt),
                                        dptr(boxes), int(n_samples), dptr(x0_lon), dptr(x0_lat), cap, C.byref(cnt), dptr(lon), dptr(lat),
                                        dptr(lt), tl.ctypes.data_as(ipt), dptr(le), dptr(de))
        k = int(cnt.value)
        if rc == 0:
            return lon[:k], lat[:k], lt[:k], tl[:k], le[:k], de[:k]
        if rc != -4:
            raise RpError(f"rp_corridor_coeffs -> {rc}")
        cap = max(k, 2 * cap)
    raise RpError("rp_corridor_coeffs: output size changed between calls")


# ---- reference-path front end (host-only entry points: no context, no GPU) -------------------------
E_DOMAIN, E_DIRECTION = -5, -6


def build_reference(reference, smooth: bool = True, resample_step: float = 1.0):
    """``rp_build_reference``: (polyline (n, 2), ref_pos, ref_theta, ref_curv, ref_curv_d) of a route centre line --
    what ``CoordinateSystem.__init__`` computes (utils_coordinate_system.py:88-118)."""
    lib = load_library()
    xy = f64(reference)
    if xy.ndim != 2 or xy.shape[1] != 2 or len(xy) < 2:
        raise ValueError("reference path: expected an (n, 2) array with n >= 2")
    cap = max(len(xy), 256)
    for _ in range(2):
        out = np.empty((cap, 2))
        tabs = [np.empty(cap) for _ in range(4)]
        n = C.c_int32(0)
        rc = lib.rp_build_reference(len(xy), dptr(xy), int(bool(smooth)), float(resample_step), cap, C.byref(n), dptr(out),
                                    *[dptr(t) for t in tabs])
        if rc == 0:
            k = n.value
            return (np.ascontiguousarray(out[:k]),) + tuple(np.ascontiguousarray(t[:k]) for t in tabs)
        if rc != -4:
            raise ValueError(f"rp_build_reference -> {rc}: unusable reference path (fewer than "
                             f"{4 if smooth else 2} distinct vertices, or non-finite coordinates)")
        cap = n.value
    raise RpError("rp_build_reference: output size changed between calls")


def project(ref_xy, ref_pos, x: float, y: float, proj_domain_d_limit: float = 20.0):
    """``rp_project``: (s, d) of a Cartesian point; ValueError outside the projection domain, as
    ``CoordinateSystem.convert_to_curvilinear_coords`` raises (utils_coordinate_system.py:176-178)."""
    lib = load_library()
    s, d = C.c_double(), C.c_double()
    rc = lib.rp_project(len(ref_pos), dptr(ref_xy), dptr(ref_pos), float(proj_domain_d_limit), float(x), float(y),
                        C.byref(s), C.byref(d))
    if rc == E_DOMAIN:
        raise ValueError("<CoordinateSystem.convert_to_curvilinear_coords>: point outside projection domain")
    if rc != 0:
        raise RpError(f"rp_project -> {rc}")
    return np.array([s.value, d.value])


def initial_state(ref_xy, ref_pos, ref_theta, ref_curv, ref_curv_d, x, y, orientation, velocity, acceleration=0.0,
                  steering_angle=0.0, wheelbase=1.0, low_vel_mode=False, proj_domain_d_limit=20.0):
    """``rp_initial_state``: (x_0_lon, x_0_lat) as ``ReactivePlanner._compute_initial_states`` returns them
    (reactive_planner.py:446-512), with the exceptions the reference raises."""
    lib = load_library()
    lon, lat = (C.c_double * 3)(), (C.c_double * 3)()
    rc = lib.rp_initial_state(len(ref_pos), dptr(ref_xy), dptr(ref_pos), dptr(ref_theta), dptr(ref_curv), dptr(ref_curv_d),
                              float(proj_domain_d_limit), float(x), float(y), float(orientation), float(velocity),
                              float(acceleration), float(steering_angle), float(wheelbase), int(bool(low_vel_mode)),
                              C.cast(lon, _DP), C.cast(lat, _DP))
    if rc == E_DOMAIN:
        raise ValueError("Initial state could not be transformed.")
    if rc == E_DIRECTION:
        raise Exception("Initial state or reference incorrect! The longitudinal velocity along the reference path is negative: "
                        "the ego vehicle does not drive in the direction of the reference path")
    if rc != 0:
        raise RpError(f"rp_initial_state -> {rc}")
    return list(lon), list(lat)


# Options every context of this process gets on top of the library's own defaults (tests and A/B measurements pin launch paths with
# them: tests/_paths.py); ``set_default_options`` also applies them to the contexts that are alive.
_default_options: dict = {}
_live_contexts = None   # weakref.WeakSet of RpContext, created with the first context


def set_default_options(options: Optional[dict]):
    """Replace the process-wide option set: every live context goes back to the values it was created with and then takes
    ``options`` (``rp_set_option`` names, include/rp_amd.h); contexts created later start with them."""
    global _default_options
    _default_options = dict(options or {})
    for ctx in list(_live_contexts or ()):
        if getattr(ctx, "_h", None):
            ctx._apply_default_options()


class RpContext:
    """Owner of one ``rp_ctx`` (device tables, work buffers, one HIP stream)."""

    def __init__(self, device: int = 0, library: Optional[str] = None):
        global _live_contexts
        self._lib = load_library(library)
        self._h = C.c_void_p()
        rc = self._lib.rp_create(C.byref(self._h), int(device))
        if rc != 0:
            msg = self._lib.rp_last_error(self._h) if self._h else b"rp_create failed"
            raise RpError(f"rp_create(device={device}) -> {rc}: {(msg or b'').decode()}")
        self._created_options = {}   # option -> the value rp_create gave it (restored when the process-wide set drops the option)
        if _live_contexts is None:
            import weakref
            _live_contexts = weakref.WeakSet()
        _live_contexts.add(self)
        if _default_options:
            self._apply_default_options()
        self.device = device
        self._N = None
        self._last_count = 0
        self._res = RpResult()    # C result of the last plan / plan_coeffs / select on this context ...
        self._last_best = None    # ... and its winner state block (what the intra-node exchange posts as they are)
        self._serial = 0          # ... and the number of that call (PlanOutput.serial)
        if _rpfast is None:       # (the extension has not been built: callers ask with getattr(ctx, "plan_packed_fast", None))
            self.plan_packed_fast = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_destroy(self._h)
            self._h = None
        self._arena = self._fast = self._fast_last = self._pk = None   # (views of memory the context owned)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RpError(f"{what} -> {rc}: {(self._lib.rp_last_error(self._h) or b'').decode()}")

    def set_option(self, key: str, value: int):
        """``rp_set_option``: a launch-policy switch of this context by name (include/rp_amd.h lists them)."""
        self._check(self._lib.rp_set_option(self._h, key.encode(), int(value)), f"rp_set_option({key!r}, {value})")

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        self._check(self._lib.rp_get_option(self._h, key.encode(), C.byref(v)), f"rp_get_option({key!r})")
        return int(v.value)

    def _apply_default_options(self):
        for k, v in self._created_options.items():
            if k not in _default_options:
                self.set_option(k, v)
        for k, v in _default_options.items():
            if k not in self._created_options:
                self._created_options[k] = self.get_option(k)
            self.set_option(k, v)

    def set_wait_mode(self, mode: int):
        """``rp_set_wait_mode``: WAIT_SPIN (default), WAIT_YIELD (poll + sched_yield), WAIT_EVENT (sleep on a HIP event)."""
        self._check(self._lib.rp_set_wait_mode(self._h, int(mode)), "rp_set_wait_mode")

    def wait_mode(self) -> int:
        return int(self._lib.rp_get_wait_mode(self._h))

    def wait_mode_name(self) -> str:
        return ("spin", "yield", "event")[self.wait_mode()]

    def set_profiling(self, every: int):
        """Time the evaluation kernel of every ``every``-th plan with HIP events (0 / False = off)."""
        self._check(self._lib.rp_set_profiling(self._h, int(every)), "rp_set_profiling")

    def last_path(self) -> int:
        """How the last plan answered the collision query: 0 eager (every pose of every candidate), 1 cost-ordered (costs first,
        candidates in ascending cost until the first free one), 2 cost-ordered stage exhausted, eager kernel decided."""
        fn = getattr(self._lib, "rp_last_path", None)
        return int(fn(self._h)) if fn is not None else 0

    def last_kernel(self) -> str:
        """``rp_last_kernel``: the kernel that evaluated the batch of the last plan -- "rp_eval_kernel" (lanes over time steps) or
        "rp_cost_kernel" (one lane per candidate; large batches that keep costs and labels only) or "rp_chunk_kernel" (one lane per candidate
        and step block of 16; mid-sized batches that keep costs and labels only)."""
        fn = getattr(self._lib, "rp_last_kernel", None)
        return KERNEL_NAMES[int(fn(self._h))] if fn is not None else "rp_eval_kernel"

    def set_collision_path(self, mode: int):
        """``rp_set_collision_path``: how production-mode plans of large batches answer the collision query -- COLLISION_AUTO
        (default: cost-ordered with a back-off that depends on the sequence of plans only), COLLISION_EAGER, COLLISION_COST_ORDERED,
        COLLISION_TIMED (the context's clock picks the faster path)."""
        self._check(self._lib.rp_set_collision_path(self._h, int(mode)), "rp_set_collision_path")

    def set_reference(self, ref_pos, ref_theta, ref_curv, ref_curv_d, ref_xy, proj_domain_d_limit: float = 20.0):
        ref_pos, ref_theta, ref_curv, ref_curv_d = f64(ref_pos), f64(ref_theta), f64(ref_curv), f64(ref_curv_d)
        xy = f64(ref_xy)
        n = len(ref_pos)
        assert xy.shape == (n, 2) and len(ref_theta) == len(ref_curv) == len(ref_curv_d) == n
        x, y = f64(xy[:, 0]), f64(xy[:, 1])
        self._check(self._lib.rp_set_reference(self._h, n, dptr(ref_pos), dptr(ref_theta), dptr(ref_curv),
                                               dptr(ref_curv_d), dptr(x), dptr(y), float(proj_domain_d_limit)),
                    "rp_set_reference")

    def set_coordinate_system(self, co):
        """Accepts anything with the reference ``CoordinateSystem`` attributes
        (utils_coordinate_system.py:114-118, :120-123)."""
        self.set_reference(co.ref_pos, co.ref_theta, co.ref_curv, co.ref_curv_d, co.reference,
                           getattr(co, "proj_domain_d_limit", 20.0))

    def set_obstacles(self, tables=None):
        from .collision import ObstacleTables
        tb = tables if tables is not None else ObstacleTables()
        nd, ns = tb.dyn_obb.shape[0], tb.dyn_obb.shape[1]
        self._check(self._lib.rp_set_obstacles(
            self._h, len(tb.static_obb), dptr(tb.static_obb), len(tb.static_tri), dptr(tb.static_tri),
            len(tb.static_circ), dptr(tb.static_circ), nd, ns, int(tb.dyn_t0), dptr(tb.dyn_obb)), "rp_set_obstacles")

    def plan(self, inp: PlanInputs, cand_begin: int = 0, cand_end: int = -1, want_best_states: bool = True) -> PlanOutput:
        res = self._res
        n = inp.params.N + 1
        best = np.empty((N_ARRAYS, n)) if want_best_states else None
        g = inp.grids()
        rc = self._lib.rp_plan(self._h, C.byref(inp.params), C.byref(inp.cost), C.byref(g), cand_begin, cand_end, 0,
                               C.byref(res), best.ctypes.data if best is not None else None)
        if rc != 0:
            self._check(rc, "rp_plan")
        self._N = inp.params.N
        self._last_count = res.n_candidates
        return self._output(res, best)

    def _packed_out(self, n: int):
        """Fresh output block of the packed calls, [14 + 13][n] doubles: (ctypes array -- what the call takes as its pointer --, the same
        memory as a 2-D array).  ``np.empty(...).ctypes.data`` costs a microsecond per call; a ctypes array of a cached type and an
        array over its buffer a third of that."""
        types = self.__dict__.get("_out_types")
        if types is None:
            types = self._out_types = {}
        t = types.get(n)
        if t is None:
            t = types[n] = C.c_double * ((N_ARRAYS + 13) * n)
        raw = t()
        return raw, np.ndarray((N_ARRAYS + 13, n), np.float64, raw)

    def plan_packed(self, params: RpParams, cost: RpCost, T, traj_len, L, D):
        """``rp_plan`` with ``RP_PLAN_PACKED``: one sampling level -> (the context's ``RpResult`` struct, state block [14, N + 1], packed output
        [N + 1, 13]) -- the last two ``None`` without a winner.  The grids go through the context's own buffer (four slice
        assignments instead of a struct of four array pointers built per call) and the result comes back as the C struct."""
        fb = getattr(self, "_fast", None)
        if fb is None:
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            self._check(self._lib.rp_fast_buffer(self._h, C.byref(ptr), C.byref(nbytes)), "rp_fast_buffer")
            raw = (C.c_char * nbytes.value).from_address(ptr.value)
            fb = self._fast = (np.frombuffer(raw, dtype=np.float64), np.frombuffer(raw, dtype=np.int32), self._lib.rp_plan)
            self._fast_dims = RpGrids()   # (sizes only: the arrays are in the buffer)
        f64v, i32v, call = fb
        nT, nL, nD = len(T), len(L), len(D)
        nd = nT + nL + nD
        if nd * 8 + nT * 4 > f64v.nbytes:
            raise RpError("plan_packed: grids larger than the context's buffer")
        # (the sampling space hands out the same array objects while its sets are unchanged: what the buffer already holds at the
        #  same place is not written again -- in a replanning loop only the velocity samples change from cycle to cycle)
        last = getattr(self, "_fast_last", None)
        same = last is not None and last[0] == (nT, nL, nD)
        #  (read-only arrays only: an array its owner may write into says nothing by being the same object)
        if not (same and last[1] is T and last[4] is traj_len and not T.flags.writeable and not traj_len.flags.writeable):
            f64v[0:nT] = T
            i32v[2 * nd:2 * nd + nT] = traj_len
        if not (same and last[2] is L and not L.flags.writeable):
            f64v[nT:nT + nL] = L
        if not (same and last[3] is D and not D.flags.writeable):
            f64v[nT + nL:nd] = D
        self._fast_last = ((nT, nL, nD), T, L, D, traj_len)
        self._cycle_last = None
        n = params.N + 1
        raw, out = self._packed_out(n)
        res = self._res
        gd = self._fast_dims
        if not same:
            gd.nT, gd.nL, gd.nD = nT, nL, nD
        rc = call(self._h, params, cost, gd, 0, -1, PLAN_PACKED, res, raw)
        if rc != 0:
            self._check(rc, "rp_plan (packed)")
        self._N = n - 1
        self._last_count = res.n_candidates
        self._serial += 1
        self._last_best = None
        if res.best_index < 0:
            return res, None, None
        return res, out[:N_ARRAYS], out[N_ARRAYS:].reshape(n, 13)

    def plan_packed_fast(self, params: RpParams, cost: RpCost, T, traj_len, L, D, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation):
        """``plan_packed`` with the cycle's fields of ``params`` (time step, low-velocity flag, flags, curvilinear state, orientation)
        written, the grids copied and ``rp_plan`` called by the ``_rpfast`` extension (csrc/rp_pyfast.c) instead of through ctypes
        descriptors, slice assignments and argument marshalling: same call, same results, ~5 us less of Python per replanning cycle
        (profiles/r05_plan_latency.txt).  ``None`` (the attribute, see ``__init__``) when the extension has not been built."""
        pk = self.__dict__.get("_pk")
        if pk is None:
            if not self._h:
                raise RpError("plan_packed_fast: the context is closed")
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            self._check(self._lib.rp_fast_buffer(self._h, C.byref(ptr), C.byref(nbytes)), "rp_fast_buffer")
            pk = self._pk = _rpfast.Packed(C.cast(self._lib.rp_plan, C.c_void_p).value, self._h.value, ptr.value, nbytes.value, C.addressof(self._res))
        n = params.N + 1
        out = np.empty((N_ARRAYS + 13, n), np.float64)   # (fresh per call: the last cycle's winner may still be referred to)
        self._fast_last = self._cycle_last = None        # (the buffer no longer holds what the ctypes paths remember of it)
        rc = pk.plan(params, cost, T, traj_len, L, D, out, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation)
        if rc != 0:
            self._check(rc, "rp_plan (packed)")
        res = self._res
        self._N = n - 1
        self._last_count = res.n_candidates
        self._serial += 1
        self._last_best = None
        if res.best_index < 0:
            return res, None, None
        return res, out[:N_ARRAYS], out[N_ARRAYS:].reshape(n, 13)

    def plan_levels_packed(self, params: RpParams, cost: RpCost, levels):
        """``rp_plan_levels`` with ``RP_PLAN_PACKED``: the level loop of ``plan()`` (reactive_planner.py:616-636) in one call and one device round
        trip.  ``levels``: [(T, traj_len, L, D), ...] in the order the loop would visit them -> (result struct, index of the level
        the result belongs to, state block [14, N + 1], packed output [N + 1, 13]); the last two ``None`` without a winner."""
        fb = getattr(self, "_fast", None)
        if fb is None:
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            self._check(self._lib.rp_fast_buffer(self._h, C.byref(ptr), C.byref(nbytes)), "rp_fast_buffer")
            raw = (C.c_char * nbytes.value).from_address(ptr.value)
            fb = self._fast = (np.frombuffer(raw, dtype=np.float64), np.frombuffer(raw, dtype=np.int32), self._lib.rp_plan)
            self._fast_dims = RpGrids()
        f64v, i32v = fb[0], fb[1]
        nlev = len(levels)
        dims = getattr(self, "_lvl_dims", None)
        if dims is None or len(dims) < nlev:
            dims = self._lvl_dims = (RpGrids * max(nlev, 8))()
            for g_ in dims:
                g_.struct_size = C.sizeof(RpGrids)
            self._lvl_out = C.c_int32(0)
        at = 0
        for k, (T, traj_len, L, D) in enumerate(levels):
            nT, nL, nD = len(T), len(L), len(D)
            nd = nT + nL + nD
            words = nd + (nT + 1) // 2
            if (at + words) * 8 > f64v.nbytes:
                raise RpError("plan_levels_packed: grids larger than the context's buffer")
            f64v[at:at + nT] = T
            f64v[at + nT:at + nT + nL] = L
            f64v[at + nT + nL:at + nd] = D
            i32v[2 * (at + nd):2 * (at + nd) + nT] = traj_len
            gk = dims[k]
            gk.nT, gk.nL, gk.nD = nT, nL, nD
            at += words
        self._fast_last = None   # (the buffer no longer holds what plan_packed left there)
        self._cycle_last = None
        n = params.N + 1
        raw, out = self._packed_out(n)
        res, lvl = self._res, self._lvl_out
        rc = self._lib.rp_plan_levels(self._h, params, cost, nlev, dims, PLAN_PACKED, res, raw, lvl)
        if rc != 0:
            self._check(rc, "rp_plan_levels (packed)")
        self._N = n - 1
        self._last_count = res.n_candidates
        self._serial += 1
        self._last_best = None
        if res.best_index < 0:
            return res, lvl.value, None, None
        return res, lvl.value, out[:N_ARRAYS], out[N_ARRAYS:].reshape(n, 13)

    def plan_cycle(self, cycle: RpCycle, params: RpParams, cost: RpCost, levels):
        """``rp_plan_cycle``: one foreign call per replanning cycle.  ``levels``: [(T, traj_len, L, Dlin), ...] -- per level the
        time samples, the longitudinal samples (set order, unfiltered) and the INSERTION sequence of the lateral sample set
        (``np.linspace(d_min, d_max, n)``); the call appends the current lateral offset the way ``set.union`` does, filters goals
        behind the vehicle in stopping mode, works the initial state out if ``cycle.have_curvilinear`` is 0, runs the level loop and
        packs the output -> (result struct, state block [14, N + 1], packed output [N + 1, 13]) (``None``, ``None`` without a
        winner); ``cycle.level`` / ``.low_vel_mode`` / ``.x0_lon`` / ``.x0_lat`` are filled in.  Arrays that are the same read-only
        objects at the same place as in the previous call are not staged again."""
        fb = getattr(self, "_fast", None)
        if fb is None:
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            self._check(self._lib.rp_fast_buffer(self._h, C.byref(ptr), C.byref(nbytes)), "rp_fast_buffer")
            raw = (C.c_char * nbytes.value).from_address(ptr.value)
            fb = self._fast = (np.frombuffer(raw, dtype=np.float64), np.frombuffer(raw, dtype=np.int32), self._lib.rp_plan)
            self._fast_dims = RpGrids()
        f64v, i32v = fb[0], fb[1]
        nlev = len(levels)
        dims = getattr(self, "_lvl_dims", None)
        if dims is None or len(dims) < nlev:
            dims = self._lvl_dims = (RpGrids * max(nlev, 8))()
            for g_ in dims:
                g_.struct_size = C.sizeof(RpGrids)
            self._lvl_out = C.c_int32(0)
        last = self.__dict__.get("_cycle_last")
        same_shape = last is not None and len(last) == nlev
        at = 0
        held = []
        for k, (T, traj_len, L, D) in enumerate(levels):
            nT, nL, nD = len(T), len(L), len(D)
            nd = nT + nL + nD
            words = nd + (nT + 1) // 2
            if (at + words) * 8 > f64v.nbytes:
                raise RpError("plan_cycle: grids larger than the context's buffer")
            lk = last[k] if same_shape else None
            keep = lk is not None and lk[0] == (at, nT, nL, nD)
            if not (keep and lk[1] is T and lk[4] is traj_len and not T.flags.writeable and not traj_len.flags.writeable):
                f64v[at:at + nT] = T
                i32v[2 * (at + nd):2 * (at + nd) + nT] = traj_len
            if not (keep and lk[2] is L and not L.flags.writeable):
                f64v[at + nT:at + nT + nL] = L
            if not (keep and lk[3] is D and not D.flags.writeable):
                f64v[at + nT + nL:at + nd] = D
            if not keep:
                gk = dims[k]
                gk.nT, gk.nL, gk.nD = nT, nL, nD
            held.append(((at, nT, nL, nD), T, L, D, traj_len))
            at += words
        self._cycle_last = held
        self._fast_last = None   # (the buffer no longer holds what plan_packed left there)
        n = params.N + 1
        raw, out = self._packed_out(n)
        res = self._res
        rc = self._lib.rp_plan_cycle(self._h, cycle, params, cost, nlev, dims, res, raw)
        if rc != 0:
            if rc in (E_DOMAIN, E_DIRECTION):
                return rc, None, None
            self._check(rc, "rp_plan_cycle")
        self._N = n - 1
        self._last_count = res.n_candidates
        self._serial += 1
        self._last_best = None
        if res.best_index < 0:
            return res, None, None
        return res, out[:N_ARRAYS], out[N_ARRAYS:].reshape(n, 13)

    def plan_levels_begin(self, params: RpParams, cost: RpCost, levels, want_best_states: bool = True):
        """First half of ``rp_plan_levels`` (collected by ``plan_wait``, which then also reports ``last_level()``)."""
        nlev = len(levels)
        inputs = [PlanInputs.trusted(params, cost, T, tl, L, D) for (T, tl, L, D) in levels]
        arr = (RpGrids * nlev)(*[q.grids() for q in inputs])
        rc = self._lib.rp_plan_levels(self._h, C.byref(params), C.byref(cost), nlev, arr, PLAN_BEGIN | (PLAN_ROWS if want_best_states else 0),
                                      None, None, None)
        if rc != 0:
            self._check(rc, "rp_plan_levels (begin)")
        self._inflight = (inputs[0], want_best_states)

    def last_level(self) -> int:
        return int(self._lib.rp_last_level(self._h))

    def plan_begin(self, inp: PlanInputs, cand_begin: int = 0, cand_end: int = -1, want_best_states: bool = True):
        """First half of ``plan``: the kernels of the plan go onto the context's stream and the call returns; ``plan_wait``
        collects the result.  One plan in flight per context; ``inp`` must stay as it is until then."""
        g = inp.grids()
        rc = self._lib.rp_plan(self._h, C.byref(inp.params), C.byref(inp.cost), C.byref(g), cand_begin, cand_end,
                               PLAN_BEGIN | (PLAN_ROWS if want_best_states else 0), None, None)
        if rc != 0:
            self._check(rc, "rp_plan (begin)")
        self._inflight = (inp, want_best_states)

    def plan_wait(self) -> PlanOutput:
        inp, want = self._inflight
        self._inflight = None
        res = self._res
        best = np.empty((N_ARRAYS, inp.params.N + 1)) if want else None
        rc = self._lib.rp_plan_wait(self._h, C.byref(res), best.ctypes.data_as(_DP) if best is not None else None)
        if rc != 0:
            self._check(rc, "rp_plan_wait")
        self._N = inp.params.N
        self._last_count = res.n_candidates
        return self._output(res, best)

    def _output(self, res: RpResult, best) -> PlanOutput:
        """``res`` (== ``self._res``) and ``best`` stay with the context as the raw result of call number ``serial``."""
        self._last_best = best
        self._serial += 1
        out = PlanOutput.from_c(res, best)
        out.serial = self._serial
        return out

    def coeffs_arena(self, cap: int):
        """``rp_coeffs_arena``: (lon_coeffs [cap, 6], lat_coeffs [cap, 6], traj_len [cap]) as NumPy views of pinned host memory the
        context owns.  Candidates written into them (leading rows) and passed to ``plan_coeffs`` as they are go to the device
        without the copy into the staging buffer.  Views handed out earlier die with a call that asks for more room."""
        fn = getattr(self._lib, "rp_coeffs_arena", None)
        if fn is None:
            raise RpError("this library build has no rp_coeffs_arena")
        have = getattr(self, "_arena", None)
        if have is not None and have[0] >= cap:
            return have[1]
        lon, lat, tl = _DP(), _DP(), C.POINTER(C.c_int32)()
        self._check(fn(self._h, int(cap), C.byref(lon), C.byref(lat), C.byref(tl)), "rp_coeffs_arena")
        arrays = (np.ctypeslib.as_array(lon, shape=(cap, 6)), np.ctypeslib.as_array(lat, shape=(cap, 6)),
                  np.ctypeslib.as_array(tl, shape=(cap,)))
        gfn = getattr(self._lib, "rp_coeffs_arena_groups", None)
        if gfn is not None:   # + (group [cap], group_first [cap]) for rp_corridor_coeffs_grouped / plan_coeffs(groups=...)
            g, gf = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
            self._check(gfn(self._h, C.byref(g), C.byref(gf)), "rp_coeffs_arena_groups")
            arrays = arrays + (np.ctypeslib.as_array(g, shape=(cap,)), np.ctypeslib.as_array(gf, shape=(cap,)))
        self._arena = (int(cap), arrays)
        return arrays

    def plan_coeffs(self, params: RpParams, cost: RpCost, lon_coeffs, lat_coeffs, lon_T, traj_len,
                    want_best_states: bool = True, groups=None) -> PlanOutput:
        """``groups``: (n_groups, group [C], group_first [n_groups]) -- candidates that share their longitudinal polynomial, as
        ``rp_corridor_coeffs_grouped`` reports them; arena arrays only (``rp_plan_coeffs`` with ``n_groups`` > 0)."""
        lon_coeffs, lat_coeffs, lon_T = f64(lon_coeffs), f64(lat_coeffs), f64(lon_T)
        traj_len = np.ascontiguousarray(traj_len, dtype=np.int32)
        cnt = len(traj_len)
        assert lon_coeffs.shape == (cnt, 6) and lat_coeffs.shape == (cnt, 6) and lon_T.shape == (cnt,)
        res = self._res
        best = np.empty((N_ARRAYS, params.N + 1)) if want_best_states else None
        ipt = C.POINTER(C.c_int32)
        grouped = groups is not None and groups[0] > 0
        self._check(self._lib.rp_plan_coeffs(self._h, C.byref(params), C.byref(cost), cnt, dptr(lon_coeffs), dptr(lat_coeffs), dptr(lon_T),
                                             traj_len.ctypes.data_as(ipt), int(groups[0]) if grouped else 0,
                                             groups[1].ctypes.data_as(ipt) if grouped else None, groups[2].ctypes.data_as(ipt) if grouped else None,
                                             C.byref(res), dptr(best)), "rp_plan_coeffs")
        self._N = params.N
        self._last_count = cnt
        return self._output(res, best)

    def fetch_status(self, first: int = 0, count: Optional[int] = None):
        count = self._last_count - first if count is None else count
        status = np.empty(count, dtype=np.uint32)
        cost = np.empty(count, dtype=np.float64)
        self._check(self._lib.rp_fetch_status(self._h, first, count, status.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              dptr(cost)), "rp_fetch_status")
        return status, cost

    def fetch_states(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self._last_count - first if count is None else count
        out = np.empty((count, N_ARRAYS, self._N + 1))
        self._check(self._lib.rp_fetch_states(self._h, first, count, dptr(out)), "rp_fetch_states")
        return out

    def eval_one(self, index: int):
        out = np.empty((N_ARRAYS, self._N + 1))
        st = C.c_uint32()
        cost = C.c_double()
        self._check(self._lib.rp_eval_one(self._h, int(index), dptr(out), C.byref(st), C.byref(cost)), "rp_eval_one")
        self._serial += 1   # (the device result block now holds this candidate's rows, not the last plan's winner)
        return out, int(st.value), float(cost.value)

    def count_collisions_before(self, cost: float, index: int) -> int:
        n = C.c_int64()
        self._check(self._lib.rp_count_collisions_before(self._h, float(cost), int(index), C.byref(n)),
                    "rp_count_collisions_before")
        return int(n.value)

    def cost_range(self):
        """(min, max, n) of the costs of the kinematically feasible candidates of the last plan (NaN, NaN, 0 if none)."""
        lo, hi, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self._lib.rp_cost_range(self._h, C.byref(lo), C.byref(hi), C.byref(n)), "rp_cost_range")
        return float(lo.value), float(hi.value), int(n.value)

    def check_swept(self, params: RpParams, x, y, theta, want_boxes: bool = False):
        """Continuous collision check of one trajectory (reactive_planner.py:1049-1058): first colliding segment
        or -1 (with ``want_boxes``: also the segments' rectangles, [n - 1][6])."""
        x, y, theta = f64(x), f64(y), f64(theta)
        if not (len(x) == len(y) == len(theta)):
            raise ValueError("check_swept: x, y, theta differ in length")
        first = C.c_int32(-1)
        boxes = np.empty((max(len(x) - 1, 0), 6)) if want_boxes else None
        self._check(self._lib.rp_check_swept(self._h, C.byref(params), len(x), dptr(x), dptr(y), dptr(theta),
                                             C.byref(first), dptr(boxes)), "rp_check_swept")
        return (int(first.value), boxes) if want_boxes else int(first.value)

    def result_device(self):
        """(device address, bytes, rows_valid) of the result block the last plan left in device memory: header + winner
        state rows, what the device-side winner exchange gathers."""
        ptr, nbytes, ok = C.c_void_p(), C.c_size_t(), C.c_int32()
        self._check(self._lib.rp_result_device(self._h, C.byref(ptr), C.byref(nbytes), C.byref(ok)), "rp_result_device")
        return int(ptr.value or 0), int(nbytes.value), bool(ok.value)

    def combine_results(self, d_msgs_ptr: int, world: int, stream: int = 0):
        """Global result from the gathered per-rank result blocks (device memory, [world][bytes]): (PlanOutput, owner rank,
        rows_valid -- False: the owner's state rows were on its host only, exchange host-packed messages for this step)."""
        res = RpResult()
        best = np.empty((N_ARRAYS, self._N + 1))
        owner, rows_ok = C.c_int32(-1), C.c_int32(1)
        self._check(self._lib.rp_combine_results(self._h, C.c_void_p(d_msgs_ptr), int(world), C.c_void_p(stream or None),
                                                 C.byref(res), dptr(best), C.byref(owner), C.byref(rows_ok)), "rp_combine_results")
        return PlanOutput.from_c(res, best), int(owner.value), bool(rows_ok.value)

    def select(self, costs, want_best_states: bool = True) -> PlanOutput:
        costs = f64(costs)
        res = self._res
        best = np.empty((N_ARRAYS, self._N + 1)) if want_best_states else None
        self._check(self._lib.rp_select(self._h, dptr(costs), len(costs), C.byref(res), dptr(best)), "rp_select")
        return self._output(res, best)


# ---------------------------------------------------------------------------------------------------------------------
HOSTLOOP_PATH = os.path.join(os.path.dirname(LIB_PATH), "librp_hostloop.so")


class RpHostLoopStats(C.Structure):
    _fields_ = [("feasible_sum", C.c_int64), ("winners", C.c_int64), ("paths", C.c_int64 * 4), ("kernel_ms_sum", C.c_double),
                ("kernel_ms_n", C.c_int64)]


class HostLoop:
    """A compiled caller's plan loop (csrc/rp_hostloop.c): ``rp_plan`` on a cycle of prepared inputs, ``steps`` calls per
    ``run`` with no Python between them.  Timing harnesses use it; the planner does not."""

    def __init__(self, ctx: "RpContext", inputs, ranges=None):
        if not os.path.exists(HOSTLOOP_PATH):
            raise RpError(f"{HOSTLOOP_PATH} is missing: build it (python -c 'import __graft_entry__ as g; g.build()')")
        self._loop = C.CDLL(HOSTLOOP_PATH)
        self._loop.rp_hostloop_run.restype = C.c_int
        self._ctx, self._inputs = ctx, list(inputs)     # (the inputs own the arrays the structs point into)
        n = self.n = len(self._inputs)
        self._grids = [q.grids() for q in self._inputs]
        self._params = (C.c_void_p * n)(*[C.addressof(q.params) for q in self._inputs])
        self._cost = (C.c_void_p * n)(*[C.addressof(q.cost) for q in self._inputs])
        self._g = (C.c_void_p * n)(*[C.addressof(g) for g in self._grids])
        self._lo = self._hi = None
        if ranges is not None:
            self._lo = (C.c_int64 * n)(*[int(r[0]) for r in ranges])
            self._hi = (C.c_int64 * n)(*[int(r[1]) for r in ranges])
        n_max = max(q.params.N for q in self._inputs) + 1
        self._best = np.empty((N_ARRAYS, n_max))
        self._plan = C.cast(ctx._lib.rp_plan, C.c_void_p)
        lp = getattr(ctx._lib, "rp_last_path", None)
        self._last_path = C.cast(lp, C.c_void_p) if lp is not None else None

    def run(self, k0: int, steps: int) -> RpHostLoopStats:
        st = RpHostLoopStats()
        ctx = self._ctx
        rc = self._loop.rp_hostloop_run(self._plan, self._last_path, ctx._h, C.c_int32(self.n), self._params, self._cost, self._g,
                                        self._lo, self._hi, C.c_int64(k0), C.c_int64(steps), C.byref(ctx._res),
                                        self._best.ctypes.data_as(_DP), C.byref(st))
        if rc != 0:
            ctx._check(rc, "rp_plan")
        last = self._inputs[(k0 + steps - 1) % self.n]
        self._last_n = last.params.N + 1
        ctx._N = last.params.N
        ctx._last_count = ctx._res.n_candidates
        ctx._serial += 1
        return st

    def run_sharded(self, mailbox, k0: int, steps: int):
        """The same on a rank of a sharded group: every step ``rp_plan`` on this rank's ranges (``ranges`` of the constructor)
        followed by the winner exchange through ``mailbox`` (a ``distributed.MailboxExchange``; both messages, the rules of
        ``distributed.local_collisions_before``) -- (stats, seconds spent in the exchanges).  The global result of the last step
        is left in ``mailbox._glob`` / ``mailbox._states``."""
        if self._lo is None:
            raise RpError("HostLoop.run_sharded: constructed without candidate ranges")
        if mailbox._broken:
            raise RuntimeError(f"MailboxExchange: unusable after a time-out ({mailbox._broken})")
        st = RpHostLoopStats()
        ctx, lib = self._ctx, self._ctx._lib
        fn = self._loop.rp_hostloop_run_sharded
        fn.restype = C.c_int
        seq, t_ex = C.c_uint64(mailbox.seq), C.c_double(0.0)
        cast = lambda f: C.cast(f, C.c_void_p)   # noqa: E731
        rc = fn(self._plan, self._last_path, cast(lib.rp_mailbox_exchange), cast(lib.rp_mailbox_sum), cast(lib.rp_count_collisions_before),
                ctx._h, mailbox._region, C.c_int32(mailbox.world), C.c_int32(mailbox.rank), C.byref(seq), C.c_int32(self.n),
                self._params, self._cost, self._g, self._lo, self._hi, C.c_int64(k0), C.c_int64(steps), C.byref(ctx._res),
                self._best.ctypes.data_as(_DP), C.byref(mailbox._glob), mailbox._states.ctypes.data_as(_DP), C.byref(st), C.byref(t_ex))
        mailbox.seq = int(seq.value)
        if rc <= -1000:
            raise mailbox._timeout("rp_hostloop_run_sharded (mailbox)", rc + 1000, "its message")
        if rc != 0:
            ctx._check(rc, "rp_plan")
        last = self._inputs[(k0 + steps - 1) % self.n]
        self._last_n = last.params.N + 1
        ctx._N = last.params.N
        ctx._last_count = ctx._res.n_candidates
        ctx._serial += 1
        return st, float(t_ex.value)

    def best_states(self) -> np.ndarray:
        """state rows [14, N + 1] of the last step's winner (meaningful only if it had one)"""
        n = self._last_n
        return self._best.ravel()[:N_ARRAYS * n].reshape(N_ARRAYS, n)
