"""Obstacle tables handed to the HIP collision kernel.

The reference keeps obstacles inside an opaque ``pycrcc.CollisionChecker`` built by
``ReactivePlanner.set_collision_checker`` (reference: commonroad_rp/reactive_planner.py:218-256)
and queries it one ego pose at a time (``_check_collisions``, reactive_planner.py:1019-1063).
The GPU path needs the same content as flat arrays, extracted once per ``reset``:

* ``static_obb``  (n, 5)   cx, cy, theta, half_length, half_width        -- static rectangles
* ``static_tri``  (n, 6)   x1, y1, x2, y2, x3, y3                        -- road-boundary / polygon triangles
* ``static_circ`` (n, 3)   cx, cy, r
* ``dyn_obb``     (n_dyn, n_steps, 5) one OBB per dynamic obstacle and scenario time step
  ``dyn_t0 + j``; rows whose cx is NaN mean "obstacle absent at that step".

Collision semantics (defined by this build, pycrcc's source is not under /root/reference):
closed-set overlap, i.e. touching shapes collide; separating-axis tests, see csrc/rp_device.h.
"""
from __future__ import annotations

import dataclasses
from typing import Iterable, Optional, Sequence

import numpy as np

__all__ = ["ObstacleTables", "triangulate_polygon_fan"]


def _arr(a, cols: int) -> np.ndarray:
    a = np.asarray(a if a is not None else np.zeros((0, cols)), dtype=np.float64)
    if a.size == 0:
        a = a.reshape(0, cols)
    assert a.ndim == 2 and a.shape[1] == cols, f"expected (n, {cols}) array, got {a.shape}"
    return np.ascontiguousarray(a)


@dataclasses.dataclass
class ObstacleTables:
    static_obb: np.ndarray = None
    static_tri: np.ndarray = None
    static_circ: np.ndarray = None
    dyn_obb: np.ndarray = None
    dyn_t0: int = 0

    def __post_init__(self):
        self.static_obb = _arr(self.static_obb, 5)
        self.static_tri = _arr(self.static_tri, 6)
        self.static_circ = _arr(self.static_circ, 3)
        d = np.asarray(self.dyn_obb if self.dyn_obb is not None else np.zeros((0, 0, 5)), dtype=np.float64)
        if d.size == 0:
            d = d.reshape(0, 0, 5)
        assert d.ndim == 3 and d.shape[2] == 5, f"dyn_obb must be (n_dyn, n_steps, 5), got {d.shape}"
        self.dyn_obb = np.ascontiguousarray(d)
        self.dyn_t0 = int(self.dyn_t0)

    @property
    def empty(self) -> bool:
        return (len(self.static_obb) + len(self.static_tri) + len(self.static_circ) == 0
                and self.dyn_obb.shape[0] * self.dyn_obb.shape[1] == 0)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_scenario(cls, scenario, road_boundary_triangles: Optional[np.ndarray] = None) -> "ObstacleTables":
        """Extract tables from a CommonRoad ``Scenario`` (duck-typed: ``static_obstacles`` /
        ``dynamic_obstacles`` with ``obstacle_shape``, ``initial_state`` and
        ``prediction.trajectory.state_list``), mirroring what ``set_collision_checker`` feeds to
        pycrcc (reactive_planner.py:234-251).  Rectangles and circles are supported directly,
        polygons are fan-triangulated (convex polygons only)."""
        sobb, stri, scirc = [], [], []
        for ob in getattr(scenario, "static_obstacles", []):
            st = ob.initial_state
            _add_shape(ob.obstacle_shape, st.position, getattr(st, "orientation", 0.0), sobb, stri, scirc)
        dyn = list(getattr(scenario, "dynamic_obstacles", []))
        t0, t1 = 0, -1
        if dyn:
            t0 = min(int(ob.initial_state.time_step) for ob in dyn)
            t1 = t0
            for ob in dyn:
                states = ob.prediction.trajectory.state_list if ob.prediction is not None else []
                t1 = max([t1] + [int(s.time_step) for s in states])
        tab = np.full((len(dyn), t1 - t0 + 1 if dyn else 0, 5), np.nan)
        for j, ob in enumerate(dyn):
            shape = ob.obstacle_shape
            assert hasattr(shape, "length") and hasattr(shape, "width"), \
                "dynamic obstacles must be rectangles for the GPU collision tables"
            states = [ob.initial_state] + (list(ob.prediction.trajectory.state_list) if ob.prediction else [])
            for s in states:
                tab[j, int(s.time_step) - t0] = (s.position[0], s.position[1], s.orientation,
                                                  0.5 * shape.length, 0.5 * shape.width)
        if road_boundary_triangles is not None:
            stri.extend(np.asarray(road_boundary_triangles, dtype=np.float64).reshape(-1, 6).tolist())
        return cls(static_obb=sobb or None, static_tri=stri or None, static_circ=scirc or None,
                   dyn_obb=tab, dyn_t0=t0)


def triangulate_polygon_fan(vertices: Sequence[Sequence[float]]) -> np.ndarray:
    """Fan triangulation of a convex polygon -> (n-2, 6)."""
    v = np.asarray(vertices, dtype=np.float64)
    if len(v) >= 2 and np.all(v[0] == v[-1]):
        v = v[:-1]
    return np.array([[*v[0], *v[i], *v[i + 1]] for i in range(1, len(v) - 1)], dtype=np.float64).reshape(-1, 6)


def _add_shape(shape, position, orientation, sobb, stri, scirc):
    if hasattr(shape, "length") and hasattr(shape, "width"):
        sobb.append([position[0], position[1], orientation, 0.5 * shape.length, 0.5 * shape.width])
    elif hasattr(shape, "radius"):
        scirc.append([position[0], position[1], shape.radius])
    elif hasattr(shape, "vertices"):
        c, s = np.cos(orientation), np.sin(orientation)
        v = np.asarray(shape.vertices, dtype=np.float64)
        w = np.stack((position[0] + c * v[:, 0] - s * v[:, 1], position[1] + s * v[:, 0] + c * v[:, 1]), axis=1)
        stri.extend(triangulate_polygon_fan(w).tolist())
    elif hasattr(shape, "shapes"):
        for sub in shape.shapes:
            _add_shape(sub, position, orientation, sobb, stri, scirc)
    else:
        raise TypeError(f"unsupported obstacle shape {type(shape).__name__}")
