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

__all__ = ["ObstacleTables", "obb_sum", "obb_sum_rows", "triangulate_polygon_fan", "road_boundary_obb", "SimpleLanelet", "lanelets_from_arrays"]


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
    def from_scenario(cls, scenario, road_boundary_triangles: Optional[np.ndarray] = None,
                      road_boundary: bool = False, continuous: bool = False) -> "ObstacleTables":
        """Extract tables from a CommonRoad ``Scenario`` (duck-typed: ``static_obstacles`` /
        ``dynamic_obstacles`` with ``obstacle_shape``, ``initial_state`` and
        ``prediction.trajectory.state_list``), mirroring what ``set_collision_checker`` feeds to
        pycrcc (reactive_planner.py:234-251).  Rectangles and circles are supported directly,
        polygons are fan-triangulated (convex polygons only).  ``road_boundary=True`` adds the
        rectangles of ``road_boundary_obb(scenario.lanelet_network.lanelets)`` (the reference calls
        ``create_road_boundary_obstacle(scenario)``, reactive_planner.py:246-248).
        ``continuous=True`` (``planning.continuous_collision_check``): every dynamic obstacle is replaced by
        ``trajectory_preprocess_obb_sum`` of itself, as the reference does before adding it to the checker
        (reactive_planner.py:238-245) -- the rectangle of time step k becomes the tight rectangle around the
        obstacle's rectangles of steps k and k + 1 (``obb_sum``), the obstacle's last step has no shape.  Both the
        per-pose test and the swept test of the ego vehicle then run against these."""
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
        if continuous:
            tab = obb_sum_rows(tab)
        if road_boundary_triangles is not None:
            stri.extend(np.asarray(road_boundary_triangles, dtype=np.float64).reshape(-1, 6).tolist())
        if road_boundary:
            sobb.extend(road_boundary_obb(scenario.lanelet_network.lanelets).tolist())
        return cls(static_obb=sobb or None, static_tri=stri or None, static_circ=scirc or None,
                   dyn_obb=tab, dyn_t0=t0)


def obb_sum(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Tight rectangle around rectangles ``a`` and ``b`` (rows cx, cy, theta, half_length, half_width; vectorised
    over leading axes): what commonroad-dc's ``trajectory_preprocess_obb_sum`` puts in place of two consecutive
    shapes.  commonroad-dc is not under /root/reference; the definition used by this build -- the planar form of
    FCL's OBB sum, identical in csrc/rp_device.h (``merge_swept``), oracle/rp_oracle.c and the fixtures' stand-in --
    is: orientation = bisector of the two headings (the second one flipped when they point apart), or the line
    through the centres when those are farther apart than twice the sum of the larger half extents; extents = both
    rectangles projected on the two axes."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    aux, auy, bux, buy = np.cos(a[..., 2]), np.sin(a[..., 2]), np.cos(b[..., 2]), np.sin(b[..., 2])
    dx, dy = b[..., 0] - a[..., 0], b[..., 1] - a[..., 1]
    lim = 2.0 * (np.maximum(a[..., 3], a[..., 4]) + np.maximum(b[..., 3], b[..., 4]))
    far = dx * dx + dy * dy > lim * lim
    sgn = np.where(aux * bux + auy * buy < 0.0, -1.0, 1.0)
    nx, ny = np.where(far, dx, aux + sgn * bux), np.where(far, dy, auy + sgn * buy)
    nrm = np.sqrt(nx * nx + ny * ny)
    nx, ny = nx / nrm, ny / nrm
    mx, my = -ny, nx

    def extent(ex, ey):
        ea = a[..., 3] * np.abs(aux * ex + auy * ey) + a[..., 4] * np.abs(auy * ex - aux * ey)
        eb = b[..., 3] * np.abs(bux * ex + buy * ey) + b[..., 4] * np.abs(buy * ex - bux * ey)
        pb = dx * ex + dy * ey
        return np.minimum(-ea, pb - eb), np.maximum(ea, pb + eb)

    lo0, hi0 = extent(nx, ny)
    lo1, hi1 = extent(mx, my)
    c0, c1 = 0.5 * (lo0 + hi0), 0.5 * (lo1 + hi1)
    return np.stack((a[..., 0] + (c0 * nx + c1 * mx), a[..., 1] + (c0 * ny + c1 * my), np.arctan2(ny, nx),
                     0.5 * (hi0 - lo0), 0.5 * (hi1 - lo1)), axis=-1)


def obb_sum_rows(dyn_obb: np.ndarray) -> np.ndarray:
    """``trajectory_preprocess_obb_sum`` applied to every row of a (n_dyn, n_steps, 5) table: step k holds the sum of
    the obstacle's rectangles of steps k and k + 1; where either is absent (NaN) -- in particular at the obstacle's
    last step -- there is no shape (the reference's object is one shape shorter, reactive_planner.py:241)."""
    dyn = np.asarray(dyn_obb, dtype=np.float64)
    out = np.full_like(dyn, np.nan)
    if dyn.shape[0] and dyn.shape[1] > 1:
        both = ~np.isnan(dyn[:, :-1, 0]) & ~np.isnan(dyn[:, 1:, 0])
        with np.errstate(invalid="ignore"):
            merged = obb_sum(dyn[:, :-1], dyn[:, 1:])
        out[:, :-1][both] = merged[both]
    return out


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


# ---------------------------------------------------------------------------------------------------
# Road boundary.  The reference delegates to commonroad_dc.boundary.create_road_boundary_obstacle
# (third party, not under /root/reference; reactive_planner.py:29,246-248).  This build generates the
# boundary as thin rectangles along the outer borders of the lanelet network -- the flavour
# commonroad-dc calls "obb_rectangles": a continuous trajectory that starts on the road cannot leave
# it without overlapping one of them.
# ---------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class SimpleLanelet:
    """The fields of a CommonRoad ``Lanelet`` the generator reads (same attribute names)."""
    lanelet_id: int
    left_vertices: np.ndarray
    right_vertices: np.ndarray
    adj_left: Optional[int] = None
    adj_right: Optional[int] = None
    predecessor: Sequence[int] = ()
    successor: Sequence[int] = ()


def lanelets_from_arrays(ids, left, right, offsets, flags) -> list:
    """Lanelets from the flat arrays of the scenario fixtures (tests/golden/make_scenarios.py):
    ``flags[k] = (has adjacent left, has adjacent right, has predecessor, has successor)``."""
    out = []
    for k, lid in enumerate(ids):
        a, b = int(offsets[k]), int(offsets[k + 1])
        f = flags[k]
        out.append(SimpleLanelet(int(lid), np.asarray(left[a:b], dtype=float), np.asarray(right[a:b], dtype=float),
                                 adj_left=-1 if f[0] else None, adj_right=-1 if f[1] else None,
                                 predecessor=(-1,) if f[2] else (), successor=(-1,) if f[3] else ()))
    return out


def _points_in_polygon(pts: np.ndarray, poly: np.ndarray) -> np.ndarray:
    """Even-odd rule, points strictly inside (vectorised over points)."""
    x, y = pts[:, 0][:, None], pts[:, 1][:, None]
    x0, y0 = poly[:, 0][None, :], poly[:, 1][None, :]
    x1, y1 = np.roll(poly[:, 0], -1)[None, :], np.roll(poly[:, 1], -1)[None, :]
    cross = (y0 > y) != (y1 > y)
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = x0 + (y - y0) * (x1 - x0) / (y1 - y0)
    return (np.sum(cross & (x < xi), axis=1) % 2) == 1


def road_boundary_obb(lanelets: Iterable, thickness: float = 0.1, inside_margin: float = 0.05,
                      simplify_tol: float = 0.02) -> np.ndarray:
    """Rectangles ``(cx, cy, theta, half_length, half_width)`` along the outer border of a lanelet network.

    Candidate segments: the left border of every lanelet without a left neighbour, the right border of every
    lanelet without a right neighbour, and the start / end edge of lanelets without predecessor / successor.
    A segment is dropped when its midpoint, moved ``inside_margin`` to either side, lies inside some lanelet on
    BOTH sides -- i.e. it runs through drivable area (overlapping lanelets of an intersection).  Chains of
    connected segments are simplified (Douglas-Peucker, ``simplify_tol`` metres; negative: keep every segment)."""
    lls = list(lanelets)
    polys = [np.concatenate((np.asarray(l.left_vertices, dtype=float), np.asarray(l.right_vertices, dtype=float)[::-1])) for l in lls]
    segs, owner = [], []
    for k, l in enumerate(lls):
        left, right = np.asarray(l.left_vertices, dtype=float), np.asarray(l.right_vertices, dtype=float)
        borders = []
        if getattr(l, "adj_left", None) is None:
            borders.append(left)
        if getattr(l, "adj_right", None) is None:
            borders.append(right)
        if not list(getattr(l, "predecessor", ()) or ()):
            borders.append(np.stack((left[0], right[0])))
        if not list(getattr(l, "successor", ()) or ()):
            borders.append(np.stack((left[-1], right[-1])))
        for b in borders:
            for p, q in zip(b[:-1], b[1:]):
                if np.hypot(*(q - p)) > 1e-9:
                    segs.append((p, q)); owner.append(k)
    if not segs:
        return np.zeros((0, 5))
    P = np.array([s[0] for s in segs]); Q = np.array([s[1] for s in segs]); owner = np.array(owner)
    mid = 0.5 * (P + Q)
    d = Q - P
    ln = np.hypot(d[:, 0], d[:, 1])
    nrm = np.stack((-d[:, 1], d[:, 0]), axis=1) / ln[:, None]
    inside_a = np.zeros(len(segs), dtype=bool); inside_b = np.zeros(len(segs), dtype=bool)
    for poly in polys:   # (the segment's own lanelet counts: its inner side is drivable area)
        inside_a |= _points_in_polygon(mid + inside_margin * nrm, poly)
        inside_b |= _points_in_polygon(mid - inside_margin * nrm, poly)
    keep = ~(inside_a & inside_b)
    # chains of connected kept segments, each simplified (Douglas-Peucker): straight borders are often given in 1 m
    # pieces, curved ones in centimetre pieces; no vertex of the original border ends up farther than simplify_tol
    # from the simplified one (well inside the rectangles' thickness)
    chains, cur = [], None
    for k in np.flatnonzero(keep):
        p, q = P[k], Q[k]
        if cur is not None and np.allclose(cur[-1], p, atol=1e-9):
            cur.append(q)
        else:
            if cur is not None:
                chains.append(np.array(cur))
            cur = [p, q]
    if cur is not None:
        chains.append(np.array(cur))

    def simplify(pts):
        if len(pts) <= 2:
            return pts
        a, b = pts[0], pts[-1]
        ab = b - a
        L = np.hypot(*ab)
        rel = pts[1:-1] - a
        if L < 1e-12:
            dist = np.hypot(rel[:, 0], rel[:, 1])
        else:
            t = np.clip((rel @ ab) / (L * L), 0.0, 1.0)
            dist = np.hypot(*(rel - t[:, None] * ab).T)
        j = int(np.argmax(dist))
        if dist[j] <= simplify_tol:
            return np.stack((a, b))
        left, right = simplify(pts[:j + 2]), simplify(pts[j + 1:])
        return np.concatenate((left[:-1], right))

    out = []
    for ch in chains:
        sp = simplify(ch) if simplify_tol >= 0 else ch
        out.extend(zip(sp[:-1], sp[1:]))
    P2 = np.array([o[0] for o in out]).reshape(-1, 2); Q2 = np.array([o[1] for o in out]).reshape(-1, 2)
    d2 = Q2 - P2
    return np.stack((0.5 * (P2[:, 0] + Q2[:, 0]), 0.5 * (P2[:, 1] + Q2[:, 1]), np.arctan2(d2[:, 1], d2[:, 0]),
                     0.5 * np.hypot(d2[:, 0], d2[:, 1]), np.full(len(out), 0.5 * thickness)), axis=1)
