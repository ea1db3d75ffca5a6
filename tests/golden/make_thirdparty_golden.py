#!/usr/bin/env python3
"""Pins for the THIRD-PARTY geometry the hot path touches (DESIGN.md section 2 "unpinned third-party boundaries").

The reference's hot path calls into packages whose source is not under /root/reference and which are NOT installed in the build
image of this repository (no wheels, no network):

  * ``pycrccosy.CurvilinearCoordinateSystem.convert_to_cartesian_coords`` / ``convert_to_curvilinear_coords`` and its
    ``ref_pos / ref_theta / ref_curv / ref_curv_d`` tables  (commonroad-drivability-checker 2024.1;
    commonroad_rp/utility/utils_coordinate_system.py:114-118,128-129,167-178; reactive_planner.py:908-917)
  * ``pycrcc`` overlap tests: ``CollisionChecker.collide`` on a ``TimeVariantCollisionObject`` of ``RectOBB`` shapes against
    rectangles, triangles, circles and time-variant obstacles  (reactive_planner.py:234-251,1033-1046)
  * ``commonroad_dc.boundary.boundary.create_road_boundary_obstacle``  (reactive_planner.py:246-250)
  * ``commonroad_dc.collision.trajectory_queries.trajectory_preprocess_obb_sum``  (reactive_planner.py:1049-1052)

This build DEFINES those semantics once (commonroad_rp_amd/coordinate_system.py, collision.py, oracle/rp_oracle.c, csrc/rp_device.h)
and uses the same definition on both sides of every parity test.  THIS SCRIPT CANNOT RUN IN THE BUILD IMAGE -- it needs the real
wheels -- and nothing in this repository tries to import them there.  A maintainer who has them runs

    pip install commonroad-io==2024.1 commonroad-drivability-checker==2024.1
    python tests/golden/make_thirdparty_golden.py [--reference-root /path/to/commonroad-reactive-planner]

which writes ``tests/golden/thirdparty_*.npz`` -- DATA only: the inputs (taken from the fixtures this repository already holds) and
what the real packages answered.  ``tests/test_thirdparty_pins.py`` then compares the oracle and the HIP path with them and SKIPS,
visibly, while the files are absent.  Commit the files; they are the pin.

Outputs (one file each per fixture / scenario):
  thirdparty_cosy_<fixture>.npz      ref_path, proj_d_limit | the package's tables | (s, d) samples -> (x, y) | (x, y) -> (s, d)
  thirdparty_collide_<fixture>.npz   the fixture's obstacle tables and vehicle | per sampled candidate: poses [n][3] and the package's
                                     ``cc.collide(ego)``, per pose the time-slice verdict
  thirdparty_obbsum_<fixture>.npz    per sampled candidate: the rows of ``trajectory_preprocess_obb_sum`` (centre, axes, half extents)
  thirdparty_boundary_<scenario>.npz the scenario's lanelet polygons | the triangles of ``create_road_boundary_obstacle``
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["cfg1_ref_l3", "cfg2_ref", "cfg3_ref", "cfg2_ref_rb", "cfg3_ref_rb"]
SCENARIOS = ["ZAM_Over-1_1", "ZAM_Tjunction-1_42_T-1", "DEU_Test-1_1_T-1"]


def need(mod):
    try:
        return __import__(mod, fromlist=["_"])
    except ImportError as e:
        sys.exit(f"{mod} is not installed ({e}).  This generator needs the real third-party wheels (see the header); "
                 f"it is not meant to run in the build image of this repository.")


def cosy_pins(name, z, pycrccosy):
    ref = np.asarray(z["ref_path"], dtype=float)
    limit = float(z["proj_d_limit"])
    cs = pycrccosy.CurvilinearCoordinateSystem(ref, limit, 0.1)   # (utils_coordinate_system.py:88-118 builds it with these defaults)
    pos = np.asarray(cs.ref_pos() if callable(getattr(cs, "ref_pos", None)) else cs.ref_pos, dtype=float)
    rng = np.random.default_rng(7)
    s = rng.uniform(pos[1], pos[-2], 4000)
    d = rng.uniform(-min(limit, 6.0), min(limit, 6.0), 4000)
    xy = np.array([cs.convert_to_cartesian_coords(float(a), float(b)) for a, b in zip(s, d)])
    back = np.array([cs.convert_to_curvilinear_coords(float(x), float(y)) for x, y in xy])
    tab = {k: np.asarray(getattr(cs, k)() if callable(getattr(cs, k)) else getattr(cs, k), dtype=float)
           for k in ("ref_pos", "ref_theta", "ref_curv", "ref_curv_d") if hasattr(cs, k)}
    np.savez_compressed(os.path.join(HERE, f"thirdparty_cosy_{name}.npz"), ref_path=ref, proj_d_limit=limit, s=s, d=d, xy=xy, sd_back=back,
                        reference=np.asarray(cs.reference_path(), dtype=float), **tab)


def checker_of(z, pycrcc):
    cc = pycrcc.CollisionChecker()
    for cx, cy, th, hl, hw in np.asarray(z["static_obb"], dtype=float).reshape(-1, 5):
        cc.add_collision_object(pycrcc.RectOBB(hl, hw, th, cx, cy))
    for t in np.asarray(z["static_tri"], dtype=float).reshape(-1, 6):
        cc.add_collision_object(pycrcc.Triangle(*[float(v) for v in t]))
    for cx, cy, r in np.asarray(z["static_circ"], dtype=float).reshape(-1, 3):
        cc.add_collision_object(pycrcc.Circle(r, cx, cy))
    dyn = np.asarray(z["dyn_obb"], dtype=float)
    t0 = int(z["dyn_t0"])
    for j in range(dyn.shape[0]):
        present = np.flatnonzero(~np.isnan(dyn[j, :, 0]))
        if len(present) == 0:
            continue
        k0, k1 = int(present[0]), int(present[-1])
        tvo = pycrcc.TimeVariantCollisionObject(t0 + k0)
        for k in range(k0, k1 + 1):
            cx, cy, th, hl, hw = dyn[j, k]
            tvo.append_obstacle(pycrcc.RectOBB(hl, hw, th, cx, cy))
        cc.add_collision_object(tvo)
    return cc


def collide_pins(name, z, pycrcc, tq):
    cc = checker_of(z, pycrcc)
    veh = {k: float(v) for k, v in zip(("length", "width", "wb_rear_axle"), np.asarray(z["vehicle"], dtype=float)[:3])} \
        if np.asarray(z["vehicle"]).dtype.kind == "f" else None
    if veh is None:
        raise SystemExit(f"{name}: unexpected `vehicle` entry")
    hl, hw, wb = 0.5 * veh["length"], 0.5 * veh["width"], veh["wb_rear_axle"]
    states = np.asarray(z["states"], dtype=float)          # [64][14][N + 1]
    t0, factor = int(z["time_step0"]), int(z["factor"])
    verdict, per_pose, poses, sums = [], [], [], []
    for blk in states:
        x, y, th = blk[0], blk[1], blk[2]
        ego = pycrcc.TimeVariantCollisionObject(t0 * factor)   # reactive_planner.py:1033-1042: one RectOBB per state, centre shifted to the vehicle's
        hits = []
        for i in range(len(x)):
            cx, cy = x[i] + wb * np.cos(th[i]), y[i] + wb * np.sin(th[i])
            rect = pycrcc.RectOBB(hl, hw, th[i], cx, cy)
            ego.append_obstacle(rect)
            hits.append(bool(cc.time_slice(t0 + i * factor).collide(rect)))
        verdict.append(bool(cc.collide(ego)))
        per_pose.append(hits)
        poses.append(np.stack((x, y, th), axis=1))
        pre, err = tq.trajectory_preprocess_obb_sum(ego)
        rows = []
        if not err:
            for i in range(len(x) - 1):
                o = pre.obstacle_at_time(t0 * factor + i)
                c = o.center()
                rows.append([c[0], c[1], o.local_x_axis()[0], o.local_x_axis()[1], o.r_x(), o.r_y()])
        sums.append(rows)
    np.savez_compressed(os.path.join(HERE, f"thirdparty_collide_{name}.npz"), fixture=name, state_index=z["state_index"], poses=np.array(poses),
                        collide=np.array(verdict), collide_per_pose=np.array(per_pose), half_length=hl, half_width=hw, wb_rear_axle=wb)
    np.savez_compressed(os.path.join(HERE, f"thirdparty_obbsum_{name}.npz"), fixture=name, poses=np.array(poses), boxes=np.array(sums, dtype=float),
                        half_length=hl, half_width=hw, wb_rear_axle=wb)


def boundary_pins(scen, root):
    from commonroad.common.file_reader import CommonRoadFileReader
    from commonroad_dc.boundary.boundary import create_road_boundary_obstacle
    path = os.path.join(root, "example_scenarios", scen + ".xml")
    scenario, _ = CommonRoadFileReader(path).open()
    _, obj = create_road_boundary_obstacle(scenario)   # (the reference's call, reactive_planner.py:246-250: default method)
    tris = []
    for sh in obj.unpack():
        v = sh.vertices()
        tris.append([v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1]])
    lls = scenario.lanelet_network.lanelets
    np.savez_compressed(os.path.join(HERE, f"thirdparty_boundary_{scen}.npz"), triangles=np.array(tris, dtype=float),
                        ll_ids=np.array([l.lanelet_id for l in lls]), ll_left=np.concatenate([l.left_vertices for l in lls]),
                        ll_right=np.concatenate([l.right_vertices for l in lls]),
                        ll_offsets=np.cumsum([0] + [len(l.left_vertices) for l in lls]))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference-root", default="/root/reference", help="checkout of commonroad-reactive-planner (for its example_scenarios/*.xml)")
    args = ap.parse_args()
    pycrcc = need("commonroad_dc.pycrcc")
    pycrccosy = need("commonroad_dc.pycrccosy")
    tq = need("commonroad_dc.collision.trajectory_queries.trajectory_queries")
    for name in FIXTURES:
        f = os.path.join(HERE, name + ".npz")
        if not os.path.exists(f):
            print("skip", name, "(fixture absent)")
            continue
        z = dict(np.load(f, allow_pickle=False))
        cosy_pins(name, z, pycrccosy)
        collide_pins(name, z, pycrcc, tq)
        print("wrote thirdparty_{cosy,collide,obbsum}_" + name + ".npz")
    for scen in SCENARIOS:
        if os.path.exists(os.path.join(args.reference_root, "example_scenarios", scen + ".xml")):
            boundary_pins(scen, args.reference_root)
            print("wrote thirdparty_boundary_" + scen + ".npz")
    print(len(glob.glob(os.path.join(HERE, "thirdparty_*.npz"))), "files under", HERE)


if __name__ == "__main__":
    main()
