#!/usr/bin/env python3
"""Golden vectors of the reference-path front end, produced by running the REFERENCE's own code -- unmodified, imported
from /root/reference -- under the third-party stand-ins of ``_ref_shims.py``:

  commonroad_rp.utility.utils_coordinate_system.CoordinateSystem.__init__   (:88-118, with smooth_ref_path :74-83 and the
      real scipy.interpolate.splprep / splev)
  commonroad_rp.reactive_planner.ReactivePlanner._compute_initial_states    (:446-512)

Runs only in the build container.  Output: tests/golden/frontend_<case>.npz -- data only: the input polyline, the
reference's tables, and for a list of Cartesian states the Frenet states (or which exception) the reference returns.
The polyline utilities and the projection behind the stand-ins are this build's definitions (oracle/frontend.py):
what the fixtures pin is everything the reference itself computes around them.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "commonroad-reactive-planner_amd"))
sys.path.insert(0, REPO)

import _ref_shims  # noqa: E402

_ref_shims.install()

from commonroad_rp.utility.utils_coordinate_system import CoordinateSystem  # noqa: E402  (reference)
from commonroad_rp.reactive_planner import ReactivePlanner  # noqa: E402  (reference)
from commonroad_rp.state import ReactivePlannerState  # noqa: E402

WHEELBASE = 1.1562 + 1.4227


def polylines():
    out = {}
    for name in ("ZAM_Over-1_1", "ZAM_Tjunction-1_42_T-1", "DEU_Test-1_1_T-1"):
        out[name.split("_")[0] + "_" + name.split("_")[1].split("-")[0]] = np.load(os.path.join(HERE, f"scenario_{name}.npz"))["centre"]
    s = np.arange(0.0, 90.0, 2.5)
    scurve = np.stack((s, 6.0 * np.sin(s / 14.0)), axis=1)
    out["scurve_dups"] = np.concatenate((scurve[:10], scurve[9:10], scurve[10:20], scurve[3:4], scurve[20:]))   # repeated vertices, one far back
    out["short5"] = np.array([[0.0, 0.0], [4.0, 0.5], [9.0, 2.0], [13.0, 5.0], [15.0, 9.0]])
    t = np.linspace(0.0, 1.6 * np.pi, 40)
    out["loop"] = np.stack((30.0 * np.cos(t) - 200.0, 30.0 * np.sin(t) + 75.0), axis=1)    # heading passes +-pi: unwrap
    return out


def states_for(co, rng):
    """Cartesian states on and beside the route, some pointing across it, one far away, one driving backwards"""
    ref, pos = np.asarray(co.reference), co.ref_pos
    out = []
    for k in range(10):
        s = rng.uniform(pos[1], pos[-2])
        i = int(np.searchsorted(pos, s)) - 1
        w = (s - pos[i]) / (pos[i + 1] - pos[i])
        p = ref[i] + w * (ref[i + 1] - ref[i])
        th = np.arctan2(*(ref[i + 1] - ref[i])[::-1])
        nrm = np.array([-np.sin(th), np.cos(th)])
        d = rng.normal(0.0, 1.2)
        out.append((p[0] + d * nrm[0], p[1] + d * nrm[1], th + rng.normal(0.0, 0.15), rng.uniform(0.0, 18.0), rng.normal(0.0, 1.0),
                    rng.normal(0.0, 0.05), int(rng.random() < 0.3)))
    out.append((ref[0][0] - 500.0, ref[0][1] + 500.0, 0.0, 5.0, 0.0, 0.0, 0))                       # outside the projection domain
    i = len(ref) // 2
    th = np.arctan2(*(ref[i + 1] - ref[i])[::-1])
    out.append((ref[i][0], ref[i][1], th + np.pi, 6.0, 0.0, 0.0, 0))                               # against the route
    out.append((ref[i][0], ref[i][1], th, 0.0, 0.0, 0.0, 1))                                       # standing still, low-velocity mode
    return out


def run_case(poly, smooth, rng):
    """the reference's CoordinateSystem.__init__ and _compute_initial_states on one polyline -> what a fixture holds"""
    co = CoordinateSystem(reference=np.array(poly, dtype=float), smooth_reference=smooth)      # the reference's __init__
    rp = object.__new__(ReactivePlanner)
    rp._co = co
    rp.vehicle_params = types.SimpleNamespace(wheelbase=WHEELBASE)
    states, lon, lat, code = [], [], [], []
    for (x, y, th, v, a, steer, low) in states_for(co, rng):
        rp._low_vel_mode = bool(low)
        x0 = ReactivePlannerState(time_step=0, position=np.array([x, y]), orientation=th, velocity=v, steering_angle=steer,
                                  acceleration=a, yaw_rate=0.0)
        try:
            l1, l2 = rp._compute_initial_states(x0)                                             # the reference's method
            c = 0
        except ValueError:
            l1, l2, c = [np.nan] * 3, [np.nan] * 3, -5
        except Exception:
            l1, l2, c = [np.nan] * 3, [np.nan] * 3, -6
        states.append((x, y, th, v, a, steer, low)); lon.append(l1); lat.append(l2); code.append(c)
    return dict(polyline=np.array(poly, dtype=float), smooth=int(smooth),
                reference=np.asarray(co.reference), ref_pos=co.ref_pos, ref_theta=co.ref_theta, ref_curv=co.ref_curv,
                ref_curv_d=co.ref_curv_d, wheelbase=WHEELBASE, states=np.array(states), x0_lon=np.array(lon, dtype=float),
                x0_lat=np.array(lat, dtype=float), code=np.array(code, dtype=np.int32))


def main():
    rng = np.random.default_rng(7)
    for name, poly in polylines().items():
        for smooth in (True, False):
            if not smooth and name not in ("scurve_dups", "short5"):
                continue
            out = run_case(poly, smooth, rng)
            tag = f"frontend_{name}" + ("" if smooth else "_raw")
            np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
            print(f"{tag:28s} in {len(poly):4d} -> {len(out['ref_pos']):4d} vertices, length {out['ref_pos'][-1]:8.2f} m, codes {sorted(set(out['code'].tolist()))}")


if __name__ == "__main__":
    main()
