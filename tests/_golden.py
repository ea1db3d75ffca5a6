"""Loader for the golden fixtures in tests/golden/*.npz (see tests/golden/make_golden.py)."""
from __future__ import annotations

import glob
import os

import numpy as np

from commonroad_rp_amd._capi import PlanInputs, make_params, make_cost, FLAG_DRAW_ALL
from commonroad_rp_amd.collision import ObstacleTables

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))) if not n.startswith("scenario_"))


class Golden:
    def __init__(self, name: str):
        self.name = name
        self.z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
        z = self.z
        veh = z["vehicle"]
        flags = FLAG_DRAW_ALL if int(z["draw"]) else 0
        params = make_params(dt=float(z["dt"]), N=int(z["N"]), factor=int(z["factor"]), time_step0=int(z["time_step0"]),
                             low_vel_mode=bool(z["low_vel_mode"]), lon_mode=int(z["lon_mode"]),
                             constraint_mask=int(z["constraint_mask"]), flags=flags, x0_lon=z["x0_lon"],
                             x0_lat=z["x0_lat"], x0_orientation=float(z["x0_orientation"]), wheelbase=veh[0],
                             wb_rear_axle=veh[1], length=veh[2], width=veh[3], a_max=veh[4], v_switch=veh[5],
                             delta_max=veh[6], v_delta_max=veh[7])
        ds, dss = float(z["desired_speed"]), float(z["desired_s"])
        cost = make_cost(kind=int(z["cost_kind"]), w_a=float(z["w_a"]), desired_speed=None if np.isnan(ds) else ds,
                         desired_d=float(z["desired_d"]), desired_s=None if np.isnan(dss) else dss)
        self.inputs = PlanInputs(params, cost, z["T"], z["traj_len"], z["L"], z["D"])
        self.obstacles = ObstacleTables(static_obb=z["static_obb"], static_tri=z["static_tri"],
                                        static_circ=z["static_circ"], dyn_obb=z["dyn_obb"], dyn_t0=int(z["dyn_t0"]))
        self.draw = bool(int(z["draw"]))

    def __getitem__(self, k):
        return self.z[k]

    def oracle_tables(self):
        from oracle.oracle import OracleTables
        z = self.z
        return OracleTables(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"],
                            float(z["proj_d_limit"]), self.obstacles)

    def setup_context(self, ctx):
        z = self.z
        ctx.set_reference(z["ref_pos"], z["ref_theta"], z["ref_curv"], z["ref_curv_d"], z["ref_path"],
                          float(z["proj_d_limit"]))
        ctx.set_obstacles(self.obstacles)
