#!/usr/bin/env python3
"""Extract the *data* of the reference's example scenarios (CommonRoad XML under
/root/reference/example_scenarios) that BASELINE.json's configs are quoted on into small .npz
fixtures: route centre line, initial state, static / dynamic obstacle boxes per time step.

Runs ONLY in the build container.  The XML parser of commonroad-io is not installed (and out of
scope), so the few fields needed are read with ElementTree.  The route
(sequence of lanelets) is fixed by hand here because commonroad-route-planner is not installed.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

SRC = "/root/reference/example_scenarios"
HERE = os.path.dirname(os.path.abspath(__file__))

ROUTES = {
    # scenario file -> lanelet ids along the route
    "ZAM_Tjunction-1_42_T-1": ["50195", "50209", "50203"],   # left turn at the junction
    "DEU_Test-1_1_T-1": ["1", "3"],
    "ZAM_Over-1_1": ["1000"],
    # the acceleration lane scenario: initial state at standstill on the middle lane (lanelet 5), goal 50 m ahead
    "ZAM-Ramp-1_1-T-1": ["5", "6", "7", "8"],
}
# metres of straight lead-in put in front of a route's centre line.  The ramp scenario starts with the vehicle's CENTRE on the
# first vertex of its lanelet, i.e. the rear axle -- the planner's reference point (state.py:52-55) -- 1.4 m in front of the
# polyline: pycrccosy extends a reference path at both ends, this build's polyline transform does not
LEAD_IN = {"ZAM-Ramp-1_1-T-1": 5.0}


def pts(node):
    return np.array([(float(p.find("x").text), float(p.find("y").text)) for p in node.findall("point")])


def first_float(node, tag):
    el = node.find(tag)
    if el is None:
        return 0.0
    ex = el.find("exact")
    return float(ex.text) if ex is not None else float(el[0].text)


def state(node):
    pos = node.find("position").find("point")
    return (float(pos.find("x").text), float(pos.find("y").text), first_float(node, "orientation"),
            int(first_float(node, "time")), first_float(node, "velocity"))


def main():
    for name, route in ROUTES.items():
        root = ET.parse(os.path.join(SRC, name + ".xml")).getroot()
        lanelets = {ll.attrib["id"]: ll for ll in root.findall("lanelet")}
        centre = []
        for lid in route:
            ll = lanelets[lid]
            left, right = pts(ll.find("leftBound")), pts(ll.find("rightBound"))
            c = 0.5 * (left + right)
            centre.append(c if not centre else c[1:])
        centre = np.concatenate(centre, axis=0)
        if name in LEAD_IN:
            d = (centre[1] - centre[0]) / np.linalg.norm(centre[1] - centre[0])
            centre = np.concatenate(([centre[0] - LEAD_IN[name] * d], centre), axis=0)
        left_b = np.concatenate([pts(lanelets[l].find("leftBound")) for l in route])
        right_b = np.concatenate([pts(lanelets[l].find("rightBound")) for l in route])
        dyn_nodes = root.findall("dynamicObstacle")
        n_steps = 0
        for ob in dyn_nodes:
            tr = ob.find("trajectory")
            n_steps = max(n_steps, 1 + (len(tr.findall("state")) if tr is not None else 0))
        dyn = np.full((len(dyn_nodes), n_steps, 5), np.nan)
        for j, ob in enumerate(dyn_nodes):
            rect = ob.find("shape").find("rectangle")
            hl, hw = 0.5 * float(rect.find("length").text), 0.5 * float(rect.find("width").text)
            sts = [state(ob.find("initialState"))] + [state(s) for s in ob.find("trajectory").findall("state")]
            for (x, y, th, t, _v) in sts:
                if 0 <= t < n_steps:
                    dyn[j, t] = (x, y, th, hl, hw)
        sobb = []
        static_nodes = root.findall("staticObstacle") + [o for o in root.findall("obstacle")
                                                         if o.find("role") is not None and o.find("role").text == "static"]
        for ob in static_nodes:
            rect = ob.find("shape").find("rectangle")
            x, y, th, _t, _v = state(ob.find("initialState"))
            sobb.append((x, y, th, 0.5 * float(rect.find("length").text), 0.5 * float(rect.find("width").text)))
        # the whole lanelet network (road-boundary generator, collision.road_boundary_obb): vertices + which
        # neighbours exist
        ll_left, ll_right, ll_off, ll_flags, ll_ids = [], [], [0], [], []
        for lid, ll in lanelets.items():
            left, right = pts(ll.find("leftBound")), pts(ll.find("rightBound"))
            assert len(left) == len(right)
            ll_left.append(left); ll_right.append(right); ll_off.append(ll_off[-1] + len(left)); ll_ids.append(int(lid))
            ll_flags.append([ll.find("adjacentLeft") is not None, ll.find("adjacentRight") is not None,
                             ll.find("predecessor") is not None, ll.find("successor") is not None])
        ini = root.find("planningProblem").find("initialState")
        x, y, th, t0, v = state(ini)
        out = os.path.join(HERE, "scenario_" + name + ".npz")
        np.savez_compressed(out, centre=centre, left_bound=left_b, right_bound=right_b, dyn_obb=dyn, dyn_t0=0,
                            static_obb=np.array(sobb).reshape(-1, 5), init=np.array([x, y, th, v]), init_time_step=t0,
                            dt=float(root.attrib["timeStepSize"]),
                            ll_ids=np.array(ll_ids), ll_left=np.concatenate(ll_left), ll_right=np.concatenate(ll_right),
                            ll_offsets=np.array(ll_off), ll_flags=np.array(ll_flags, dtype=bool))
        print(name, "centre", centre.shape, "dyn", dyn.shape, "static", len(sobb), "init", (x, y, th, v),
              f"{os.path.getsize(out) / 1024:.0f}KB")


if __name__ == "__main__":
    main()
