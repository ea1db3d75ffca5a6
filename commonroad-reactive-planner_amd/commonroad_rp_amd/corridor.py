"""Driving corridors for the adaptive sampling space (``sampling.CorridorSampling``).

The reference takes its corridor from the optional package CommonRoad-Reach (``commonroad_reach``: ``DrivingCorridor`` and
``utility.reach_operation``; imports at commonroad_rp/sampling.py:17-25, used at :311-315, :344-389), which is not under
/root/reference and not installed here.  What ``CorridorSampling`` needs of it is small: per time step a set of reach
nodes, and four questions about such sets.  This module provides both in a self-contained form:

* ``ReachBox``: one node = an axis-aligned box in the curvilinear frame with a longitudinal velocity interval;
* a ``DrivingCorridor`` = ``{time step: [ReachBox, ...]}`` (consecutive integer keys);
* the four operations with the names and call signatures of ``commonroad_reach.utility.reach_operation``.

The definitions are this build's (connectedness = overlapping or touching lateral intervals); the fixtures that pin the
sampling logic around them were produced by the reference's own ``CorridorSampling`` running on exactly these definitions
(tests/golden/make_golden.py, stand-in module ``commonroad_reach``).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Iterable, List, Tuple


@dataclasses.dataclass(frozen=True)
class ReachBox:
    p_lon_min: float
    p_lon_max: float
    p_lat_min: float
    p_lat_max: float
    v_lon_min: float
    v_lon_max: float


DrivingCorridor = Dict[int, List[ReachBox]]


def lon_velocity_interval_connected_set(nodes: Iterable[ReachBox]) -> Tuple[float, float]:
    nodes = list(nodes)
    return min(n.v_lon_min for n in nodes), max(n.v_lon_max for n in nodes)


def determine_overlapping_nodes_with_lon_pos(nodes: Iterable[ReachBox], lon_pos: float) -> List[ReachBox]:
    return [n for n in nodes if n.p_lon_min <= lon_pos <= n.p_lon_max]


def determine_connected_components(nodes: List[ReachBox]) -> List[List[ReachBox]]:
    """Groups of nodes whose lateral intervals overlap or touch (transitively), each group and the list of groups in the
    order of their first members."""
    nodes = list(nodes)
    comp = list(range(len(nodes)))

    def find(i):
        while comp[i] != i:
            comp[i] = comp[comp[i]]
            i = comp[i]
        return i
    for i in range(len(nodes)):
        for j in range(i + 1, len(nodes)):
            if nodes[i].p_lat_min <= nodes[j].p_lat_max and nodes[j].p_lat_min <= nodes[i].p_lat_max:
                a, b = find(i), find(j)
                if a != b:
                    comp[max(a, b)] = min(a, b)
    groups: Dict[int, List[ReachBox]] = {}
    for i, n in enumerate(nodes):
        groups.setdefault(find(i), []).append(n)
    return [groups[k] for k in sorted(groups)]


def lat_interval_connected_set(nodes: Iterable[ReachBox]) -> Tuple[float, float]:
    nodes = list(nodes)
    return min(n.p_lat_min for n in nodes), max(n.p_lat_max for n in nodes)
