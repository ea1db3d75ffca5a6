"""Road-boundary generator (commonroad_rp_amd.collision.road_boundary_obb): SURVEY 8f item 1.  The reference
delegates to commonroad_dc.boundary.create_road_boundary_obstacle (not under /root/reference); these are
known-answer tests of this build's own definition (thin rectangles along the outer border of the network)."""
import numpy as np

from commonroad_rp_amd.collision import SimpleLanelet, road_boundary_obb, lanelets_from_arrays, ObstacleTables


def _straight(lid, y_right, y_left, x0=0.0, x1=40.0, n=5, **kw):
    x = np.linspace(x0, x1, n)
    return SimpleLanelet(lid, np.stack((x, np.full(n, y_left)), axis=1), np.stack((x, np.full(n, y_right)), axis=1), **kw)


def test_two_adjacent_lanes_keep_only_the_outer_border():
    a = _straight(1, 0.0, 3.5, adj_left=2)
    b = _straight(2, 3.5, 7.0, adj_right=1)
    obb = road_boundary_obb([a, b], thickness=0.2)
    # right border of lane 1 (y = 0), left border of lane 2 (y = 7): 4 collinear segments each, merged into one
    # 40 m rectangle; 2 + 2 end caps
    ys = np.round(obb[:, 1], 6)
    assert np.sum(ys == 0.0) == 1 and np.sum(ys == 7.0) == 1
    assert np.allclose(obb[(ys == 0.0) | (ys == 7.0), 3], 20.0)
    caps = obb[np.abs(np.abs(obb[:, 2]) - np.pi / 2) < 1e-9]
    assert len(caps) == 4 and np.allclose(sorted(caps[:, 0]), [0, 0, 40, 40])
    assert len(obb) == 6 and np.allclose(obb[:, 4], 0.1)
    assert not np.any(ys == 3.5)           # the shared border is drivable
    # without merging: one rectangle per polyline segment
    assert len(road_boundary_obb([a, b], simplify_tol=-1.0)) == 12


def test_successor_and_predecessor_remove_end_caps():
    a = _straight(1, 0.0, 3.5, x0=0, x1=20, successor=(2,))
    b = _straight(2, 0.0, 3.5, x0=20, x1=40, predecessor=(1,))
    obb = road_boundary_obb([a, b])
    caps = obb[np.abs(np.abs(obb[:, 2]) - np.pi / 2) < 1e-9]
    assert np.allclose(sorted(caps[:, 0]), [0, 40])      # only the two outer ends are closed


def test_crossing_lanelets_drop_borders_inside_the_other_lanelet():
    a = _straight(1, -1.75, 1.75, x0=-20, x1=20, n=41)
    x = np.linspace(-20, 20, 41)
    b = SimpleLanelet(2, np.stack((np.full(41, -1.75), x), axis=1), np.stack((np.full(41, 1.75), x), axis=1))   # along y
    obb = road_boundary_obb([a, b])
    # no boundary rectangle may lie inside the crossing square (-1.75, 1.75)^2
    inside = (np.abs(obb[:, 0]) < 1.7) & (np.abs(obb[:, 1]) < 1.7)
    assert not inside.any()
    # and the borders outside the crossing survive: 2 borders x (40 - 4 crossing segments) per lanelet + 4 caps
    assert len(road_boundary_obb([a, b], simplify_tol=-1.0)) == 2 * 2 * 36 + 4
    # merged: every border is cut into two runs by the crossing
    assert len(obb) == 2 * 2 * 2 + 4 and np.isclose(np.sum(2 * obb[:, 3]), 2 * 2 * 36 + 4 * 3.5)


def test_fixture_networks_and_tables():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "scenario_ZAM_Tjunction-1_42_T-1.npz"))
    lls = lanelets_from_arrays(z["ll_ids"], z["ll_left"], z["ll_right"], z["ll_offsets"], z["ll_flags"])
    obb = road_boundary_obb(lls)
    assert len(lls) == 12 and 60 < len(obb) < 400 and np.all(obb[:, 3] > 0)
    assert len(obb) <= len(road_boundary_obb(lls, simplify_tol=-1.0))

    class _Net: lanelets = lls
    class _Sc: static_obstacles = []; dynamic_obstacles = []; lanelet_network = _Net()
    tab = ObstacleTables.from_scenario(_Sc(), road_boundary=True)
    assert len(tab.static_obb) == len(obb)
