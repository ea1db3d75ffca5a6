"""Reference-path front end on RANDOM polylines: the reference's CoordinateSystem.__init__ (smoothing, resampling, tables) and
_compute_initial_states run live against oracle/frontend.py and the library's rp_build_reference / rp_initial_state / rp_project --
the assertions of tests/test_frontend.py on reference output made here.  Only where the reference is mounted.
usage: python tests/sweeps/live_frontend.py [n=60] [first_seed=100]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "commonroad-reactive-planner_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")]
if not os.path.isdir("/root/reference/commonroad_rp"):
    sys.exit("reference not mounted")
import logging
logging.disable(logging.CRITICAL)
import make_frontend_golden as mf
import test_frontend as T
from commonroad_rp_amd import _capi
from oracle import frontend as ofe

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = states = 0
only = int(os.environ.get("LIVE_FRONTEND_ONLY", -1))
for k in range(n):
    if only >= 0 and k != only:
        continue
    rng = np.random.default_rng(seed0 + k)
    m = int(rng.integers(5, 260))
    step = rng.uniform(0.4, 6.0, m) if k % 3 else np.full(m, float(rng.uniform(0.5, 3.0)))
    kappa = np.cumsum(rng.normal(0, 0.004, m)) * float(rng.uniform(0.2, 2.0))
    th = float(rng.uniform(-np.pi, np.pi)) + np.cumsum(kappa * step)
    pts = np.stack((np.cumsum(step * np.cos(th)), np.cumsum(step * np.sin(th))), axis=1) + rng.uniform(-5e3, 5e3, 2)
    if k % 5 == 0 and m > 8:   # repeated vertices
        j = int(rng.integers(2, m - 2))
        pts = np.concatenate((pts[:j], pts[j - 1:j], pts[j:]))
    smooth = bool(k % 4 != 3)
    t0 = time.time()
    try:
        z = mf.run_case(pts, smooth, rng)
    except Exception as e:
        print(f"case {k:3d}: the reference raises {type(e).__name__}: {str(e)[:70]} -- skipped", flush=True)
        continue
    ok, why = True, ""
    try:
        wb = float(z["wheelbase"])
        tables = ofe.build_reference(z["polyline"], smooth=smooth)
        T._check_tables(tables, z)
        T._check_states(lambda x, y, th_, v, a, st, low: ofe.initial_state(tables, x, y, th_, v, a, st, wb, low), z)
        nat = _capi.build_reference(z["polyline"], smooth=smooth)
        # the library's tables: vertices / arc length / orientation as the tests ask (1e-9); curvature and its rate are second and third
        # differences over the vertex spacing h -- where resampling leaves a last vertex millimetres behind the one before, 2e-12 m of
        # difference between SciPy's spline and the restated one is 2e-6 1/m there, for the reference's own NumPy as much as for the
        # C++ side: per-vertex tolerance 1e-9 + 1e-11 / h^2 (curvature), that over h (rate)
        np.testing.assert_allclose(nat[0], z["reference"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(nat[1], z["ref_pos"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(nat[2], z["ref_theta"], rtol=0, atol=1e-8)
        h = np.diff(z["ref_pos"])
        hmin = np.minimum(np.r_[h[0], h], np.r_[h, h[-1]])
        hmin = np.minimum(hmin, np.minimum(np.r_[hmin[1:], hmin[-1]], np.r_[hmin[0], hmin[:-1]]))   # (a vertex's differences reach its neighbours')
        tol = 1e-9 + 1e-11 / hmin ** 2
        assert np.all(np.abs(nat[3] - z["ref_curv"]) <= tol), ("curvature", float(np.max(np.abs(nat[3] - z["ref_curv"]) / tol)))
        assert np.all(np.abs(nat[4] - z["ref_curv_d"]) <= 1e-8 + 4 * tol / hmin), ("curvature rate", float(np.max(np.abs(nat[4] - z["ref_curv_d"]) / (1e-8 + 4 * tol / hmin))))
        # (the library's Frenet states: 1e-7 -- they read curvature and its rate off the tables, with the conditioning described above
        #  near a path's last vertices; the tests' fixtures hold 1e-8)
        for st, lon, lat, code in zip(z["states"], z["x0_lon"], z["x0_lat"], z["code"]):
            try:
                got, gc = _capi.initial_state(*nat, st[0], st[1], st[2], st[3], st[4], st[5], wb, bool(st[6])), 0
            except ValueError:
                got, gc = None, -5
            except Exception:
                got, gc = None, -6
            assert gc == code, ("which states the library refuses", gc, int(code))
            if gc == 0:
                np.testing.assert_allclose(got[0], lon, rtol=1e-7, atol=1e-7)
                np.testing.assert_allclose(got[1], lat, rtol=1e-7, atol=1e-7)
        for st, code in zip(z["states"], z["code"]):
            if code != -5:
                # (library against oracle, both this build's definition of the projection.  Round 5: with the textbook root formula the two
                #  were 5e-5 m apart on nearly straight stretches -- and both as far from the root; see oracle/frontend.py: project)
                np.testing.assert_allclose(_capi.project(nat[0], nat[1], st[0], st[1]), ofe.project(nat[0], nat[1], st[0], st[1]), rtol=0, atol=1e-9)
    except AssertionError as e:
        import traceback
        ok, why = False, (str(e)[:300] or traceback.format_exc().strip().splitlines()[-3].strip())
    except Exception as e:   # (pytest.raises inside the helpers: a state the reference refused and this build did not, or the reverse)
        ok, why = False, f"{type(e).__name__}: {str(e)[:200]}"
    bad += not ok
    states += len(z["states"])
    print(f"case {k:3d}: {len(pts):3d} -> {len(z['ref_pos']):4d} vertices, length {z['ref_pos'][-1]:7.1f} m, smooth {int(smooth)}, max |curvature| {np.abs(z['ref_curv']).max():.3f}, "
          f"codes {sorted(set(z['code'].tolist()))}: {'OK' if ok else 'MISMATCH ' + why} ({time.time() - t0:.1f} s)", flush=True)
print(f"{states} states: " + ("all live front-end checks passed" if not bad else f"{bad} MISMATCHES"))
sys.exit(1 if bad else 0)
