"""An object with the interface of ``commonroad_rp_amd._capi.RpContext`` backed by the CPU oracle.

TEST INFRASTRUCTURE: lets the CPU-only test suite exercise the host-side glue (planner mirror,
multi-rank exchange) that normally sits on top of the HIP library.  Never used by the product."""
import numpy as np

from commonroad_rp_amd._capi import N_ARRAYS
from commonroad_rp_amd.collision import ObstacleTables
from oracle import oracle


class OracleContext:
    def __init__(self, device=0):
        self._ref = None
        self._obs = ObstacleTables()
        self._run = None
        self._inp = None
        self._range = (0, 0)
        self._N = None
        from commonroad_rp_amd import _capi
        if _capi._rpfast is None:   # (extension not built: the planner asks with getattr and takes the ctypes-shaped call)
            self.plan_packed_fast = None

    def close(self):
        pass

    def set_profiling(self, enable):
        pass

    def set_reference(self, ref_pos, ref_theta, ref_curv, ref_curv_d, ref_xy, proj_domain_d_limit=20.0):
        self._ref = (ref_pos, ref_theta, ref_curv, ref_curv_d, ref_xy, proj_domain_d_limit)

    def set_coordinate_system(self, co):
        self.set_reference(co.ref_pos, co.ref_theta, co.ref_curv, co.ref_curv_d, co.reference,
                           getattr(co, "proj_domain_d_limit", 20.0))

    def set_obstacles(self, tables=None):
        self._obs = tables if tables is not None else ObstacleTables()

    def _tables(self):
        return oracle.OracleTables(*self._ref, obstacles=self._obs)

    def plan(self, inp, cand_begin=0, cand_end=-1, want_best_states=True):
        end = inp.n_candidates if cand_end < 0 else cand_end
        self._run = oracle.plan(inp, self._tables(), cand_begin, end, want_states=True)
        self._inp, self._range, self._N = inp, (cand_begin, end), inp.params.N
        return self._run.out

    def plan_packed(self, params, cost, T, traj_len, L, D):
        from commonroad_rp_amd._capi import PlanInputs, pack_trajectory
        out = self.plan(PlanInputs(params, cost, T, traj_len, L, D))
        if out.best_index < 0:
            return out, None, None
        return out, out.best_states, pack_trajectory(np.ascontiguousarray(out.best_states), params.dt, params.wheelbase, params.x0_orientation)

    def plan_packed_fast(self, params, cost, T, traj_len, L, D, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation):
        """``RpContext.plan_packed_fast`` with the binding's REAL extension module (csrc/rp_pyfast.c) in front of the oracle: the module
        writes the cycle's fields into ``params``, iterates the sample sets into a buffer and calls what it is given as ``rp_plan`` -- here
        a callback that takes the grids out of that buffer; the oracle then plans on exactly what the library would have been handed.
        (The planner's CPU tests, the live runs against the reference included, so go through the same Python and C as on the device.)"""
        import ctypes as C
        from commonroad_rp_amd import _capi
        st = self.__dict__.get("_pk_state")
        if st is None:
            buf = (C.c_char * 32768)()
            res = _capi.RpResult()
            got = {}

            def fake(ctx, p, cst, g, lo, hi, fl, result, out):
                g = g.contents
                nd = g.nT + g.nL + g.nD
                f64, i32 = np.frombuffer(buf, dtype=np.float64), np.frombuffer(buf, dtype=np.int32)
                got["grids"] = (f64[:g.nT].copy(), i32[2 * nd:2 * nd + g.nT].copy(), f64[g.nT:g.nT + g.nL].copy(), f64[g.nT + g.nL:nd].copy())
                got["call"] = (lo, hi, fl)
                return 0
            fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_capi.RpParams), C.POINTER(_capi.RpCost), C.POINTER(_capi.RpGrids), C.c_int64, C.c_int64,
                             C.c_uint32, C.POINTER(_capi.RpResult), C.POINTER(C.c_double))(fake)
            pk = _capi._rpfast.Packed(C.cast(fn, C.c_void_p).value, 1, C.addressof(buf), 32768, C.addressof(res))
            st = self._pk_state = (pk, got, (fn, buf, res))
        pk, got, _keep = st
        out = np.empty((N_ARRAYS + 13, params.N + 1))
        rc = pk.plan(params, cost, T, traj_len, L, D, out, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation)
        assert rc == 0 and got["call"] == (0, -1, _capi.PLAN_PACKED), (rc, got.get("call"))
        return self.plan_packed(params, cost, *got["grids"])

    def plan_levels_packed(self, params, cost, levels):
        """the level loop, level by level (what rp_plan_levels does in one device round trip)"""
        res = blk = buf = None
        lvl = 0
        for lvl, (T, tl, L, D) in enumerate(levels):
            if len(T) * len(L) * len(D) == 0 and lvl + 1 < len(levels):
                continue
            res, blk, buf = self.plan_packed(params, cost, T, tl, L, D)
            if blk is not None:
                break
        self._level = lvl
        return res, lvl, blk, buf

    def plan_levels_begin(self, params, cost, levels, want_best_states=True):
        self._pending_levels = (params, cost, levels)
        self._pending = "levels"

    def last_level(self):
        return getattr(self, "_level", 0)

    def plan_begin(self, inp, cand_begin=0, cand_end=-1, want_best_states=True):
        self._pending = (inp, cand_begin, cand_end)

    def plan_wait(self):
        if self._pending == "levels":
            from commonroad_rp_amd._capi import PlanInputs
            params, cost, levels = self._pending_levels
            self._pending = None
            out = None
            for lvl, (T, tl, L, D) in enumerate(levels):
                if len(T) * len(L) * len(D) == 0 and lvl + 1 < len(levels):
                    continue
                out = self.plan(PlanInputs(params, cost, T, tl, L, D))
                self._level = lvl
                if out.best_index >= 0:
                    break
            return out
        inp, lo, hi = self._pending
        self._pending = None
        return self.plan(inp, lo, hi)

    def last_path(self):
        return 0

    def last_kernel(self):
        return "rp_eval_kernel"

    def plan_coeffs(self, params, cost, lon_coeffs, lat_coeffs, lon_T, traj_len, want_best_states=True):
        self._run = oracle.plan_coeffs(params, cost, self._tables(), lon_coeffs, lat_coeffs, traj_len)
        self._range, self._N = (0, len(traj_len)), params.N
        return self._run.out

    def fetch_status(self, first=0, count=None):
        count = len(self._run.status) - first if count is None else count
        return self._run.status[first:first + count].copy(), self._run.cost[first:first + count].copy()

    def fetch_states(self, first=0, count=None):
        count = len(self._run.status) - first if count is None else count
        return self._run.states[first:first + count].copy()

    def eval_one(self, index):
        i = index - self._range[0]
        return self._run.states[i].copy(), int(self._run.status[i]), float(self._run.cost[i])

    def count_collisions_before(self, cost, index):
        return oracle.count_collisions_before(self._run.status, self._run.cost, self._range[0], cost, index)

    def cost_range(self):
        lab = self._run.status & 3
        c = self._run.cost[((lab == 1) | (lab == 3)) & ~np.isnan(self._run.cost)]
        return (float(c.min()), float(c.max()), len(c)) if len(c) else (float("nan"), float("nan"), 0)

    def check_swept(self, params, x, y, theta, want_boxes=False):
        first, boxes = oracle.check_swept(params, self._tables(), x, y, theta, want_boxes)
        return (first, boxes) if want_boxes else first

    def select(self, costs, want_best_states=True):
        run = self._run
        lab = run.status & 3
        cost = np.where((lab == 1) | (lab == 3), np.asarray(costs, dtype=float), run.cost)
        run.cost = cost
        ok = np.flatnonzero((lab == 1) & ~np.isnan(cost))
        out = run.out
        if len(ok):
            k = ok[np.lexsort((ok, cost[ok]))[0]]
            out.best_index, out.best_cost = int(self._range[0] + k), float(cost[k])
            out.best_states = run.states[k].copy()
        else:
            out.best_index, out.best_cost, out.best_states = -1, float("nan"), None
        out.n_collision_before_best = self.count_collisions_before(out.best_cost, out.best_index)
        return out
