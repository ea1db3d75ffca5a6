/*
 * rp_oracle.c -- CPU restatement (plain C, IEEE double, no FMA contraction) of the reference
 * hot path of commonroad_rp 2024.1.
 *
 * TEST INFRASTRUCTURE.  This file is the parity checker and the timed CPU baseline; it is NOT
 * part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call it.  The product path (commonroad_rp_amd + librp_amd.so) never links
 * or imports anything from oracle/ and fails loudly when the HIP library is missing.
 *
 * Parity status: PINNED for everything that is arithmetic of the reference itself -- checked
 * against the 18 golden fixtures in tests/golden (npz files) that were produced by running the
 * reference's own Python (tests/golden/make_golden.py); see tests/test_oracle_golden.py.
 * UNPINNED at the two third-party boundaries whose source is not under /root/reference
 * (pycrccosy (s,d)->(x,y) and pycrcc overlap tests, commonroad-drivability-checker 2024.1):
 * this build defines those semantics itself (polyline + interpolated vertex tangent; closed-set
 * separating-axis tests) and uses the same definition in the fixtures.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * The arithmetic keeps the reference's operation order (NumPy expression order, np.sum's
 * pairwise blocking, np.round's scale-rint-unscale) so that results agree to a few ulp.
 */
#include "rp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS 1e-5 /* _EPS, commonroad_rp/reactive_planner.py:49 */
static const double PI_ = 3.14159265358979323846;
static const double TWO_PI_ = 2.0 * 3.14159265358979323846;

/* ------------------------------------------------------------------------------------------- */
/* coefficient solves                                                                          */
/* ------------------------------------------------------------------------------------------- */

/* np.linalg.solve (LAPACK dgesv: LU with partial pivoting, then two triangular solves) for a
 * tiny system, column by column as dgetf2/dgetrs do.  a is row-major n x n, overwritten. */
static int lu_solve(int n, double *a, double *b) {
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(a[i * n + k]) > best) { best = fabs(a[i * n + k]); p = i; }
        if (best == 0.0) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        double rcp = 1.0 / a[k * n + k]; /* dgetf2 scales the column by the reciprocal pivot */
        for (int i = k + 1; i < n; ++i) a[i * n + k] *= rcp;
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= a[i * n + k] * a[k * n + j];
    }
    for (int k = 0; k < n; ++k) /* L y = b (unit lower) */
        for (int i = k + 1; i < n; ++i) b[i] -= b[k] * a[i * n + k];
    for (int k = n - 1; k >= 0; --k) { /* U x = y */
        b[k] /= a[k * n + k];
        for (int i = 0; i < k; ++i) b[i] -= b[k] * a[i * n + k];
    }
    return 0;
}

/* QuinticTrajectory._calc_coeffs_static, commonroad_rp/polynomial_trajectory.py:292-320 */
void rpo_quintic_coeffs(double p0, double v0, double a0, double pf, double vf, double af, double T, double c[6]) {
    double t2 = T * T; /* np.power(delta_tau, 2) */
    double t3 = t2 * T, t4 = t2 * t2, t5 = t4 * T;
    double a[9] = {t3, t4, t5, 3. * t2, 4. * t3, 5. * t4, 6. * T, 12. * t2, 20. * t3};
    double b[3] = {pf - (p0 + v0 * T + .5 * a0 * t2), vf - (v0 + a0 * T), af - a0};
    if (lu_solve(3, a, b) != 0) b[0] = b[1] = b[2] = NAN;
    c[0] = p0; c[1] = v0; c[2] = .5 * a0; c[3] = b[0]; c[4] = b[1]; c[5] = b[2];
}

/* QuarticTrajectory._calc_coeffs_static_, commonroad_rp/polynomial_trajectory.py:341-360 */
void rpo_quartic_coeffs(double p0, double v0, double a0, double T, double vd, double c[6]) {
    double t2 = T * T, t3 = t2 * T;
    double a[4] = {3. * t2, 4. * t3, 6. * T, 12. * t2};
    double b[2] = {vd - v0 - a0 * T, -a0};
    if (lu_solve(2, a, b) != 0) b[0] = b[1] = NAN;
    c[0] = p0; c[1] = v0; c[2] = .5 * a0; c[3] = b[0]; c[4] = b[1]; c[5] = 0.;
}

/* PolynomialTrajectory.calc_position/velocity/acceleration, polynomial_trajectory.py:240-271 */
static inline double poly_pos(const double *c, double t, double t2, double t3, double t4, double t5) {
    return c[0] + c[1] * t + c[2] * t2 + c[3] * t3 + c[4] * t4 + c[5] * t5;
}
static inline double poly_vel(const double *c, double t, double t2, double t3, double t4) {
    return c[1] + 2. * c[2] * t + 3. * c[3] * t2 + 4. * c[4] * t3 + 5. * c[5] * t4;
}
static inline double poly_acc(const double *c, double t, double t2, double t3) {
    return 2 * c[2] + 6 * c[3] * t + 12 * c[4] * t2 + 20 * c[5] * t3;
}

/* ------------------------------------------------------------------------------------------- */
/* reference-path helpers                                                                      */
/* ------------------------------------------------------------------------------------------- */

/* make_valid_orientation (commonroad-io 2024.1, third party; used at
 * commonroad_rp/utility/utils_coordinate_system.py:43): angle % 2pi, minus 2pi in [pi, 2pi]. */
static inline double make_valid_orientation(double a) {
    double m = fmod(a, TWO_PI_);
    if (m != 0.0) { if (m < 0.0) m += TWO_PI_; } else m = 0.0;
    if (PI_ <= m && m <= TWO_PI_) m -= TWO_PI_;
    return m;
}

/* interpolate_angle, commonroad_rp/utility/utils_coordinate_system.py:25-43 */
static inline double interpolate_angle(double x, double x1, double x2, double y1, double y2) {
    double delta = y2 - y1;
    return make_valid_orientation(delta * (x - x1) / (x2 - x1) + y1);
}

/* np.argmax(ref_pos > s) - 1, commonroad_rp/reactive_planner.py:835: index of the first table
 * entry greater than s, minus one; -1 when no entry is greater (argmax of all-False is 0) and
 * also when s < ref_pos[0] (NaN compares false everywhere => -1 as well). */
static inline int seg_index(const double *ref_pos, int n, double s) {
    for (int i = 0; i < n; ++i)
        if (ref_pos[i] > s) return i - 1;
    return -1;
}

/* Vertex tangents; same operation order as commonroad_rp_amd.coordinate_system.compute_vertex_tangents */
void rpo_vertex_tangents(int n, const double *x, const double *y, double *tx, double *ty) {
    double *ux = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n - 1 : 1));
    double *uy = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n - 1 : 1));
    for (int i = 0; i + 1 < n; ++i) {
        double ex = x[i + 1] - x[i], ey = y[i + 1] - y[i];
        double ln = sqrt(ex * ex + ey * ey);
        ux[i] = ex / ln; uy[i] = ey / ln;
    }
    tx[0] = ux[0]; ty[0] = uy[0];
    tx[n - 1] = ux[n - 2]; ty[n - 1] = uy[n - 2];
    for (int i = 1; i + 1 < n; ++i) {
        double sx = ux[i - 1] + ux[i], sy = uy[i - 1] + uy[i];
        double tn = sqrt(sx * sx + sy * sy);
        tx[i] = sx / tn; ty[i] = sy / tn;
    }
    free(ux); free(uy);
}

/* (s, d) -> (x, y): CoordinateSystem.convert_to_cartesian_coords
 * (commonroad_rp/utility/utils_coordinate_system.py:167-174 delegates to pycrccosy; semantics
 * defined by this build, see commonroad_rp_amd/coordinate_system.py).  Returns 0 for "None". */
static inline int to_cartesian(const rpo_tables *tb, const double *tx, const double *ty, double s, double d,
                               double *x, double *y) {
    const double *rp = tb->ref_pos;
    int n = tb->n_ref;
    if (!(s >= rp[0] && s <= rp[n - 1] && fabs(d) <= tb->proj_d_limit)) return 0;
    int k = seg_index(rp, n, s); /* last vertex with ref_pos[k] <= s */
    if (k < 0 || k > n - 2) k = n - 2;
    double lam = (s - rp[k]) / (rp[k + 1] - rp[k]);
    double px = tb->ref_x[k] + lam * (tb->ref_x[k + 1] - tb->ref_x[k]);
    double py = tb->ref_y[k] + lam * (tb->ref_y[k + 1] - tb->ref_y[k]);
    double ax = tx[k] + lam * (tx[k + 1] - tx[k]);
    double ay = ty[k] + lam * (ty[k + 1] - ty[k]);
    double tn = sqrt(ax * ax + ay * ay);
    *x = px - d * (ay / tn);
    *y = py + d * (ax / tn);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* collision primitives (closed sets: touching = colliding); ego = RectOBB(half_l, half_w, th,
 * cx, cy) of commonroad_rp/reactive_planner.py:1041.  Same formulas as tests/golden/_ref_shims.py */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double cx, cy, ux, uy, hl, hw; } obb_t;

static inline int obb_obb(const obb_t *a, const obb_t *b) {
    double avx = -a->uy, avy = a->ux, bvx = -b->uy, bvy = b->ux;
    double tx = b->cx - a->cx, ty = b->cy - a->cy;
    double uu = a->ux * b->ux + a->uy * b->uy;
    double uv = a->ux * bvx + a->uy * bvy;
    double vu = avx * b->ux + avy * b->uy;
    double vv = avx * bvx + avy * bvy;
    if (fabs(tx * a->ux + ty * a->uy) > a->hl + (b->hl * fabs(uu) + b->hw * fabs(uv))) return 0;
    if (fabs(tx * avx + ty * avy) > a->hw + (b->hl * fabs(vu) + b->hw * fabs(vv))) return 0;
    if (fabs(tx * b->ux + ty * b->uy) > b->hl + (a->hl * fabs(uu) + a->hw * fabs(vu))) return 0;
    if (fabs(tx * bvx + ty * bvy) > b->hw + (a->hl * fabs(uv) + a->hw * fabs(vv))) return 0;
    return 1;
}

static inline double min3(double a, double b, double c) { return fmin(a, fmin(b, c)); }
static inline double max3(double a, double b, double c) { return fmax(a, fmax(b, c)); }

static inline int obb_tri(const obb_t *a, const double *t) {
    double vx = -a->uy, vy = a->ux, lx[3], ly[3];
    for (int k = 0; k < 3; ++k) {
        double px = t[2 * k] - a->cx, py = t[2 * k + 1] - a->cy;
        lx[k] = px * a->ux + py * a->uy;
        ly[k] = px * vx + py * vy;
    }
    if (min3(lx[0], lx[1], lx[2]) > a->hl || max3(lx[0], lx[1], lx[2]) < -a->hl) return 0;
    if (min3(ly[0], ly[1], ly[2]) > a->hw || max3(ly[0], ly[1], ly[2]) < -a->hw) return 0;
    for (int k = 0; k < 3; ++k) {
        int k2 = (k + 1) % 3;
        double ex = lx[k2] - lx[k], ey = ly[k2] - ly[k];
        double nx = -ey, ny = ex;
        double p0 = lx[0] * nx + ly[0] * ny, p1 = lx[1] * nx + ly[1] * ny, p2 = lx[2] * nx + ly[2] * ny;
        double r = a->hl * fabs(nx) + a->hw * fabs(ny);
        if (min3(p0, p1, p2) > r || max3(p0, p1, p2) < -r) return 0;
    }
    return 1;
}

static inline int obb_circ(const obb_t *a, const double *c) {
    double vx = -a->uy, vy = a->ux;
    double px = c[0] - a->cx, py = c[1] - a->cy;
    double lx = px * a->ux + py * a->uy, ly = px * vx + py * vy;
    double dx = fmax(fabs(lx) - a->hl, 0.0), dy = fmax(fabs(ly) - a->hw, 0.0);
    return dx * dx + dy * dy <= c[2] * c[2];
}

/* cc.collide(ego) for one pose at scenario time index t (reactive_planner.py:1040-1042) */
static int pose_collides(const rpo_tables *tb, const obb_t *ego, int t) {
    for (int j = 0; j < tb->n_sobb; ++j) {
        const double *o = tb->sobb + 5 * j;
        obb_t b = {o[0], o[1], cos(o[2]), sin(o[2]), o[3], o[4]};
        if (obb_obb(ego, &b)) return 1;
    }
    for (int j = 0; j < tb->n_tri; ++j)
        if (obb_tri(ego, tb->tri + 6 * j)) return 1;
    for (int j = 0; j < tb->n_circ; ++j)
        if (obb_circ(ego, tb->circ + 3 * j)) return 1;
    int k = t - tb->dyn_t0;
    if (k >= 0 && k < tb->n_steps) {
        for (int j = 0; j < tb->n_dyn; ++j) {
            const double *o = tb->dyn + ((size_t)j * (size_t)tb->n_steps + (size_t)k) * 5;
            if (isnan(o[0])) continue;
            obb_t b = {o[0], o[1], cos(o[2]), sin(o[2]), o[3], o[4]};
            if (obb_obb(ego, &b)) return 1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* numpy reductions / rounding                                                                 */
/* ------------------------------------------------------------------------------------------- */

/* np.sum over a contiguous float64 vector: NumPy's pairwise summation (8 running sums per block
 * of <= 128, recursive halving above), added to the identity 0.0. */
static double np_pairwise(const double *a, long n) {
    if (n < 8) {
        double r = 0.;
        for (long i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
    }
}
double rpo_np_sum(const double *a, long n) { return 0.0 + np_pairwise(a, n); }

/* round(np.float64, 5) == np.round(x, 5): multiply by 1e5, rint, divide
 * (used on the yaw rate at commonroad_rp/reactive_planner.py:995) */
static inline double np_round5(double x) { return rint(x * 1e5) / 1e5; }

/* ------------------------------------------------------------------------------------------- */
/* one candidate                                                                               */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    const rp_params *p;
    const rp_cost *cost;
    const rpo_tables *tb;
    const double *tx, *ty; /* vertex tangents */
    double kappa_max;      /* np.tan(delta_max) / wheelbase, reactive_planner.py:985 */
} eval_env;

/* ReactivePlanner._check_constraints, commonroad_rp/reactive_planner.py:971-1017.
 * Returns RP_REASON_NONE when all enabled checks pass, else the first failing reason. */
static unsigned check_constraints(const eval_env *e, const double *v, const double *kappa, const double *theta,
                                  const double *a, int i) {
    const rp_params *p = e->p;
    if (p->constraint_mask & RP_CHECK_VELOCITY)
        if (v[i] < -EPS) return RP_REASON_VELOCITY;
    double kappa_max = e->kappa_max;
    if (p->constraint_mask & RP_CHECK_KAPPA)
        if (fabs(kappa[i]) > kappa_max) return RP_REASON_KAPPA;
    if (p->constraint_mask & RP_CHECK_YAW_RATE) {
        double yaw_rate = i > 0 ? (theta[i] - theta[i - 1]) / p->dt : 0.;
        double theta_dot_max = kappa_max * v[i];
        if (fabs(np_round5(yaw_rate)) > theta_dot_max) return RP_REASON_YAW_RATE;
    }
    if (p->constraint_mask & RP_CHECK_KAPPA_DOT) {
        double steering_angle = atan2(p->wheelbase * kappa[i], 1.0);
        double cs = cos(steering_angle);
        double kappa_dot_max = p->v_delta_max / (p->wheelbase * (cs * cs));
        double kappa_dot = i > 0 ? (kappa[i] - kappa[i - 1]) / p->dt : 0.;
        if (fabs(kappa_dot) > kappa_dot_max) return RP_REASON_KAPPA_DOT;
    }
    if (p->constraint_mask & RP_CHECK_ACCELERATION) {
        double a_max = v[i] > p->v_switch ? p->a_max * p->v_switch / v[i] : p->a_max;
        double a_min = -p->a_max;
        if (!(a_min <= a[i] && a[i] <= a_max)) return RP_REASON_ACCELERATION;
    }
    return RP_REASON_NONE;
}

/* DefaultCostFunction.evaluate / DefaultCostFunctionFailSafe.evaluate,
 * commonroad_rp/cost_function.py:51-71, 82-92 (over the extended N+1 arrays). tmp: N+1 scratch */
static double eval_cost(const rp_cost *c, int n, const double *st, double *tmp) {
    const double *a = st + RP_A * n, *v = st + RP_V * n, *s = st + RP_S * n, *d = st + RP_D * n,
                 *th = st + RP_THETA_CL * n;
    double costs;
    if (c->kind == RP_COST_FAILSAFE) {
        for (int i = 0; i < n; ++i) { double q = 1 * a[i]; tmp[i] = q * q; }
        costs = rpo_np_sum(tmp, n);
        for (int i = 0; i < n; ++i) { double q = 0.25 * d[i]; tmp[i] = q * q; }
        double e = 20 * d[n - 1];
        costs += rpo_np_sum(tmp, n) + e * e;
        for (int i = 0; i < n; ++i) { double q = 0.25 * fabs(th[i]); tmp[i] = q * q; }
        e = 5 * fabs(th[n - 1]);
        costs += rpo_np_sum(tmp, n) + e * e;
        return costs;
    }
    costs = 0.0;
    for (int i = 0; i < n; ++i) { double q = c->w_a * a[i]; tmp[i] = q * q; }
    costs += rpo_np_sum(tmp, n);
    if (!isnan(c->desired_speed)) {
        double vd = c->desired_speed;
        for (int i = 0; i < n; ++i) { double q = 5 * (v[i] - vd); tmp[i] = q * q; }
        double e1 = v[n - 1] - vd, e2 = v[n / 2] - vd; /* int(len(v) / 2) */
        costs += rpo_np_sum(tmp, n) + (50 * (e1 * e1)) + (100 * (e2 * e2));
    }
    if (!isnan(c->desired_s)) {
        double sd = c->desired_s;
        for (int i = 0; i < n; ++i) { double q = 0.25 * (sd - s[i]); tmp[i] = q * q; }
        double e = 20 * (sd - s[n - 1]);
        costs += rpo_np_sum(tmp, n) + e * e;
    }
    {
        double dd = c->desired_d;
        for (int i = 0; i < n; ++i) { double q = 0.25 * (dd - d[i]); tmp[i] = q * q; }
        double e = 20 * (dd - d[n - 1]);
        costs += rpo_np_sum(tmp, n) + e * e;
    }
    for (int i = 0; i < n; ++i) { double q = 0.25 * fabs(th[i]); tmp[i] = q * q; }
    double e = 5 * fabs(th[n - 1]);
    costs += rpo_np_sum(tmp, n) + e * e;
    return costs;
}

/* CartesianSample.enlarge + CurviLinearSample.enlarge, commonroad_rp/trajectories.py:168-197,
 * 302-332.  st: 14 rows of n = N+1; L = traj_len (< n). */
static void enlarge(double *st, int n, int L, double dt) {
    double *x = st + RP_X * n, *y = st + RP_Y * n, *th = st + RP_THETA * n, *v = st + RP_V * n, *a = st + RP_A * n,
           *ka = st + RP_KAPPA * n, *kd = st + RP_KAPPA_DOT * n;
    double *s = st + RP_S * n, *d = st + RP_D * n, *tc = st + RP_THETA_CL * n, *sd = st + RP_S_DOT * n,
           *sdd = st + RP_S_DDOT * n, *dd = st + RP_D_DOT * n, *ddd = st + RP_D_DDOT * n;
    int last = L - 1, steps = n - L;
    /* Cartesian (trajectories.py:173-197) */
    for (int k = 0; k < steps; ++k) a[L + k] = a[last];
    double cth = cos(th[last]), sth = sin(th[last]);
    double cx = 0., cy = 0.; /* np.cumsum runs sequentially */
    for (int k = 0; k < steps; ++k) {
        double t = (double)(k + 1) * dt;        /* np.arange(1, steps + 1, 1) * dt */
        double vt = v[last] + t * a[n - 1];       /* a[-1] already holds a[last] */
        vt = vt * (vt >= 0 ? 1.0 : 0.0);          /* v_temp * np.greater_equal(v_temp, 0) */
        v[L + k] = vt;
        th[L + k] = th[last];
        ka[L + k] = ka[last];
        kd[L + k] = kd[last];
        cx = (k == 0) ? dt * vt * cth : cx + dt * vt * cth;
        cy = (k == 0) ? dt * vt * sth : cy + dt * vt * sth;
        x[L + k] = x[last] + cx;
        y[L + k] = y[last] + cy;
    }
    /* curvilinear (trajectories.py:307-332): s_ddot[-1] / d_ddot[-1] are read before the arrays
     * are extended, i.e. they are the zero padding */
    double sdd_pad = sdd[n - 1], ddd_pad = ddd[n - 1];
    for (int k = 0; k < steps; ++k) {
        double t = (double)(k + 1) * dt;
        double sv = sd[last] + t * sdd_pad;
        sv = sv * (sv >= 0 ? 1.0 : 0.0);
        sd[L + k] = sv;
        dd[L + k] = dd[last] + t * ddd_pad;
    }
    for (int k = 0; k < steps; ++k) {
        double t = (double)(k + 1) * dt;
        sdd[L + k] = sdd[last];
        ddd[L + k] = ddd[last];
        tc[L + k] = tc[last];
        s[L + k] = s[last] + t * sd[last];
        d[L + k] = d[last] + t * dd[last];
    }
}

/* One iteration of the loop body of ReactivePlanner._check_kinematics
 * (commonroad_rp/reactive_planner.py:731-960) + cost + eager collision check.
 * st: 14 x (N+1) block (fully written); returns the status word, *cost_out = cost or NaN. */
static uint32_t eval_candidate(const eval_env *e, const double *lon, const double *lat, int traj_len, double *st,
                               double *tmp, double *cost_out) {
    const rp_params *p = e->p;
    const rpo_tables *tb = e->tb;
    const int N = p->N, n = N + 1;
    const int draw = (p->flags & RP_FLAG_DRAW_ALL) != 0;
    const double dt = p->dt;
    double *x = st + RP_X * n, *y = st + RP_Y * n, *theta_gl = st + RP_THETA * n, *v = st + RP_V * n,
           *a = st + RP_A * n, *kappa_gl = st + RP_KAPPA * n, *kappa_dot = st + RP_KAPPA_DOT * n;
    double *s = st + RP_S * n, *d = st + RP_D * n, *theta_cl = st + RP_THETA_CL * n, *s_vel = st + RP_S_DOT * n,
           *s_acc = st + RP_S_DDOT * n, *d_vel = st + RP_D_DOT * n, *d_acc = st + RP_D_DDOT * n;
    memset(st, 0, sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n); /* np.zeros(N + 1), :740-745,781-791 */
    *cost_out = NAN;
    if (traj_len > n) traj_len = n;

    /* polynomial evaluation, :733-772 */
    for (int i = 0; i < traj_len; ++i) {
        double t = (double)i * dt; /* np.arange(0, ., dt)[i] */
        double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        s[i] = poly_pos(lon, t, t2, t3, t4, t5);
        s_vel[i] = poly_vel(lon, t, t2, t3, t4);
        s_acc[i] = poly_acc(lon, t, t2, t3);
        if (!p->low_vel_mode) {
            d[i] = poly_pos(lat, t, t2, t3, t4, t5);
            d_vel[i] = poly_vel(lat, t, t2, t3, t4);
            d_acc[i] = poly_acc(lat, t, t2, t3);
        }
    }
    if (p->low_vel_mode) {
        for (int i = 0; i < traj_len; ++i) {
            double s1 = s[i] - s[0];
            double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
            d[i] = poly_pos(lat, s1, s2, s3, s4, s5);
            d_vel[i] = poly_vel(lat, s1, s2, s3, s4);
            d_acc[i] = poly_acc(lat, s1, s2, s3);
        }
    }
    for (int i = 0; i < n; ++i) { /* :776-777 */
        if (fabs(s_vel[i]) < EPS) s_vel[i] = 0.0;
        if (fabs(d_vel[i]) < EPS) d_vel[i] = 0.0;
    }

    /* pre-filter, :796-805 (label stays None) */
    if (!draw) {
        int bad = 0;
        for (int i = 0; i < n; ++i) if (fabs(s_acc[i]) > p->a_max) { bad = 1; break; }
        if (bad) return RP_LABEL_NONE | (RP_REASON_ACCELERATION << 4);
        for (int i = 0; i < n; ++i) if (s_vel[i] < -EPS) { bad = 1; break; }
        if (bad) return RP_LABEL_NONE | (RP_REASON_VELOCITY << 4);
    }

    /* per-step conversion + constraints, :807-904 */
    int feasible = 1;
    unsigned reason = RP_REASON_NONE;
    int bad_step = 0;
    const double *ref_pos = tb->ref_pos;
    for (int i = 0; i < traj_len; ++i) {
        double dp, dpp;
        if (!p->low_vel_mode) {
            dp = s_vel[i] > 0.001 ? d_vel[i] / s_vel[i] : 0.;
            double ddot = d_acc[i] - dp * s_acc[i];
            dpp = s_vel[i] > 0.001 ? ddot / (s_vel[i] * s_vel[i]) : 0.;
        } else {
            dp = d_vel[i];
            dpp = d_acc[i];
        }
        int s_idx = seg_index(ref_pos, tb->n_ref, s[i]);
        /* the guard at :836 can never fire; Python's negative index wraps to the last entry */
        int k0 = s_idx < 0 ? tb->n_ref - 1 : s_idx, k1 = s_idx + 1;
        double s_lambda = (s[i] - ref_pos[k0]) / (ref_pos[k1] - ref_pos[k0]);
        if (s_vel[i] > 0.001 || p->low_vel_mode) {
            theta_cl[i] = atan2(dp, 1.0);
            theta_gl[i] = theta_cl[i] +
                          interpolate_angle(s[i], ref_pos[k0], ref_pos[k1], tb->ref_theta[k0], tb->ref_theta[k1]);
        } else {
            theta_gl[i] = i == 0 ? p->x0_orientation : theta_gl[i - 1];
            theta_cl[i] = theta_gl[i] -
                          interpolate_angle(s[i], ref_pos[k0], ref_pos[k1], tb->ref_theta[k0], tb->ref_theta[k1]);
        }
        double k_r = (tb->ref_curv[k1] - tb->ref_curv[k0]) * s_lambda + tb->ref_curv[k0];
        double k_r_d = (tb->ref_curv_d[k1] - tb->ref_curv_d[k0]) * s_lambda + tb->ref_curv_d[k0];
        double oneKrD = 1 - k_r * d[i];
        double cosTheta = cos(theta_cl[i]);
        double tanTheta = tan(theta_cl[i]);
        double q = cosTheta / oneKrD;
        kappa_gl[i] = (dpp + (k_r * dp + k_r_d * d[i]) * tanTheta) * cosTheta * (q * q) + q * k_r;
        v[i] = s_vel[i] * (oneKrD / cosTheta);
        a[i] = s_acc[i] * oneKrD / cosTheta +
               ((s_vel[i] * s_vel[i]) / cosTheta) *
                   (oneKrD * tanTheta * (kappa_gl[i] * oneKrD / cosTheta - k_r) - (k_r_d * d[i] + k_r * dp));
        if (feasible) {
            unsigned r = check_constraints(e, v, kappa_gl, theta_gl, a, i);
            if (r != RP_REASON_NONE) { feasible = 0; reason = r; bad_step = i; }
        }
        if (!feasible && !draw) break;
    }
    if (!feasible && !draw)
        return RP_LABEL_INFEASIBLE_KINEMATIC | (reason << 4) | ((uint32_t)bad_step << 8);

    /* positions, :907-917 */
    int kin_ok = feasible;
    for (int i = 0; i < traj_len; ++i) {
        if (!to_cartesian(tb, e->tx, e->ty, s[i], d[i], &x[i], &y[i])) {
            if (feasible) { reason = RP_REASON_OUT_OF_DOMAIN; bad_step = i; }
            feasible = 0;
            break;
        }
    }
    if (!feasible && !draw) /* label stays None, no reason counter */
        return RP_LABEL_NONE | (reason << 4) | ((uint32_t)bad_step << 8);

    /* kappa_dot over the padded array + horizon extension, :919-934 / :942-957 */
    kappa_dot[0] = 0;
    for (int i = 1; i < n; ++i) kappa_dot[i] = kappa_gl[i] - kappa_gl[i - 1];
    if (n > traj_len) enlarge(st, n, traj_len, dt);
    if (!feasible) { /* draw mode: stored as INFEASIBLE_KINEMATIC (:942-943), never costed */
        (void)kin_ok;
        return RP_LABEL_INFEASIBLE_KINEMATIC | (reason << 4) | ((uint32_t)bad_step << 8);
    }

    *cost_out = eval_cost(e->cost, n, st, tmp);

    /* eager version of _check_collisions, commonroad_rp/reactive_planner.py:1031-1046 */
    if (!(p->flags & RP_FLAG_SKIP_COLLISION)) {
        double hl = 0.5 * p->length, hw = 0.5 * p->width;
        for (int i = 0; i < n; ++i) {
            double c = cos(theta_gl[i]), sn = sin(theta_gl[i]);
            obb_t ego = {x[i] + p->wb_rear_axle * c, y[i] + p->wb_rear_axle * sn, c, sn, hl, hw};
            if (pose_collides(tb, &ego, p->time_step0 + i * p->factor))
                return RP_LABEL_INFEASIBLE_COLLISION | ((uint32_t)i << 8);
        }
    }
    return RP_LABEL_FEASIBLE;
}

/* ------------------------------------------------------------------------------------------- */
/* selection: TrajectoryBundle.sort + lazy _check_collisions == lexicographic (cost, index) min   */
/* over collision-free feasible candidates (trajectories.py:502-510, reactive_planner.py:1031-1063) */
/* ------------------------------------------------------------------------------------------- */
static void select_best(int64_t C, int64_t base, const uint32_t *status, const double *cost, rp_result *res) {
    int64_t best = -1;
    double bc = NAN;
    memset(res->reason_counts, 0, sizeof(res->reason_counts));
    res->n_feasible = res->n_collision = res->n_collision_before_best = 0;
    for (int64_t i = 0; i < C; ++i) {
        unsigned lab = RP_STATUS_LABEL(status[i]), rs = RP_STATUS_REASON(status[i]);
        if (rs) res->reason_counts[rs]++;
        if (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION) res->n_feasible++;
        if (lab == RP_LABEL_INFEASIBLE_COLLISION) res->n_collision++;
        if (lab == RP_LABEL_FEASIBLE && (best < 0 || cost[i] < bc)) { best = i; bc = cost[i]; }
    }
    for (int64_t i = 0; i < C; ++i)
        if (RP_STATUS_LABEL(status[i]) == RP_LABEL_INFEASIBLE_COLLISION &&
            (best < 0 || cost[i] < bc || (cost[i] == bc && i < best)))
            res->n_collision_before_best++;
    res->best_index = best < 0 ? -1 : base + best;
    res->best_cost = bc;
    res->n_candidates = C;
}

int64_t rp_oracle_count_collisions_before(int64_t C, int64_t base, const uint32_t *status, const double *cost,
                                          double wcost, int64_t windex) {
    int64_t n = 0;
    for (int64_t i = 0; i < C; ++i)
        if (RP_STATUS_LABEL(status[i]) == RP_LABEL_INFEASIBLE_COLLISION &&
            (windex < 0 || cost[i] < wcost || (cost[i] == wcost && base + i < windex)))
            ++n;
    return n;
}

/* ------------------------------------------------------------------------------------------- */
/* continuous collision check of one trajectory, commonroad_rp/reactive_planner.py:1049-1058.   */
/* trajectory_preprocess_obb_sum belongs to commonroad-dc 2024.1 (absent): the tight rectangle   */
/* around two consecutive ego rectangles is DEFINED here as the planar form of FCL's OBB sum     */
/* (OBB::operator+: bisector orientation, or the centre line when the centres are farther apart  */
/* than twice the sum of the larger half extents; extents from projecting both rectangles) --    */
/* same definition as tests/golden/_ref_shims.py, which the reference's own _check_collisions is */
/* run with for the fixtures.  Parity with the real commonroad-dc routine: unpinned.            */
/* ------------------------------------------------------------------------------------------- */
static obb_t merge_swept(const obb_t *a, const obb_t *b) {
    double dx = b->cx - a->cx, dy = b->cy - a->cy;
    double lim = 2.0 * (fmax(a->hl, a->hw) + fmax(b->hl, b->hw));
    double nx, ny;
    if (dx * dx + dy * dy > lim * lim) { nx = dx; ny = dy; }
    else {
        double s = (a->ux * b->ux + a->uy * b->uy) < 0.0 ? -1.0 : 1.0;
        nx = a->ux + s * b->ux; ny = a->uy + s * b->uy;
    }
    double nrm = sqrt(nx * nx + ny * ny);
    nx = nx / nrm; ny = ny / nrm;
    double mx = -ny, my = nx;
    double ea0 = a->hl * fabs(a->ux * nx + a->uy * ny) + a->hw * fabs(a->uy * nx - a->ux * ny);
    double ea1 = a->hl * fabs(a->ux * mx + a->uy * my) + a->hw * fabs(a->uy * mx - a->ux * my);
    double eb0 = b->hl * fabs(b->ux * nx + b->uy * ny) + b->hw * fabs(b->uy * nx - b->ux * ny);
    double eb1 = b->hl * fabs(b->ux * mx + b->uy * my) + b->hw * fabs(b->uy * mx - b->ux * my);
    double pb0 = dx * nx + dy * ny, pb1 = dx * mx + dy * my;
    double lo0 = fmin(-ea0, pb0 - eb0), hi0 = fmax(ea0, pb0 + eb0);
    double lo1 = fmin(-ea1, pb1 - eb1), hi1 = fmax(ea1, pb1 + eb1);
    double c0 = 0.5 * (lo0 + hi0), c1 = 0.5 * (lo1 + hi1);
    obb_t m = {a->cx + (c0 * nx + c1 * mx), a->cy + (c0 * ny + c1 * my), nx, ny, 0.5 * (hi0 - lo0), 0.5 * (hi1 - lo1)};
    return m;
}

/* trajectory_preprocess_obb_sum of a dynamic OBSTACLE (commonroad_rp/reactive_planner.py:238-245, applied by
 * set_collision_checker when planning.continuous_collision_check is on): rows of (cx, cy, theta, hl, hw), one per time
 * step, NaN cx = absent.  out[k] = rectangle around rows k and k + 1 (same definition as merge_swept above); where
 * either is absent, and at the last step, out[k] is absent: the reference's object is one shape shorter (:241). */
void rp_oracle_obb_sum_rows(int n_steps, const double *rows, double *out) {
    for (int k = 0; k < n_steps; ++k) {
        double *o = out + 5 * (size_t)k;
        o[0] = o[1] = o[2] = o[3] = o[4] = NAN;
        if (k + 1 >= n_steps) continue;
        const double *ra = rows + 5 * (size_t)k, *rb = ra + 5;
        if (isnan(ra[0]) || isnan(rb[0])) continue;
        obb_t a = {ra[0], ra[1], cos(ra[2]), sin(ra[2]), ra[3], ra[4]};
        obb_t b = {rb[0], rb[1], cos(rb[2]), sin(rb[2]), rb[3], rb[4]};
        obb_t m = merge_swept(&a, &b);
        o[0] = m.cx; o[1] = m.cy; o[2] = atan2(m.uy, m.ux); o[3] = m.hl; o[4] = m.hw;
    }
}

/* first colliding segment or -1; boxes: optional [n-1][6] (cx, cy, ux, uy, hl, hw).  Segment i is tested at scenario
 * time index time_step0 + i: the ego object of :1050 starts at x_0.time_step, one shape per index, no factor. */
int rp_oracle_check_swept(const rp_params *p, const rpo_tables *tb, int n, const double *x, const double *y,
                          const double *theta, double *boxes) {
    double hl = 0.5 * p->length, hw = 0.5 * p->width;
    int first = -1;
    for (int i = 0; i + 1 < n; ++i) {
        double c0 = cos(theta[i]), s0 = sin(theta[i]), c1 = cos(theta[i + 1]), s1 = sin(theta[i + 1]);
        obb_t a = {x[i] + p->wb_rear_axle * c0, y[i] + p->wb_rear_axle * s0, c0, s0, hl, hw};
        obb_t b = {x[i + 1] + p->wb_rear_axle * c1, y[i + 1] + p->wb_rear_axle * s1, c1, s1, hl, hw};
        obb_t m = merge_swept(&a, &b);
        if (boxes) {
            double *o = boxes + 6 * (size_t)i;
            o[0] = m.cx; o[1] = m.cy; o[2] = m.ux; o[3] = m.uy; o[4] = m.hl; o[5] = m.hw;
        }
        if (first < 0 && pose_collides(tb, &m, p->time_step0 + i)) {
            first = i;
            if (!boxes) break;
        }
    }
    return first;
}

/* cc.collide(ego rectangle of pose i at scenario time index time_step0 + i * factor), pose by pose (reactive_planner.py:1033-1046):
 * hit[n] = 0 / 1.  What tests/test_thirdparty_pins.py compares with the verdicts of the real pycrcc. */
void rp_oracle_check_poses(const rp_params *p, const rpo_tables *tb, int n, const double *x, const double *y, const double *theta, int32_t *hit) {
    const double hl = 0.5 * p->length, hw = 0.5 * p->width;
    for (int i = 0; i < n; ++i) {
        const double c0 = cos(theta[i]), s0 = sin(theta[i]);
        obb_t ego = {x[i] + p->wb_rear_axle * c0, y[i] + p->wb_rear_axle * s0, c0, s0, hl, hw};
        hit[i] = pose_collides(tb, &ego, p->time_step0 + i * p->factor);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* sampling: FixedIntervalSampling.generate_trajectories_at_level, commonroad_rp/sampling.py:202-242 */
/* ------------------------------------------------------------------------------------------- */
void rp_oracle_sample(const rp_params *p, const rp_grids *g, int64_t idx, double lon[6], double lat[6],
                      double *lat_T, int *traj_len) {
    int64_t nLD = (int64_t)g->nL * g->nD;
    int iT = (int)(idx / nLD);
    int iL = (int)((idx % nLD) / g->nD);
    int iD = (int)(idx % g->nD);
    double T = g->T[iT];
    if (p->lon_mode == RP_LON_STOPPING) /* sampling.py:259-263: end state [s, 0, 0] */
        rpo_quintic_coeffs(p->x0_lon[0], p->x0_lon[1], p->x0_lon[2], g->L[iL], 0.0, 0.0, T, lon);
    else /* sampling.py:254-258: end state [v, 0] */
        rpo_quartic_coeffs(p->x0_lon[0], p->x0_lon[1], p->x0_lon[2], T, g->L[iL], lon);
    double tau = T;
    if (p->low_vel_mode) { /* sampling.py:229-234; evaluate_state_at_tau, polynomial_trajectory.py:212-218 */
        double t2 = T * T, t3 = t2 * T, t4 = t2 * t2, t5 = t3 * t2;
        double s_goal = poly_pos(lon, T, t2, t3, t4, t5) - p->x0_lon[0];
        if (s_goal <= 0) s_goal = T;
        tau = s_goal;
    }
    rpo_quintic_coeffs(p->x0_lat[0], p->x0_lat[1], p->x0_lat[2], g->D[iD], 0.0, 0.0, tau, lat); /* sampling.py:227 */
    *lat_T = tau;
    *traj_len = g->traj_len[iT];
}

static int run(const rp_params *p, const rp_cost *cost, const rpo_tables *tb, const rp_grids *g, int64_t C,
               int64_t base, const double *lon_in, const double *lat_in, const int32_t *tl_in, uint32_t *status,
               double *cost_out, double *coeffs, double *states, rp_result *result, double *best_states,
               int nthreads) {
    if (!p || !cost || !tb || !result || tb->n_ref < 2 || p->N < 1) return RP_EINVAL;
    const int n = p->N + 1;
    const size_t blk = (size_t)RP_N_ARRAYS * (size_t)n;
    double *tx = (double *)malloc(sizeof(double) * (size_t)tb->n_ref);
    double *ty = (double *)malloc(sizeof(double) * (size_t)tb->n_ref);
    uint32_t *st_own = status ? NULL : (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(C > 0 ? C : 1));
    double *co_own = cost_out ? NULL : (double *)malloc(sizeof(double) * (size_t)(C > 0 ? C : 1));
    if (!status) status = st_own;
    if (!cost_out) cost_out = co_own;
    rpo_vertex_tangents(tb->n_ref, tb->ref_x, tb->ref_y, tx, ty);
    eval_env env = {p, cost, tb, tx, ty, tan(p->delta_max) / p->wheelbase};
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        double *scratch = (double *)malloc(sizeof(double) * (blk + (size_t)n));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 32)
#endif
        for (int64_t i = 0; i < C; ++i) {
            double lon[6], lat[6], lat_T = NAN;
            int tl;
            if (g) {
                rp_oracle_sample(p, g, base + i, lon, lat, &lat_T, &tl);
            } else {
                memcpy(lon, lon_in + 6 * i, sizeof(lon));
                memcpy(lat, lat_in + 6 * i, sizeof(lat));
                tl = tl_in[i];
            }
            if (coeffs) {
                memcpy(coeffs + 13 * i, lon, sizeof(lon));
                memcpy(coeffs + 13 * i + 6, lat, sizeof(lat));
                coeffs[13 * i + 12] = lat_T;
            }
            double *st = states ? states + blk * (size_t)i : scratch;
            status[i] = eval_candidate(&env, lon, lat, tl, st, scratch + blk, &cost_out[i]);
        }
        free(scratch);
    }
    select_best(C, base, status, cost_out, result);
    result->kernel_ms = 0;
    result->best_lat_T = NAN;
    for (int k = 0; k < 6; ++k) result->best_lon_coeffs[k] = result->best_lat_coeffs[k] = NAN;
    if (result->best_index >= 0) {
        int64_t w = result->best_index - base;
        double lon[6], lat[6], lat_T = NAN, c;
        int tl;
        if (g) rp_oracle_sample(p, g, result->best_index, lon, lat, &lat_T, &tl);
        else { memcpy(lon, lon_in + 6 * w, sizeof(lon)); memcpy(lat, lat_in + 6 * w, sizeof(lat)); tl = tl_in[w]; }
        memcpy(result->best_lon_coeffs, lon, sizeof(lon));
        memcpy(result->best_lat_coeffs, lat, sizeof(lat));
        result->best_lat_T = lat_T;
        if (best_states) {
            double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
            eval_candidate(&env, lon, lat, tl, best_states, tmp, &c);
            free(tmp);
        }
    }
    free(tx); free(ty); free(st_own); free(co_own);
    return RP_OK;
}

int rp_oracle_plan(const rp_params *p, const rp_cost *cost, const rp_grids *g, const rpo_tables *tb,
                   int64_t cand_begin, int64_t cand_end, uint32_t *status, double *cost_out, double *coeffs,
                   double *states, rp_result *result, double *best_states, int nthreads) {
    if (!g || g->nT < 0 || g->nL < 0 || g->nD < 0) return RP_EINVAL;
    int64_t total = (int64_t)g->nT * g->nL * g->nD;
    if (cand_end < 0) cand_end = total;
    if (cand_begin < 0 || cand_begin > cand_end || cand_end > total) return RP_EINVAL;
    return run(p, cost, tb, g, cand_end - cand_begin, cand_begin, NULL, NULL, NULL, status, cost_out, coeffs, states,
               result, best_states, nthreads);
}

int rp_oracle_plan_coeffs(const rp_params *p, const rp_cost *cost, const rpo_tables *tb, int64_t C,
                          const double *lon_coeffs, const double *lat_coeffs, const int32_t *traj_len,
                          uint32_t *status, double *cost_out, double *states, rp_result *result,
                          double *best_states, int nthreads) {
    if (C < 0 || !lon_coeffs || !lat_coeffs || !traj_len) return RP_EINVAL;
    return run(p, cost, tb, NULL, C, 0, lon_coeffs, lat_coeffs, traj_len, status, cost_out, NULL, states, result,
               best_states, nthreads);
}
