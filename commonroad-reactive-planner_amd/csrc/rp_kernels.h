// rp_kernels.h -- the fused sample -> convert -> check -> cost -> collide kernel and the small
// selection kernels around it.  Replaces, per candidate, the loop body of
// ReactivePlanner._check_kinematics (commonroad_rp/reactive_planner.py:731-960) together with
// sampling (sampling.py:218-241), cost (cost_function.py:51-71) and the collision query
// (reactive_planner.py:1031-1046); citations inline.
#pragma once

#include "rp_device.h"

#define RP_BLOCK 256  // 4 wavefronts per workgroup

// Everything the evaluation kernel needs; passed by value (kernarg segment).
struct KArgs {
    // problem
    double dt;
    int32_t N, factor, time_step0, low_vel_mode, lon_mode;
    uint32_t constraint_mask, flags;
    double x0_lon[3], x0_lat[3], x0_orientation;
    double wheelbase, wb_rear_axle, half_length, half_width, ego_radius;
    double a_max, v_switch, v_delta_max, kappa_max;
    // cost
    int32_t cost_kind, has_speed, has_s, pad0_;
    double w_a, desired_speed, desired_d, desired_s;
    // grids (device pointers) or explicit polynomials
    int32_t nT, nL, nD, pad1_;
    const double *T;
    const int32_t *traj_len;
    const double *L;
    const double *D;
    const double *lon_coeffs;  // [C][6]  (COEFFS_IN)
    const double *lat_coeffs;  // [C][6]
    const int32_t *traj_len_c; // [C]
    // candidate range: global indices [cand_begin, cand_begin + count)
    int64_t cand_begin, count;
    const int64_t *single_index;  // != nullptr: evaluate exactly this one (global) candidate -> slot 0
    // tables
    const double *tables;  // [TB_ROWS][n_ref]
    int32_t n_ref, search_iters;
    double proj_d_limit;
    ObsTables obs;
    // outputs
    uint32_t *status;  // [count]
    double *cost;      // [count]
    double *states;    // [count][14][N+1] (MAT) or nullptr
    double *coeffs;    // [count][13] or nullptr (lon 6, lat 6, lat_T)
    struct BlockPartial *partials;  // [gridDim.x] or nullptr
};

struct BlockPartial {
    double best_cost;
    int64_t best_index;  // global index, -1 none
    int64_t n_feasible, n_collision;
    int64_t reasons[8];
};

struct DevResult {
    rp_result r;
    // followed in the same allocation by best_states[14][N+1]
};

__device__ __forceinline__ bool better(double c, int64_t i, double bc, int64_t bi) {
    return bi < 0 || c < bc || (c == bc && i < bi);
}

// ------------------------------------------------------------------------------------------------
// The fused kernel.  G lanes per candidate; MAT: write the 14 state rows; COEFFS_IN: polynomials
// are given; LDS_TABLES: reference-path tables staged in LDS (else read through L1/L2).
// ------------------------------------------------------------------------------------------------
template <int G, bool MAT, bool COEFFS_IN, bool LDS_TABLES>
__global__ __launch_bounds__(RP_BLOCK) void rp_eval_kernel(const KArgs a) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int n_ref = a.n_ref;
    const double *tab;
    if (LDS_TABLES) {
        for (int k = tid; k < TB_ROWS * n_ref; k += RP_BLOCK) lds[k] = a.tables[k];
        __syncthreads();
        tab = lds;
    } else {
        tab = a.tables;
    }
    const double *t_pos = tab + TB_POS * n_ref, *t_theta = tab + TB_THETA * n_ref, *t_curv = tab + TB_CURV * n_ref,
                 *t_curv_d = tab + TB_CURV_D * n_ref, *t_x = tab + TB_X * n_ref, *t_y = tab + TB_Y * n_ref,
                 *t_tx = tab + TB_TX * n_ref, *t_ty = tab + TB_TY * n_ref;
    const double pos_first = t_pos[0], pos_last = t_pos[n_ref - 1];

    const int lane = tid & 63;
    const int gl = lane & (G - 1);        // lane inside the group = time step inside the chunk
    const int gbase = lane & ~(G - 1);    // first lane of the group inside the wave
    constexpr int GPB = RP_BLOCK / G;     // groups per block
    constexpr int GPW = 64 / G;           // groups per wave
    const int wave_in_block = tid >> 6;
    const int group_in_wave = gbase / G;

    const int N = a.N, n = N + 1;
    const int nchunks = (n + G - 1) / G;
    const double dt = a.dt;
    const bool draw = (a.flags & RP_FLAG_DRAW_ALL) != 0;
    const bool low = a.low_vel_mode != 0;
    const bool check_coll = (a.flags & RP_FLAG_SKIP_COLLISION) == 0 &&
                            (a.obs.n_sobb + a.obs.n_tri + a.obs.n_circ > 0 || (a.obs.n_dyn > 0 && a.obs.n_steps > 0));
    const uint32_t cm = a.constraint_mask;
    const int mid = n / 2;  // int(len(v) / 2), cost_function.py:59

    // running per-group selection state (identical in all lanes of a group)
    double best_cost = 0.0;
    int64_t best_index = -1;
    int cnt_feasible = 0, cnt_collision = 0;
    int cnt_reason[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    if (a.single_index && *a.single_index < 0) return;   // no winner to re-evaluate (uniform)

    const int64_t total_groups = (int64_t)gridDim.x * GPB;
    const int64_t wave_first = ((int64_t)blockIdx.x * GPB) + (int64_t)wave_in_block * GPW;  // first group of this wave
    const int64_t count = a.single_index ? 1 : a.count;

    for (int64_t w0 = wave_first; w0 < count; w0 += total_groups) {   // wave-uniform trip count
        const int64_t slot = w0 + group_in_wave;                      // local candidate slot of this group
        const bool valid = slot < count;
        const int64_t gidx = a.single_index ? *a.single_index : a.cand_begin + (valid ? slot : 0);

        // ---- sampling: FixedIntervalSampling.generate_trajectories_at_level, sampling.py:218-241 ----
        Poly lon, lat;
        double lat_T;
        int L;
        if (COEFFS_IN) {
            const double *pl = a.lon_coeffs + 6 * gidx, *pt = a.lat_coeffs + 6 * gidx;
            lon = {pl[0], pl[1], pl[2], pl[3], pl[4], pl[5]};
            lat = {pt[0], pt[1], pt[2], pt[3], pt[4], pt[5]};
            lat_T = 0.0;
            L = a.traj_len_c[gidx];
        } else {
            const int64_t nLD = (int64_t)a.nL * a.nD;
            const int iT = (int)(gidx / nLD);
            const int rem = (int)(gidx - (int64_t)iT * nLD);
            const int iL = rem / a.nD, iD = rem - iL * a.nD;
            const double T = a.T[iT];
            L = a.traj_len[iT];
            if (a.lon_mode == RP_LON_STOPPING)
                lon = quintic_coeffs(a.x0_lon[0], a.x0_lon[1], a.x0_lon[2], a.L[iL], 0.0, 0.0, T);  // sampling.py:259-263
            else
                lon = quartic_coeffs(a.x0_lon[0], a.x0_lon[1], a.x0_lon[2], T, a.L[iL]);             // sampling.py:254-258
            lat_T = T;
            if (low) {  // sampling.py:229-234 (evaluate_state_at_tau clamps nothing here: tau == delta_tau)
                double sg = lon.pos(T) - a.x0_lon[0];
                lat_T = sg <= 0.0 ? T : sg;
            }
            lat = quintic_coeffs(a.x0_lat[0], a.x0_lat[1], a.x0_lat[2], a.D[iD], 0.0, 0.0, lat_T);  // sampling.py:227
        }
        L = L > n ? n : (L < 1 ? 1 : L);
        if (a.coeffs && valid && gl == 0) {
            double *o = a.coeffs + 13 * slot;
            o[0] = lon.c0; o[1] = lon.c1; o[2] = lon.c2; o[3] = lon.c3; o[4] = lon.c4; o[5] = lon.c5;
            o[6] = lat.c0; o[7] = lat.c1; o[8] = lat.c2; o[9] = lat.c3; o[10] = lat.c4; o[11] = lat.c5;
            o[12] = lat_T;
        }

        // ---- pre-filter, reactive_planner.py:796-805 (label stays None) ----
        uint32_t pre_reason = RP_REASON_NONE;
        if (!draw) {
            bool bad_a = false, bad_v = false;
            for (int c = 0; c < nchunks; ++c) {
                const int i = c * G + gl;
                const double t = (double)i * dt;
                double sd = lon.vel(t), sdd = lon.acc(t);
                if (fabs(sd) < RP_EPS) sd = 0.0;
                bad_a |= (i < L) && (fabs(sdd) > a.a_max);
                bad_v |= (i < L) && (sd < -RP_EPS);
            }
            const bool any_a = group_ballot<G>(bad_a, gbase) != 0, any_v = group_ballot<G>(bad_v, gbase) != 0;
            pre_reason = any_a ? RP_REASON_ACCELERATION : (any_v ? RP_REASON_VELOCITY : RP_REASON_NONE);
        }

        // per-candidate carried state
        int fail_step = -1, ood_step = -1;
        uint32_t fail_reason = RP_REASON_NONE;
        bool collide = false;
        double acc_a = 0.0, acc_v = 0.0, acc_s = 0.0, acc_d = 0.0, acc_th = 0.0;
        double v_end = 0.0, v_mid = 0.0, s_end = 0.0, d_end = 0.0, th_end = 0.0;

        bool alive = valid && pre_reason == RP_REASON_NONE;
        if (__ballot(alive) != 0) {   // wave-uniform
            double theta_carry = a.x0_orientation;  // theta_gl[base-1]; at base 0 the i == 0 rule of :866
            double kappa_carry = 0.0;
            double cumx = 0.0, cumy = 0.0;
            double x_l = 0, y_l = 0, th_l = 0, v_l = 0, a_l = 0, ka_l = 0, kd_l = 0, s_l = 0, d_l = 0, thc_l = 0, sd_l = 0,
                   sdd_l = 0, dd_l = 0, ddd_l = 0, cth_l = 1, sth_l = 0;

            for (int c = 0; c < nchunks; ++c) {   // wave-uniform
                const int base = c * G;
                const int i = base + gl;
                const bool live = i <= N;
                const bool act = i < L;

                // -- polynomial evaluation, reactive_planner.py:733-777
                const double t = (double)i * dt;
                double s = lon.pos(t), sd = lon.vel(t), sdd = lon.acc(t);
                const double tau = low ? s - lon.c0 : t;   // s[0] == c0 exactly (:762)
                double d = lat.pos(tau), dd = lat.vel(tau), ddd = lat.acc(tau);
                if (fabs(sd) < RP_EPS) sd = 0.0;
                if (fabs(dd) < RP_EPS) dd = 0.0;

                // -- d', d'' (:810-832)
                const bool moving = sd > 0.001;
                double dp, dpp;
                if (!low) {
                    dp = moving ? dd / sd : 0.0;
                    const double ddot = ddd - dp * sdd;
                    dpp = moving ? ddot / (sd * sd) : 0.0;
                } else {
                    dp = dd;
                    dpp = ddd;
                }

                // -- segment lookup + interpolation factors (:835-839); Python's negative index wraps
                const int ub = upper_bound(t_pos, n_ref, a.search_iters, s);
                const int s_idx = (ub == n_ref) ? -1 : ub - 1;
                const int k0 = s_idx < 0 ? n_ref - 1 : s_idx, k1 = s_idx + 1;
                const double p0 = t_pos[k0], p1 = t_pos[k1];
                const double lam = (s - p0) / (p1 - p0);
                const double th0 = t_theta[k0];
                const double th_ref = make_valid_orientation((t_theta[k1] - th0) * (s - p0) / (p1 - p0) + th0);  // interpolate_angle

                // -- orientations (:842-873) incl. the standstill carry of :866
                const bool use_atan = moving || low;
                const double th_cl_m = atan(dp);          // np.arctan2(dp, 1.0)
                const double th_gl_m = th_cl_m + th_ref;
                const uint64_t mv = group_ballot<G>(use_atan && act, gbase);
                const uint64_t below = mv & ((1ull << gl) - 1ull);
                const int src = below ? 63 - __clzll(below) : 0;
                const double th_from = group_bcast<G>(th_gl_m, src);
                double th_gl = use_atan ? th_gl_m : (below ? th_from : theta_carry);
                double th_cl = use_atan ? th_cl_m : th_gl - th_ref;

                // -- curvature, velocity, acceleration (:876-896)
                const double c0 = t_curv[k0], cd0 = t_curv_d[k0];
                const double k_r = (t_curv[k1] - c0) * lam + c0;
                const double k_r_d = (t_curv_d[k1] - cd0) * lam + cd0;
                const double oneKrD = 1.0 - k_r * d;
                const double cosT = cos(th_cl), tanT = tan(th_cl);
                const double q = cosT / oneKrD;
                double kappa = (dpp + (k_r * dp + k_r_d * d) * tanT) * cosT * (q * q) + q * k_r;
                double v = sd * (oneKrD / cosT);
                double acc = sdd * oneKrD / cosT +
                             ((sd * sd) / cosT) * (oneKrD * tanT * (kappa * oneKrD / cosT - k_r) - (k_r_d * d + k_r * dp));

                // -- previous-step values for the finite differences
                double th_prev = __shfl_up(th_gl, 1, G), ka_prev = __shfl_up(kappa, 1, G);
                if (gl == 0) { th_prev = theta_carry; ka_prev = kappa_carry; }

                // -- _check_constraints, reactive_planner.py:971-1017 (order: velocity, kappa, yaw, kappa_dot, acc)
                uint32_t reason = RP_REASON_NONE;
                if (act) {
                    const double yaw = i > 0 ? (th_gl - th_prev) / dt : 0.0;
                    const double kdot_t = i > 0 ? (kappa - ka_prev) / dt : 0.0;
                    const double sa = atan(a.wheelbase * kappa);
                    const double cs = cos(sa);
                    const double kdot_max = a.v_delta_max / (a.wheelbase * (cs * cs));
                    const double amax = v > a.v_switch ? a.a_max * a.v_switch / v : a.a_max;
                    if ((cm & RP_CHECK_VELOCITY) && v < -RP_EPS) reason = RP_REASON_VELOCITY;
                    else if ((cm & RP_CHECK_KAPPA) && fabs(kappa) > a.kappa_max) reason = RP_REASON_KAPPA;
                    else if ((cm & RP_CHECK_YAW_RATE) && fabs(rint(yaw * 1e5) / 1e5) > a.kappa_max * v) reason = RP_REASON_YAW_RATE;
                    else if ((cm & RP_CHECK_KAPPA_DOT) && fabs(kdot_t) > kdot_max) reason = RP_REASON_KAPPA_DOT;
                    else if ((cm & RP_CHECK_ACCELERATION) && !(-a.a_max <= acc && acc <= amax)) reason = RP_REASON_ACCELERATION;
                }
                const uint64_t fm = group_ballot<G>(reason != RP_REASON_NONE, gbase);
                const int fl = fm ? __ffsll((unsigned long long)fm) - 1 : 0;
                const uint32_t r_first = (uint32_t)__shfl((int)reason, fl, G);
                if (fm && fail_step < 0) { fail_step = base + fl; fail_reason = r_first; }
                if (!draw && fail_step >= 0) alive = false;
                if (__ballot(alive) == 0) break;   // wave-uniform: every candidate of this wave is decided

                // -- (s, d) -> (x, y), reactive_planner.py:908-917
                const bool in_dom = s >= pos_first && s <= pos_last && fabs(d) <= a.proj_d_limit;
                int k = ub - 1;
                k = k < 0 ? 0 : (k > n_ref - 2 ? n_ref - 2 : k);
                const double q0 = t_pos[k];
                const double lam2 = (s - q0) / (t_pos[k + 1] - q0);
                const double bx = t_x[k], by = t_y[k], ux0 = t_tx[k], uy0 = t_ty[k];
                const double px = bx + lam2 * (t_x[k + 1] - bx), py = by + lam2 * (t_y[k + 1] - by);
                const double ax = ux0 + lam2 * (t_tx[k + 1] - ux0), ay = uy0 + lam2 * (t_ty[k + 1] - uy0);
                const double tn = sqrt(ax * ax + ay * ay);
                double x = px - d * (ay / tn), y = py + d * (ax / tn);
                const uint64_t om = group_ballot<G>(act && !in_dom, gbase);
                if (om && ood_step < 0) ood_step = base + __ffsll((unsigned long long)om) - 1;
                if (ood_step >= 0 && i >= ood_step) { x = 0.0; y = 0.0; }   // x, y stay np.zeros past the break

                double kdot = i > 0 ? kappa - ka_prev : 0.0;   // np.append([0], np.diff(kappa_gl)), :923

                // -- last valid state -> horizon extension (trajectories.py:168-197, 302-332)
                const int ll = L - 1 - base;
                {   // unconditional shuffles (all lanes), values only kept when this chunk holds step L-1
                    const int sl = (ll >= 0 && ll < G) ? ll : 0;
                    const bool take = (ll >= 0 && ll < G);
                    double t0;
                    t0 = group_bcast<G>(x, sl); x_l = take ? t0 : x_l;
                    t0 = group_bcast<G>(y, sl); y_l = take ? t0 : y_l;
                    t0 = group_bcast<G>(th_gl, sl); th_l = take ? t0 : th_l;
                    t0 = group_bcast<G>(v, sl); v_l = take ? t0 : v_l;
                    t0 = group_bcast<G>(acc, sl); a_l = take ? t0 : a_l;
                    t0 = group_bcast<G>(kappa, sl); ka_l = take ? t0 : ka_l;
                    t0 = group_bcast<G>(kdot, sl); kd_l = take ? t0 : kd_l;
                    t0 = group_bcast<G>(s, sl); s_l = take ? t0 : s_l;
                    t0 = group_bcast<G>(d, sl); d_l = take ? t0 : d_l;
                    t0 = group_bcast<G>(th_cl, sl); thc_l = take ? t0 : thc_l;
                    t0 = group_bcast<G>(sd, sl); sd_l = take ? t0 : sd_l;
                    t0 = group_bcast<G>(sdd, sl); sdd_l = take ? t0 : sdd_l;
                    t0 = group_bcast<G>(dd, sl); dd_l = take ? t0 : dd_l;
                    t0 = group_bcast<G>(ddd, sl); ddd_l = take ? t0 : ddd_l;
                    if (take) { cth_l = cos(th_l); sth_l = sin(th_l); }
                }
                double termx = 0.0, termy = 0.0;
                if (!act) {   // i >= L: extended state
                    const double tk = (double)(i - L + 1) * dt;   // np.arange(1, steps + 1) * dt
                    acc = a_l;                                     // :179
                    double vt = v_l + tk * a_l;                    // :182 (a[-1] already holds a[last])
                    vt = vt * (vt >= 0.0 ? 1.0 : 0.0);             // :184
                    v = vt;
                    th_gl = th_l; kappa = ka_l; kdot = kd_l;       // :188-192
                    termx = dt * vt * cth_l;                       // :195-196
                    termy = dt * vt * sth_l;
                    double sv = sd_l + tk * 0.0;                   // :313, s_ddot[-1] is the zero padding
                    sd = sv * (sv >= 0.0 ? 1.0 : 0.0);             // :315
                    dd = dd_l + tk * 0.0;                          // :319
                    sdd = sdd_l; ddd = ddd_l; th_cl = thc_l;       // :323-327
                    s = s_l + tk * sd_l;                           // :330
                    d = d_l + tk * dd_l;                           // :331
                }
                const double scx = group_scan<G>(termx, gl) + cumx, scy = group_scan<G>(termy, gl) + cumy;   // np.cumsum
                if (!act) { x = x_l + scx; y = y_l + scy; }
                cumx = group_bcast<G>(scx, G - 1);
                cumy = group_bcast<G>(scy, G - 1);
                theta_carry = group_bcast<G>(th_gl, G - 1);
                kappa_carry = group_bcast<G>(kappa, G - 1);

                // -- cost terms over the extended arrays, cost_function.py:51-71 / 82-92
                if (live) {
                    double e;
                    e = a.w_a * acc; acc_a += e * e;
                    e = 5.0 * (v - a.desired_speed); acc_v += e * e;
                    e = 0.25 * (a.desired_s - s); acc_s += e * e;
                    e = 0.25 * (a.desired_d - d); acc_d += e * e;
                    e = 0.25 * fabs(th_cl); acc_th += e * e;
                }
                {
                    const int le = N - base, lm = mid - base;
                    const bool he = le >= 0 && le < G, hm = lm >= 0 && lm < G;
                    double t0;
                    t0 = group_bcast<G>(v, he ? le : 0); v_end = he ? t0 : v_end;
                    t0 = group_bcast<G>(s, he ? le : 0); s_end = he ? t0 : s_end;
                    t0 = group_bcast<G>(d, he ? le : 0); d_end = he ? t0 : d_end;
                    t0 = group_bcast<G>(th_cl, he ? le : 0); th_end = he ? t0 : th_end;
                    t0 = group_bcast<G>(v, hm ? lm : 0); v_mid = hm ? t0 : v_mid;
                }

                // -- eager collision query for every pose, reactive_planner.py:1033-1046
                if (check_coll) {
                    bool hit = false;
                    if (live && alive && fail_step < 0 && ood_step < 0) {
                        double sn, cn;
                        sincos(th_gl, &sn, &cn);
                        const Obb ego = {x + a.wb_rear_axle * cn, y + a.wb_rear_axle * sn, cn, sn, a.half_length, a.half_width};
                        hit = pose_collides(a.obs, ego, a.ego_radius, a.time_step0 + i * a.factor);
                    }
                    collide |= group_ballot<G>(hit, gbase) != 0;
                }

                // -- state block (coalesced along the time axis)
                if (MAT) {
                    if (live && valid && pre_reason == RP_REASON_NONE) {
                        double *o = a.states + ((size_t)slot * RP_N_ARRAYS) * (size_t)n + (size_t)i;
                        o[(size_t)RP_X * n] = x;
                        o[(size_t)RP_Y * n] = y;
                        o[(size_t)RP_THETA * n] = th_gl;
                        o[(size_t)RP_V * n] = v;
                        o[(size_t)RP_A * n] = acc;
                        o[(size_t)RP_KAPPA * n] = kappa;
                        o[(size_t)RP_KAPPA_DOT * n] = kdot;
                        o[(size_t)RP_S * n] = s;
                        o[(size_t)RP_D * n] = d;
                        o[(size_t)RP_THETA_CL * n] = th_cl;
                        o[(size_t)RP_S_DOT * n] = sd;
                        o[(size_t)RP_S_DDOT * n] = sdd;
                        o[(size_t)RP_D_DOT * n] = dd;
                        o[(size_t)RP_D_DDOT * n] = ddd;
                    }
                }
            }
        }

        // ---- cost reduction (all lanes take part) ----
        const double sum_a = group_sum<G>(acc_a), sum_v = group_sum<G>(acc_v), sum_s = group_sum<G>(acc_s),
                     sum_d = group_sum<G>(acc_d), sum_th = group_sum<G>(acc_th);

        // ---- label, reason, cost ----
        uint32_t status;
        double cost = __builtin_nan("");
        if (pre_reason != RP_REASON_NONE) {
            status = RP_LABEL_NONE | (pre_reason << 4);
        } else if (fail_step >= 0) {   // :902 INFEASIBLE_KINEMATIC
            status = RP_LABEL_INFEASIBLE_KINEMATIC | (fail_reason << 4) | ((uint32_t)fail_step << 8);
        } else if (ood_step >= 0) {    // :915; draw mode stores it as INFEASIBLE_KINEMATIC (:942-943)
            status = (draw ? RP_LABEL_INFEASIBLE_KINEMATIC : RP_LABEL_NONE) | (RP_REASON_OUT_OF_DOMAIN << 4) |
                     ((uint32_t)ood_step << 8);
        } else {
            double e;
            if (a.cost_kind == RP_COST_FAILSAFE) {   // cost_function.py:82-92 (w_a == 1, desired_d == 0 set by the host)
                cost = sum_a;
                e = 20.0 * d_end; cost += sum_d + e * e;
                e = 5.0 * fabs(th_end); cost += sum_th + e * e;
            } else {                                 // cost_function.py:51-71
                cost = 0.0 + sum_a;
                if (a.has_speed) {
                    const double e1 = v_end - a.desired_speed, e2 = v_mid - a.desired_speed;
                    cost += sum_v + (50.0 * (e1 * e1)) + (100.0 * (e2 * e2));
                }
                if (a.has_s) { e = 20.0 * (a.desired_s - s_end); cost += sum_s + e * e; }
                e = 20.0 * (a.desired_d - d_end); cost += sum_d + e * e;
                e = 5.0 * fabs(th_end); cost += sum_th + e * e;
            }
            status = collide ? RP_LABEL_INFEASIBLE_COLLISION : RP_LABEL_FEASIBLE;
        }
        if (valid) {
            if (gl == 0) {
                a.status[slot] = status;
                a.cost[slot] = cost;
            }
            const uint32_t lab = RP_STATUS_LABEL(status), rs = RP_STATUS_REASON(status);
#pragma unroll
            for (int r = 0; r < 8; ++r) cnt_reason[r] += (rs == (uint32_t)r);   // no runtime-indexed register array
            cnt_feasible += (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION);
            cnt_collision += (lab == RP_LABEL_INFEASIBLE_COLLISION);
            if (lab == RP_LABEL_FEASIBLE && cost == cost && better(cost, gidx, best_cost, best_index)) {
                best_cost = cost;
                best_index = gidx;
            }
        }
    }

    // ---- block partial: lexicographic (cost, index) min + counters ----
    if (a.partials) {
        __shared__ double sh_cost[GPB];
        __shared__ int64_t sh_idx[GPB];
        __shared__ int sh_cnt[GPB][10];
        const int g = tid / G;
        __syncthreads();
        if (gl == 0) {
            sh_cost[g] = best_cost;
            sh_idx[g] = best_index;
            sh_cnt[g][0] = cnt_feasible;
            sh_cnt[g][1] = cnt_collision;
            for (int r = 0; r < 8; ++r) sh_cnt[g][2 + r] = cnt_reason[r];
        }
        __syncthreads();
        if (tid == 0) {
            BlockPartial bp;
            bp.best_cost = 0.0;
            bp.best_index = -1;
            bp.n_feasible = bp.n_collision = 0;
            for (int r = 0; r < 8; ++r) bp.reasons[r] = 0;
            for (int k = 0; k < GPB; ++k) {
                if (sh_idx[k] >= 0 && better(sh_cost[k], sh_idx[k], bp.best_cost, bp.best_index)) {
                    bp.best_cost = sh_cost[k];
                    bp.best_index = sh_idx[k];
                }
                bp.n_feasible += sh_cnt[k][0];
                bp.n_collision += sh_cnt[k][1];
                for (int r = 0; r < 8; ++r) bp.reasons[r] += sh_cnt[k][2 + r];
            }
            a.partials[blockIdx.x] = bp;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Block partials from the status / cost arrays (plug-in cost path, rp_select).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RP_BLOCK) void rp_partials_kernel(const uint32_t *status, const double *cost, int64_t count,
                                                               int64_t cand_begin, BlockPartial *partials) {
    double bc = 0.0;
    int64_t bi = -1;
    int nf = 0, nc = 0, rs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * RP_BLOCK + threadIdx.x; i < count; i += (int64_t)gridDim.x * RP_BLOCK) {
        const uint32_t st = status[i], lab = RP_STATUS_LABEL(st);
        const double c = cost[i];
        const uint32_t rsn = RP_STATUS_REASON(st);
#pragma unroll
        for (int r = 0; r < 8; ++r) rs[r] += (rsn == (uint32_t)r);
        nf += (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION);
        nc += (lab == RP_LABEL_INFEASIBLE_COLLISION);
        if (lab == RP_LABEL_FEASIBLE && c == c && better(c, cand_begin + i, bc, bi)) { bc = c; bi = cand_begin + i; }
    }
    __shared__ double sh_cost[RP_BLOCK];
    __shared__ int64_t sh_idx[RP_BLOCK];
    __shared__ int sh_cnt[10];
    if (threadIdx.x < 10) sh_cnt[threadIdx.x] = 0;
    sh_cost[threadIdx.x] = bc;
    sh_idx[threadIdx.x] = bi;
    __syncthreads();
    atomicAdd(&sh_cnt[0], nf);
    atomicAdd(&sh_cnt[1], nc);
    for (int r = 0; r < 8; ++r)
        if (rs[r]) atomicAdd(&sh_cnt[2 + r], rs[r]);
    __syncthreads();
    if (threadIdx.x == 0) {
        BlockPartial bp;
        bp.best_cost = 0.0;
        bp.best_index = -1;
        for (int k = 0; k < RP_BLOCK; ++k)
            if (sh_idx[k] >= 0 && better(sh_cost[k], sh_idx[k], bp.best_cost, bp.best_index)) {
                bp.best_cost = sh_cost[k];
                bp.best_index = sh_idx[k];
            }
        bp.n_feasible = sh_cnt[0];
        bp.n_collision = sh_cnt[1];
        for (int r = 0; r < 8; ++r) bp.reasons[r] = sh_cnt[2 + r];
        partials[blockIdx.x] = bp;
    }
}

// ------------------------------------------------------------------------------------------------
// Final reduction of the block partials -> DevResult (one workgroup).  TrajectoryBundle.sort +
// "first collision-free sample" (trajectories.py:502-510, reactive_planner.py:1031-1063) is the
// lexicographic (cost, list index) minimum over feasible, collision-free candidates.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RP_BLOCK) void rp_reduce_kernel(const BlockPartial *partials, int n_partials, int64_t count,
                                                             rp_result *res) {
    __shared__ double sh_cost[RP_BLOCK];
    __shared__ int64_t sh_idx[RP_BLOCK];
    __shared__ unsigned long long sh_cnt[10];
    double bc = 0.0;
    int64_t bi = -1;
    unsigned long long c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < n_partials; k += RP_BLOCK) {
        const BlockPartial p = partials[k];
        if (p.best_index >= 0 && better(p.best_cost, p.best_index, bc, bi)) { bc = p.best_cost; bi = p.best_index; }
        c[0] += (unsigned long long)p.n_feasible;
        c[1] += (unsigned long long)p.n_collision;
        for (int r = 0; r < 8; ++r) c[2 + r] += (unsigned long long)p.reasons[r];
    }
    if (threadIdx.x < 10) sh_cnt[threadIdx.x] = 0;
    sh_cost[threadIdx.x] = bc;
    sh_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int r = 0; r < 10; ++r)
        if (c[r]) atomicAdd(&sh_cnt[r], c[r]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double fc = 0.0;
        int64_t fi = -1;
        for (int k = 0; k < RP_BLOCK; ++k)
            if (sh_idx[k] >= 0 && better(sh_cost[k], sh_idx[k], fc, fi)) { fc = sh_cost[k]; fi = sh_idx[k]; }
        res->best_index = fi;
        res->best_cost = fi >= 0 ? fc : __builtin_nan("");
        res->n_candidates = count;
        res->n_feasible = (int64_t)sh_cnt[0];
        res->n_collision = (int64_t)sh_cnt[1];
        res->n_collision_before_best = 0;
        for (int r = 0; r < 8; ++r) res->reason_counts[r] = (int64_t)sh_cnt[2 + r];
        res->kernel_ms = 0.0;
    }
}

// infeasible_count_collision: colliding feasible samples that precede the winner in cost order
// (the lazy loop of reactive_planner.py:1031-1046 touches exactly those).  out += count.
__global__ __launch_bounds__(RP_BLOCK) void rp_count_before_kernel(const uint32_t *status, const double *cost, int64_t count,
                                                                   int64_t cand_begin, const rp_result *res,
                                                                   double wcost_in, int64_t widx_in, int use_args,
                                                                   unsigned long long *out) {
    if (!use_args && res->n_collision == 0) return;
    const double wc = use_args ? wcost_in : res->best_cost;
    const int64_t wi = use_args ? widx_in : res->best_index;
    int nloc = 0;
    for (int64_t i = (int64_t)blockIdx.x * RP_BLOCK + threadIdx.x; i < count; i += (int64_t)gridDim.x * RP_BLOCK) {
        if (RP_STATUS_LABEL(status[i]) == RP_LABEL_INFEASIBLE_COLLISION) {
            const double c = cost[i];
            nloc += (wi < 0 || c < wc || (c == wc && cand_begin + i < wi));
        }
    }
    __shared__ int sh;
    if (threadIdx.x == 0) sh = 0;
    __syncthreads();
    if (nloc) atomicAdd(&sh, nloc);
    __syncthreads();
    if (threadIdx.x == 0 && sh) atomicAdd(out, (unsigned long long)sh);
}

// plug-in costs: overwrite the cost of every candidate that has one (feasible / colliding)
__global__ void rp_apply_costs_kernel(const double *user, double *cost, const uint32_t *status, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t lab = RP_STATUS_LABEL(status[i]);
        if (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION) cost[i] = user[i];
    }
}
