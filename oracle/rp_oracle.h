/*
 * rp_oracle.h -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE, see rp_oracle.c).
 * Shares the POD types of the product ABI (include/rp_amd.h) so that tests feed both sides the
 * same structs.
 */
#ifndef RP_ORACLE_H
#define RP_ORACLE_H
#include "../include/rp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpo_tables {
    int32_t n_ref, n_sobb, n_tri, n_circ, n_dyn, n_steps, dyn_t0, reserved_;
    double proj_d_limit;
    const double *ref_pos, *ref_theta, *ref_curv, *ref_curv_d, *ref_x, *ref_y;
    const double *sobb, *tri, *circ, *dyn;
} rpo_tables;

/* status[C], cost[C], coeffs[C][13] (lon 6, lat 6, lat_T), states[C][14][N+1]: each may be NULL. */
int rp_oracle_plan(const rp_params *p, const rp_cost *cost, const rp_grids *g, const rpo_tables *tb,
                   int64_t cand_begin, int64_t cand_end, uint32_t *status, double *cost_out, double *coeffs,
                   double *states, rp_result *result, double *best_states, int nthreads);
int rp_oracle_plan_coeffs(const rp_params *p, const rp_cost *cost, const rpo_tables *tb, int64_t C,
                          const double *lon_coeffs, const double *lat_coeffs, const int32_t *traj_len,
                          uint32_t *status, double *cost_out, double *states, rp_result *result,
                          double *best_states, int nthreads);
int64_t rp_oracle_count_collisions_before(int64_t C, int64_t base, const uint32_t *status, const double *cost,
                                          double wcost, int64_t windex);
int rp_oracle_check_swept(const rp_params *p, const rpo_tables *tb, int n, const double *x, const double *y,
                          const double *theta, double *boxes);
void rp_oracle_check_poses(const rp_params *p, const rpo_tables *tb, int n, const double *x, const double *y, const double *theta,
                           int32_t *hit);
void rp_oracle_obb_sum_rows(int n_steps, const double *rows, double *out);
void rp_oracle_sample(const rp_params *p, const rp_grids *g, int64_t idx, double lon[6], double lat[6],
                      double *lat_T, int *traj_len);
void rpo_quintic_coeffs(double p0, double v0, double a0, double pf, double vf, double af, double T, double c[6]);
void rpo_quartic_coeffs(double p0, double v0, double a0, double T, double vd, double c[6]);
void rpo_vertex_tangents(int n, const double *x, const double *y, double *tx, double *ty);
double rpo_np_sum(const double *a, long n);

#ifdef __cplusplus
}
#endif
#endif
