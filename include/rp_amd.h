/*
 * rp_amd.h -- C ABI of the MI355X-native batch trajectory sampler/evaluator.
 *
 * One call (rp_plan) replaces, for one sampling level of one replanning cycle, the reference's
 *
 *   SamplingSpace.generate_trajectories_at_level      commonroad_rp/sampling.py:202-242
 *   Quartic/QuinticTrajectory coefficient solves       commonroad_rp/polynomial_trajectory.py:292-320,341-360
 *   ReactivePlanner._check_kinematics                  commonroad_rp/reactive_planner.py:715-969
 *   ReactivePlanner._check_constraints                 commonroad_rp/reactive_planner.py:971-1017
 *   CartesianSample.enlarge / CurviLinearSample.enlarge commonroad_rp/trajectories.py:168-197,302-332
 *   DefaultCostFunction(.FailSafe).evaluate            commonroad_rp/cost_function.py:51-71,82-92
 *   TrajectoryBundle.sort + _check_collisions          commonroad_rp/trajectories.py:502-510,
 *                                                      commonroad_rp/reactive_planner.py:1019-1063
 *   ReactivePlanner._get_optimal_trajectory            commonroad_rp/reactive_planner.py:1065-1136
 *
 * i.e. the two calls plan() makes per level (reactive_planner.py:620,624).  The reference is pure
 * Python and has no FFI of its own; the binding a maintainer adds is the ctypes stub shown in
 * INTEGRATION.md (shipped as commonroad_rp_amd/_capi.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every host buffer is caller-owned, C-contiguous, and only read
 *     or written during the call.  The library owns all device memory inside rp_ctx.
 *   - all functions return 0 on success and a negative RP_E* code on error; the message is
 *     available from rp_last_error().  Nothing throws across the ABI.  Numerical failures (NaN
 *     coefficients, leaving the projection domain) become per-candidate labels, never errors.
 *   - one rp_ctx per thread / HIP stream; calls on one ctx must be serialised by the caller.
 *     A ctx must not be used across fork() (the reference's debug.multiproc fan-out,
 *     reactive_planner.py:1084-1111, is what the GPU batch replaces).
 *   - candidate index = (iT * nL + iL) * nD + iD: the reference's list order
 *     (for t in T: for lon in L: for d in D, sampling.py:218-241) when the grids are passed in
 *     the reference's set-iteration order.
 */
#ifndef RP_AMD_H
#define RP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_ABI_VERSION 2   /* 2 (round 5): struct_size fields + RP_EABI; the plan entries folded into rp_plan / rp_plan_levels / rp_plan_coeffs
                              with a flags word; the collision path defaults to RP_COLLISION_AUTO (since round 4) */

/* error codes */
#define RP_OK 0
#define RP_EINVAL (-1)  /* bad argument / inconsistent sizes */
#define RP_EHIP (-2)    /* HIP runtime error (message has the hipError string) */
#define RP_ESTATE (-3)  /* call sequence error (e.g. rp_plan before rp_set_reference) */
#define RP_ENOMEM (-4)
#define RP_EDOMAIN (-5)    /* point outside the projection domain of the reference path */
#define RP_EDIRECTION (-6) /* the vehicle does not drive along the reference path (negative longitudinal velocity) */
#define RP_EABI (-7)       /* a struct's struct_size is not sizeof() of this library's struct: the caller was built against another
                              version of this header (every struct below starts with its size; RP_*_INIT set it) */

/* per-candidate status word:  label | reason << 4 | first_bad_step << 8 */
#define RP_LABEL_NONE 0u                 /* reference label None: pre-filtered or left the projection domain */
#define RP_LABEL_FEASIBLE 1u             /* FeasibilityStatus.FEASIBLE and (eagerly checked) collision-free */
#define RP_LABEL_INFEASIBLE_KINEMATIC 2u /* FeasibilityStatus.INFEASIBLE_KINEMATIC */
#define RP_LABEL_INFEASIBLE_COLLISION 3u /* kinematically feasible but colliding.  The reference marks these
                                            lazily (only samples cheaper than the winner).  Plans that ran the
                                            eager collision query (rp_last_path() == RP_PATH_EAGER) label every
                                            colliding feasible sample; plans that ran the cost-ordered stage
                                            (RP_PATH_LAZY) label the ones they had to look at -- at least every
                                            sample that sorts before the winner -- the others keep FEASIBLE. */
#define RP_STATUS_LABEL(s) ((s) & 3u)
#define RP_STATUS_REASON(s) (((s) >> 4) & 7u)
#define RP_STATUS_STEP(s) (((s) >> 8) & 0xFFFu)

/* first-failure reasons; 1..5 index infeasible_reason_dict (reactive_planner.py:799,803,981-1015) */
#define RP_REASON_NONE 0u
#define RP_REASON_VELOCITY 1u
#define RP_REASON_ACCELERATION 2u
#define RP_REASON_KAPPA 3u
#define RP_REASON_KAPPA_DOT 4u
#define RP_REASON_YAW_RATE 5u
#define RP_REASON_OUT_OF_DOMAIN 6u /* convert_to_cartesian_coords returned None (reactive_planner.py:910-917) */

/* constraints_to_check bit mask (config.py:127-128) */
#define RP_CHECK_VELOCITY (1u << 0)
#define RP_CHECK_ACCELERATION (1u << 1)
#define RP_CHECK_KAPPA (1u << 2)
#define RP_CHECK_KAPPA_DOT (1u << 3)
#define RP_CHECK_YAW_RATE (1u << 4)
#define RP_CHECK_ALL 0x1Fu

/* state block of one candidate: RP_N_ARRAYS rows of (N + 1) doubles, in this order
 * (CartesianSample then CurviLinearSample, trajectories.py:61-75,200-213) */
#define RP_N_ARRAYS 14
enum rp_array {
    RP_X = 0, RP_Y, RP_THETA, RP_V, RP_A, RP_KAPPA, RP_KAPPA_DOT,
    RP_S, RP_D, RP_THETA_CL, RP_S_DOT, RP_S_DDOT, RP_D_DOT, RP_D_DDOT
};

#define RP_LON_VELOCITY_KEEPING 0 /* quartic longitudinal, L = target velocities (sampling.py:254-258) */
#define RP_LON_STOPPING 1         /* quintic longitudinal, L = target positions  (sampling.py:259-263) */

#define RP_COST_DEFAULT 0  /* DefaultCostFunction          cost_function.py:35-71 */
#define RP_COST_FAILSAFE 1 /* DefaultCostFunctionFailSafe  cost_function.py:74-92 */
#define RP_COST_EXTERNAL 2 /* plug-in cost: states are materialised, costs come back through rp_select */

/* flags in rp_params.flags */
#define RP_FLAG_DRAW_ALL (1u << 0)        /* _draw_traj_set: no pre-filter, no early exit (reactive_planner.py:796,903) */
#define RP_FLAG_MATERIALIZE_ALL (1u << 1) /* keep every candidate's state block on the device (rp_fetch_states) */
#define RP_FLAG_SKIP_COLLISION (1u << 2)  /* do not test against the obstacle tables */

typedef struct rp_params {
    uint32_t struct_size;   /* sizeof(rp_params), set by the caller (RP_PARAMS_INIT): checked by every entry, RP_EABI on mismatch */
    int32_t N;              /* planning.time_steps_computation; arrays have N + 1 entries */
    int32_t factor;         /* planning.factor (collision time index = time_step0 + i * factor) */
    int32_t time_step0;     /* x_0.time_step */
    int32_t low_vel_mode;   /* reactive_planner.py:594 */
    int32_t lon_mode;       /* RP_LON_* */
    uint32_t constraint_mask; /* RP_CHECK_* */
    uint32_t flags;         /* RP_FLAG_* */
    double dt;              /* planning.dt */
    double x0_lon[3];       /* s, s_dot, s_ddot */
    double x0_lat[3];       /* d, d_dot, d_ddot (derivatives w.r.t. s in low-velocity mode) */
    double x0_orientation;  /* x_0.orientation (standstill branch, reactive_planner.py:866) */
    double wheelbase, wb_rear_axle, length, width;
    double a_max, v_switch, delta_max, v_delta_max;
} rp_params;

typedef struct rp_cost {
    uint32_t struct_size; /* sizeof(rp_cost) */
    int32_t kind;         /* RP_COST_* */
    double w_a;           /* DefaultCostFunction.w_a (5, or 1 in stopping mode) */
    double desired_speed; /* NaN = None */
    double desired_d;
    double desired_s;     /* NaN = None */
} rp_cost;

typedef struct rp_grids {
    uint32_t struct_size;    /* sizeof(rp_grids) */
    int32_t nT, nL, nD;
    const double *T;         /* [nT] durations, reference iteration order */
    const int32_t *traj_len; /* [nT] len(np.arange(0, round(T + dt, 5), dt)) (reactive_planner.py:733,748) */
    const double *L;         /* [nL] longitudinal samples (stopping mode: already filtered by
                                filter_goals_behind, trajectories.py:545-550) */
    const double *D;         /* [nD] lateral samples including the appended d0 (sampling.py:226) */
} rp_grids;

typedef struct rp_result {
    uint32_t struct_size;     /* sizeof(rp_result), set by the CALLER before the call (RP_RESULT_INIT): the library writes no more than that */
    uint32_t reserved_;
    int64_t best_index;       /* winner = lexicographic min over (cost, index) among feasible,
                                 collision-free candidates of this call's range; -1 if none */
    double best_cost;         /* NaN if none */
    int64_t n_candidates;     /* candidates evaluated by this call (cand_end - cand_begin) */
    int64_t n_feasible;       /* kinematically feasible (colliding ones included):
                                 infeasible_count_kinematics = n_candidates - n_feasible */
    int64_t n_collision_before_best; /* infeasible_count_collision (lazy semantics of _check_collisions) */
    int64_t n_collision;      /* colliding feasible candidates FOUND: all of them after an eager plan, the ones the
                                 cost-ordered stage looked at (>= n_collision_before_best) after a lazy one */
    int64_t reason_counts[8]; /* [RP_REASON_*]: 1..5 = infeasible_reason_dict, 6 = out of domain */
    double best_lon_coeffs[6];
    double best_lat_coeffs[6];
    double best_lat_T;        /* delta_tau of the winner's lateral polynomial (low-velocity mode: a distance) */
    double kernel_ms;         /* device time of this call's kernels (HIP events on the ctx stream);
                                 0 unless profiling is enabled */
} rp_result;

#define RP_PARAMS_INIT {(uint32_t)sizeof(rp_params)}
#define RP_COST_INIT {(uint32_t)sizeof(rp_cost)}
#define RP_GRIDS_INIT {(uint32_t)sizeof(rp_grids)}
#define RP_RESULT_INIT {(uint32_t)sizeof(rp_result)}

typedef struct rp_ctx rp_ctx;

/* ---- life cycle ------------------------------------------------------------------------------ */
int rp_abi_version(void);
/* first 16 hex digits of the SHA-256 of the sources this library was built from (csrc/Makefile): measurement files name
   the code they were taken on */
const char *rp_source_hash(void);
int rp_create(rp_ctx **out, int device);
void rp_destroy(rp_ctx *ctx);
const char *rp_last_error(const rp_ctx *ctx);
/* enable = k > 0: record HIP events around the evaluation kernel of every k-th rp_plan on this ctx
 * (rp_result.kernel_ms; 0 for the calls in between); 0 = off */
int rp_set_profiling(rp_ctx *ctx, int enable);
/* How the host thread waits for a plan's result (the kernels write it, and a completion ticket behind it, into pinned host memory):
 *   RP_WAIT_SPIN   (default) poll the ticket with `pause` in between: the result is seen ~1 us after it lands, one host core is busy
 *                  for the length of the plan;
 *   RP_WAIT_YIELD  poll, but give the core away between polls (sched_yield): for more ranks than cores -- eight ranks under a
 *                  16-CPU quota leave the binding's own work little else (commonroad_rp_amd.distributed picks it when
 *                  cpu quota < 2 x local world size);
 *   RP_WAIT_EVENT  no ticket: an event behind the plan's last launch, the thread sleeps in hipEventSynchronize (the driver's
 *                  interrupt path: some 10 us later than the ticket, no core burnt).
 * Every mode falls back to a stream synchronisation after 200 ms without a ticket.  Results are the same in every mode. */
#define RP_WAIT_SPIN 0
#define RP_WAIT_YIELD 1
#define RP_WAIT_EVENT 2
int rp_set_wait_mode(rp_ctx *ctx, int mode);
int rp_get_wait_mode(const rp_ctx *ctx);
/* Launch-policy switches of a context, by name (tests, A/B measurements; a planner has no need for them).  The environment variable
 * named with each key is read ONCE, by rp_create, as the context's default; no call that plans reads the environment.
 *   "lanes"             RP_AMD_G                    0 by batch | 16 | 32 | 64 lanes per candidate of rp_eval_kernel
 *   "eval_block"        RP_AMD_EVAL_BLOCK           0 by batch | 64 | 256 threads per workgroup of the batch's rp_eval_kernel launch
 *   "cost_kernel"       RP_AMD_COST_KERNEL          -1 by batch | 0 never | 1 whenever rp_cost_kernel applies
 *   "chunk_kernel"      RP_AMD_CHUNK_KERNEL         -1 by batch | 0 never | 1 whenever rp_chunk_kernel applies
 *   "lazy"              RP_AMD_LAZY                 -1 by rp_set_collision_path | 0 never | 1 cost-ordered stage whenever the launch path allows
 *   "sweep"             RP_AMD_SWEEP                -1 by batch | 0 | 1: the cost-ordered stage as a bounded sweep (RP_PATH_SWEEP) instead of list rounds
 *   "fused_lon"         RP_AMD_NO_FUSED_LON         1 | 0: single-launch variant of small batches
 *   "fused_lon_blocks"  RP_AMD_FUSED_LON_BLOCKS     -1 (4 workgroups per CU) | largest grid that takes it
 *   "auto_materialize"  RP_AMD_NO_AUTO_MATERIALIZE  1 | 0: small batches whose winner rows are wanted write every candidate's rows
 *   "stage_out", "row_padding", "row_align" (0 | 8 | 16), "tail_split"   RP_AMD_NO_STAGE_OUT, _NO_ROW_PADDING, _ROW_ALIGN, _NO_TAIL_SPLIT
 *   "table_window", "lon_publish", "inline_grids", "zero_copy", "coeff_groups", "winner_skip_query"   RP_AMD_NO_<NAME>: 1 | 0
 *   "fold_threshold"    RP_AMD_FOLD_THRESHOLD       block partials beyond which they are folded before the epilogue
 *   "event_bracket", "winner_lanes_as_batch", "lazy_trace", "print_stamps", "timing"   RP_AMD_<NAME>: 0 | 1 (measurement variants, diagnostics on stderr)
 *   "wait_mode"         RP_AMD_WAIT_MODE            RP_WAIT_* (as rp_set_wait_mode)
 *   "wait_fallbacks"    (read-only, rp_get_option) waits for a completion ticket that ended in the 200-ms fall-back: 0 unless a kernel
 *                       chain failed to hand its ticket over
 * RP_EINVAL: unknown key or value out of range; RP_ESTATE: a plan is in flight. */
int rp_set_option(rp_ctx *ctx, const char *key, int64_t value);
int rp_get_option(const rp_ctx *ctx, const char *key, int64_t *value);

/* ---- tables (once per reset / reference path) -------------------------------------------------- */
/* CoordinateSystem tables (utils_coordinate_system.py:114-118) + polyline vertices (.reference). */
int rp_set_reference(rp_ctx *ctx, int32_t n, const double *ref_pos, const double *ref_theta,
                     const double *ref_curv, const double *ref_curv_d, const double *ref_x,
                     const double *ref_y, double proj_domain_d_limit);
/* Obstacle content of the collision checker (reactive_planner.py:234-251) as flat tables:
 * sobb[n_sobb][5] = cx,cy,theta,half_l,half_w; tri[n_tri][6]; circ[n_circ][3] = cx,cy,r;
 * dyn[n_dyn][n_steps][5] for scenario time steps dyn_t0 .. dyn_t0+n_steps-1 (cx = NaN: absent). */
int rp_set_obstacles(rp_ctx *ctx, int32_t n_sobb, const double *sobb, int32_t n_tri, const double *tri,
                     int32_t n_circ, const double *circ, int32_t n_dyn, int32_t n_steps, int32_t dyn_t0,
                     const double *dyn);

/* ---- the hot path ------------------------------------------------------------------------------ */
/* Three entries (round 5; rounds 1-4 had nine -- their names remain as inline wrappers at the end of this header):
 *   rp_plan         one sampling level: candidates [cand_begin, cand_end) of the nT*nL*nD grid (cand_end < 0: all)
 *   rp_plan_levels  the LEVEL LOOP of plan() in one call
 *   rp_plan_coeffs  explicit polynomials of a foreign sampling space
 * and rp_plan_wait, the second half of a call made with RP_PLAN_BEGIN.  `flags`:
 *   RP_PLAN_BEGIN   validate, stage, put the kernels of the plan on the context's stream and return (result and out may be NULL);
 *                   rp_plan_wait waits for the completion ticket the last kernel writes into pinned host memory and unpacks the result.
 *                   Between the two the host is free -- the replanning loop packs the previous cycle's output and does its
 *                   bookkeeping.  One plan in flight per context; the calls that read "the last plan" refer to the last COLLECTED
 *                   one.  Plans whose chain needs decisions of the host in between (the list rounds of the cost-ordered collision
 *                   stage of large batches) run them inside the call.
 *   RP_PLAN_ROWS    with RP_PLAN_BEGIN: the winner's state block will be asked for by rp_plan_wait (without RP_PLAN_BEGIN a non-NULL
 *                   `out` says so)
 *   RP_PLAN_PACKED  for a binding whose per-argument cost matters (ctypes: ~0.5 us per array argument): the grids come from a buffer
 *                   of the context the caller has filled -- rp_fast_buffer hands it out once, 32 KB, valid for the life of the
 *                   context: per level [T (nT doubles) | L (nL) | D (nD) | traj_len (nT int32, padded to 8 bytes)], the levels
 *                   one behind the other -- grids[k] holds the sizes only (T, traj_len, L, D NULL); and `out`
 *                   ([RP_N_ARRAYS + 13][N + 1] doubles) receives, behind the winner's state block [RP_N_ARRAYS][N + 1], the packed
 *                   output of rp_pack_trajectory [N + 1][13].  Whole grids only (cand_begin 0, cand_end -1).
 * out: NULL or [RP_N_ARRAYS][N + 1] receiving the winner's state block (RP_PLAN_PACKED: see there). */
#define RP_PLAN_BEGIN (1u << 0)
#define RP_PLAN_ROWS (1u << 1)
#define RP_PLAN_PACKED (1u << 2)
int rp_plan(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, const rp_grids *grids,
            int64_t cand_begin, int64_t cand_end, uint32_t flags, rp_result *result, double *out);
int rp_plan_wait(rp_ctx *ctx, rp_result *result, double *out);

/* The LEVEL LOOP of plan() (commonroad_rp/reactive_planner.py:616-636: `while optimal_trajectory is None and i < sampling_level`,
 * one _create_trajectory_bundle + _get_optimal_trajectory per sampling level) in one call and ONE device round trip: grids[k] are the
 * sample grids of the levels in the order the loop would visit them.  The kernels of every level go onto the stream back to back;
 * the epilogue of a level that finds a winner closes a gate in device memory, and the kernels of the levels behind it leave at
 * once -- the result is the one of the first level that has a winner (or of the last level), as if the levels had been planned one
 * after the other with rp_plan.  *level: index into grids of the level the result belongs to (also rp_last_level); the calls that
 * read "the last plan" (rp_fetch_status, rp_cost_range, rp_eval_one, ...) refer to that level.
 * Levels of up to 16 384 candidates whose grids fit the launch block (96 doubles: the reference's sampling levels hold at most
 * 29 + 17 + 18 samples) ride in the chain; a level that does not, and the levels behind it, are planned one by one inside the call.
 * flags as rp_plan (RP_PLAN_BEGIN: *level through rp_last_level after rp_plan_wait). */
int rp_plan_levels(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int32_t n_levels, const rp_grids *grids, uint32_t flags,
                   rp_result *result, double *out, int32_t *level);
int rp_last_level(const rp_ctx *ctx);

/* Generic entry for foreign SamplingSpace plug-ins (sampling.py:165-175): the polynomials come
 * from the plug-in's TrajectorySample objects.  lon_coeffs/lat_coeffs: [C][6]; lon_T: [C]
 * delta_tau of each polynomial; traj_len: [C].  Candidate index = list index.
 * Candidates that come in groups with a common longitudinal polynomial (and traj_len) -- all lateral samples of one (time, velocity)
 * sample of a corridor level -- may say so: n_groups > 0, group[C] ascending without gaps from 0, the candidates of a group adjacent,
 * group_first[n_groups]; the rows of a group's first candidate stand for the group (one longitudinal profile per group on the device
 * instead of one per candidate).  Groups only with arrays out of the context's arena (rp_coeffs_arena, rp_coeffs_arena_groups);
 * n_groups == 0 or group == NULL: one profile per candidate.  Results are the same either way. */
int rp_plan_coeffs(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int64_t C, const double *lon_coeffs,
                   const double *lat_coeffs, const double *lon_T, const int32_t *traj_len, int64_t n_groups,
                   const int32_t *group, const int32_t *group_first, rp_result *result, double *best_states);
/* Pinned host arrays for the explicit polynomials of up to `cap` candidates, owned by the context: lon_coeffs [cap][6],
 * lat_coeffs [cap][6], traj_len [cap].  A sampling space that writes its candidates straight into them (rp_corridor_coeffs
 * with these as its outputs; CorridorSampling.generate_trajectories_at_level, sampling.py:340-397, builds one object per
 * candidate instead) and hands the very same pointers to rp_plan_coeffs saves it the copy into its own staging buffer and the
 * transfers to the device -- the kernels read the pinned arrays themselves, every value once per plan (or per round of the
 * cost-ordered stage).  The arrays stay
 * valid until rp_coeffs_arena is called with a larger cap (which frees them and hands out new ones) or rp_destroy; a call with
 * cap <= the current capacity returns the same arrays.  A plan that took its candidates from the arena reads the winner's
 * coefficients back from it (rp_result.best_*_coeffs, also in a later rp_select): leave the rows alone until the plan's results
 * have been collected -- writing the next level's candidates is what ends a level anyway. */
int rp_coeffs_arena(rp_ctx *ctx, int64_t cap, double **lon_coeffs, double **lat_coeffs, int32_t **traj_len);
/* Two more arrays of the arena handed out last, int32 [cap] each: room for the group of every candidate and the first candidate of
   every group (rp_corridor_coeffs_grouped writes them, rp_plan_coeffs reads them). */
int rp_coeffs_arena_groups(rp_ctx *ctx, int32_t **group, int32_t **group_first);

/* ONE call per replanning cycle (round 5): what ReactivePlanner.plan() does between its entry and the optimal trajectory
 * (commonroad_rp/reactive_planner.py:570-665) for the standard sampling space, so that a binding makes one foreign call per cycle:
 *   1. the curvilinear initial state: taken from `cycle` (reset() was handed it, :586-588 skipped) or worked out from the Cartesian
 *      state (_compute_initial_states, :446-512 -- rp_initial_state on the tables of rp_set_reference) with the low-velocity flag of
 *      the PREVIOUS cycle (`prev_low_vel_mode`: the reference sets the flag of this cycle only afterwards, :594);
 *   2. low_vel_mode = velocity < low_vel_mode_threshold (:594);
 *   3. the grids of the levels: T, longitudinal samples and traj_len as staged by the caller in the context's buffer (rp_fast_buffer,
 *      layout of RP_PLAN_PACKED: per level [T | L | D | traj_len]); the lateral samples D there are the INSERTION sequence of the
 *      level's set -- np.linspace(d_min, d_max, n), sampling.py:95-99 -- and the call appends the current offset x0_lat[0] the way
 *      `samples_d.union({x_0_lat[0]})` does (sampling.py:226: iteration order of a CPython set, csrc/rp_corridor.h); in stopping
 *      mode longitudinal samples that are not ahead of the vehicle are dropped (filter_goals_behind, trajectories.py:545-550);
 *   4. the level loop (rp_plan_levels) and the output packing (rp_pack_trajectory) -- `out` as for RP_PLAN_PACKED.
 * `params` is the template of the cycle: everything but x0_lon, x0_lat, x0_orientation and low_vel_mode, which the call fills in.
 * Errors of step 1: RP_EDOMAIN / RP_EDIRECTION (as rp_initial_state). */
typedef struct rp_cycle {
    uint32_t struct_size;        /* sizeof(rp_cycle) */
    int32_t have_curvilinear;    /* in: 1 = x0_lon / x0_lat below are the state; 0 = from the Cartesian state */
    int32_t prev_low_vel_mode;   /* in: the planner's low-velocity flag before this cycle (step 1) */
    int32_t low_vel_mode;        /* out */
    int32_t level;               /* out: index (into dims) of the level the result belongs to */
    int32_t reserved_;
    double x, y, orientation, velocity, acceleration, steering_angle;   /* in: x_0 (rear axle) */
    double low_vel_mode_threshold;                                     /* in: planning.low_vel_mode_threshold */
    double x0_lon[3], x0_lat[3];                                        /* in (have_curvilinear) / out */
} rp_cycle;
#define RP_CYCLE_INIT {(uint32_t)sizeof(rp_cycle)}
int rp_plan_cycle(rp_ctx *ctx, rp_cycle *cycle, const rp_params *params, const rp_cost *cost, int32_t n_levels,
                  const rp_grids *dims /* [n_levels]: sizes of the staged arrays */, rp_result *result,
                  double *out /* [(RP_N_ARRAYS + 13) * (N + 1)] */);

/* ---- results of the last rp_plan / rp_plan_coeffs on this ctx ---------------------------------- */
/* How the last plan answered the collision query (reactive_planner.py:1019-1063):
 *   RP_PATH_EAGER          every pose of every candidate inside the evaluation kernel (draw / materialising plans, small
 *                          batches, plug-in costs);
 *   RP_PATH_LAZY           the reference's own order of work: costs first, then the candidates in ascending (cost, index)
 *                          until the first free one -- production-mode plans of large batches with obstacles;
 *   RP_PATH_LAZY_FALLBACK  the cost-ordered stage ran out of candidates (a scene where nearly everything collides) and the
 *                          eager kernel decided; results as RP_PATH_EAGER.
 * Winner, cost, infeasible_count_collision and the kinematic counters are the same on every path; what differs are the labels of
 * colliding candidates BEHIND the winner (RP_LABEL_INFEASIBLE_COLLISION above) and rp_result.n_collision.
 * Which path a production-mode plan of a large batch takes is a setting of the context (rp_set_collision_path):
 *   RP_COLLISION_AUTO          (default) cost-ordered, except that a stage that ran out of candidates or needed its third and last
 *                              list to find the winner keeps the next 1, 2, 4 .. 64 eligible plans eager.  A rule over the
 *                              SEQUENCE OF PLANS: the same calls label the same candidates in every run.
 *   RP_COLLISION_EAGER         always the eager query: every colliding candidate is labelled (what rp_select needs).
 *   RP_COLLISION_COST_ORDERED  always the cost-ordered stage where the launch path allows it (eager fallback when it runs dry).
 *   RP_COLLISION_TIMED         the context's own clock decides: it times its whole-call plans (rp_plan, rp_plan_packed,
 *                              rp_plan_coeffs) per path, takes the faster one and measures the other again every 64th plan.
 *                              Fastest on average; which candidates behind the winner carry a collision label then depends on
 *                              earlier timings -- read rp_last_path() before interpreting them.
 * The option "lazy" = 0 / 1 (rp_set_option; RP_AMD_LAZY as a context's default) pins the choice (tests, measurements). */
#define RP_PATH_EAGER 0
#define RP_PATH_LAZY 1
#define RP_PATH_LAZY_FALLBACK 2
#define RP_PATH_SWEEP 3   /* cost-ordered, as one bounded sweep: costs of every candidate first, then the eager query over the batch for the
                             candidates no cheaper free one rules out -- same guarantees as RP_PATH_LAZY (every candidate in front of the
                             winner is labelled); rp_result.n_collision == n_collision_before_best */
int rp_last_path(const rp_ctx *ctx);
/* Which kernel evaluated the batch of the last plan (diagnostic: what a measurement prices the plan against).
 *   RP_KERNEL_EVAL  rp_eval_kernel: 16 / 32 / 64 lanes per candidate, a lane per time step (every plan that keeps state rows)
 *   RP_KERNEL_COST  rp_cost_kernel: one lane per candidate walking the steps (large batches that keep costs and labels only)
 * Results are the same bits either way. */
#define RP_KERNEL_EVAL 0
#define RP_KERNEL_COST 1
#define RP_KERNEL_CHUNK 2   /* rp_chunk_kernel: one lane per candidate and step block of 16 steps (mid-sized batches that keep costs and labels only) */
int rp_last_kernel(const rp_ctx *ctx);
#define RP_COLLISION_AUTO 0
#define RP_COLLISION_EAGER 1
#define RP_COLLISION_COST_ORDERED 2
#define RP_COLLISION_TIMED 3
int rp_set_collision_path(rp_ctx *ctx, int mode);
/* status[count], cost[count] (NaN where no cost) for local candidates first .. first+count-1
 * (local = relative to cand_begin).  Either pointer may be NULL. */
int rp_fetch_status(rp_ctx *ctx, int64_t first, int64_t count, uint32_t *status, double *cost);
/* state blocks [count][RP_N_ARRAYS][N + 1]; needs RP_FLAG_MATERIALIZE_ALL in the last plan. */
int rp_fetch_states(rp_ctx *ctx, int64_t first, int64_t count, double *states);
/* state block of one (global) candidate index of the last plan, re-evaluated on the device. */
int rp_eval_one(rp_ctx *ctx, int64_t index, double *states, uint32_t *status, double *cost);
/* Multi-GPU second pass: number of colliding feasible local candidates that precede the global
 * winner (cost, index) in the reference's sorted order.  After a cost-ordered plan (rp_last_path() == RP_PATH_LAZY) the labels are
 * complete up to the plan's own winner: a key that sorts behind it is refused with RP_ESTATE. */
int rp_count_collisions_before(rp_ctx *ctx, double cost, int64_t index, int64_t *count);
/* Plug-in cost functions (RP_COST_EXTERNAL): upload one cost per local candidate (NaN = skip) and
 * redo the selection (argmin + collision counters) on the device.  Needs a plan that labelled every colliding candidate: after a
 * cost-ordered plan (rp_last_path() == RP_PATH_LAZY) the call returns RP_ESTATE (plans with RP_COST_EXTERNAL or
 * RP_FLAG_MATERIALIZE_ALL -- what a plug-in cost needs anyway -- always take the eager query). */
int rp_select(rp_ctx *ctx, const double *costs, int64_t count, rp_result *result, double *best_states);

/* Smallest and largest cost among the local candidates of the last plan that have one (the kinematically feasible ones,
 * labels FEASIBLE and INFEASIBLE_COLLISION) and their number: what TrajectoryBundle.min_costs() / max_costs() return once
 * _get_optimal_trajectory has left the feasible samples in the bundle (commonroad_rp/reactive_planner.py:1128; read by the
 * standstill branch of plan(), :650-651).  n == 0: *min_cost and *max_cost are NaN. */
int rp_cost_range(rp_ctx *ctx, double *min_cost, double *max_cost, int64_t *n);

/* Continuous collision check of ONE trajectory -- the second test of ReactivePlanner._check_collisions
 * (commonroad_rp/reactive_planner.py:1049-1058, planning.continuous_collision_check), which the reference applies
 * to the first candidate of the sorted list that passed the per-pose test: the ego rectangles (half extents
 * length/2, width/2, centre = (x, y) moved by wb_rear_axle along theta, :1033-1035,1052) of poses i and i + 1 are
 * replaced by the tight rectangle around both (commonroad-dc's trajectory_preprocess_obb_sum; see DESIGN.md for
 * the definition used) and segment i is tested against the obstacle tables at scenario time index
 * params->time_step0 + i (the reference's ego object starts at x_0.time_step and holds one shape per index;
 * planning.factor does not enter, :1050).
 * x, y, theta: [n_poses] rear-axle poses (cartesian.x / .y / .theta of the sample).  first_hit: smallest colliding
 * segment, -1 if none (also for n_poses < 2).  boxes (optional): [n_poses - 1][6] = cx, cy, ux, uy, half length,
 * half width of every segment's rectangle.  Uses the obstacle tables of rp_set_obstacles. */
int rp_check_swept(rp_ctx *ctx, const rp_params *params, int32_t n_poses, const double *x, const double *y,
                   const double *theta, int32_t *first_hit, double *boxes);

/* ---- output packing (host only; no GPU involved) ---------------------------------------------------
   What ReactivePlanner._compute_trajectory_pair makes of the optimal trajectory's arrays (commonroad_rp/reactive_planner.py:
   514-568), in one pass over the winner's state block [RP_N_ARRAYS][n]: out[n][13] = position x, y | curvilinear position s, d |
   lon sample s, s', s'' | lat sample d, d', d'' | orientation shifted into [x0_orientation - pi, x0_orientation + pi]
   (shift_orientation, utility/general.py:49-55) | steering angle atan2(wheelbase * kappa, 1) (:539) | yaw rate
   (theta[i] - theta[i-1]) / dt (:535; entry 0 is 0: state 0 carries x_0.yaw_rate). */
int rp_pack_trajectory(int32_t n, const double *states, double dt, double wheelbase, double x0_orientation, double *out);
/* The grid buffer of RP_PLAN_PACKED calls (rp_plan above): handed out once, 32 KB, valid for the life of the context. */
int rp_fast_buffer(rp_ctx *ctx, void **ptr, size_t *bytes);

/* ---- adaptive sampling space (host only; no GPU involved) -----------------------------------------
   rp_corridor_coeffs: the candidates CorridorSampling.generate_trajectories_at_level returns (commonroad_rp/sampling.py:
   345-397), as the coefficient arrays rp_plan_coeffs takes.  Per time sample k (in the iteration order of the level's time
   set): duration T[k], traj_len[k], the corridor's velocity interval [v_low[k], v_up[k]] at that time step (:363-364) and its reach
   nodes boxes[box_off[k] .. box_off[k+1])[6] = p_lon_min, p_lon_max, p_lat_min, p_lat_max, v_lon_min, v_lon_max.  For every
   velocity of set(np.linspace(v_low, v_up, n_samples)) (:367): quartic to (v, 0) (:368), its end position (:369), the nodes that
   contain it (:374-375), their connected parts (lateral intervals that overlap or touch, :378), per part the lateral samples
   set(np.linspace(lo, hi, n_samples)), with the reference path added where the interval straddles it (:382-386), per lateral sample
   a quintic to (d, 0, 0) over T (:390-392).  Order = the reference's: iteration order of CPython sets of floats (restated,
   csrc/rp_corridor.h; rp_pyset_order exposes it: the values of set(values), or of set(values).union({0}), in iteration order).
   Output rows 0 .. *count - 1; with RP_ENOMEM *count is the room needed (cap too small).  The time samples are worked out on a
   small pool of host threads of the library (csrc/rp_pool.h; RP_AMD_HOST_THREADS, default 8, 1 = the calling thread only) and
   joined in the reference's order; concurrent callers take turns. */
int rp_corridor_coeffs(int32_t nT, const double *T, const int32_t *traj_len, const double *v_low, const double *v_up, const int32_t *box_off,
                       const double *boxes, int32_t n_samples, const double *x0_lon, const double *x0_lat, int64_t cap, int64_t *count,
                       double *lon_coeffs, double *lat_coeffs, double *lon_T, int32_t *traj_len_out, double *lon_end, double *lat_end);
/* The same, and which candidates share their longitudinal polynomial: the lateral samples of one (time, velocity) sample are
   adjacent rows and form a group.  group[cap]: number of a candidate's group (0, 1, ... in order of appearance); group_first[cap]:
   first candidate of a group; *n_groups.  What rp_plan_coeffs_grouped takes (one longitudinal profile per group on the device
   instead of one per candidate: `share work across d`, as the grid plans do).  Any of the three may be NULL. */
int rp_corridor_coeffs_grouped(int32_t nT, const double *T, const int32_t *traj_len, const double *v_low, const double *v_up,
                               const int32_t *box_off, const double *boxes, int32_t n_samples, const double *x0_lon, const double *x0_lat,
                               int64_t cap, int64_t *count, double *lon_coeffs, double *lat_coeffs, double *lon_T, int32_t *traj_len_out,
                               double *lon_end, double *lat_end, int32_t *group, int32_t *group_first, int64_t *n_groups);
int rp_pyset_order(int32_t n, const double *values, int32_t union_zero, double *out /* [n + 1] */, int32_t *n_out);

/* ---- reference-path front end (host only; no GPU involved) ---------------------------------------
   rp_build_reference: what CoordinateSystem.__init__ makes of a route centre line
   (commonroad_rp/utility/utils_coordinate_system.py:88-118): duplicate vertices removed (:95-96), with `smooth`
   the cubic-spline smoothing of smooth_ref_path (:74-83: interpolating cubic B-spline over the chord-length
   parameter, 200 samples, equidistant resampling at `resample_step` metres) and duplicates removed again (:103-104),
   then the tables path length, unwrapped orientation, curvature and curvature rate (:114-117) -- the inputs of
   rp_set_reference.  Output arrays hold `cap` vertices; *n_out is the number written, or, with RP_ENOMEM, the room
   needed.  rp_project: (x, y) -> (s, d), what CoordinateSystem.convert_to_curvilinear_coords returns (:176-178;
   RP_EDOMAIN where the reference raises ValueError).  rp_initial_state: ReactivePlanner._compute_initial_states
   (commonroad_rp/reactive_planner.py:446-512): Frenet state x0_lon = (s, s', s''), x0_lat = (d, d', d'') of a
   Cartesian rear-axle state; derivatives of d w.r.t. arc length in low-velocity mode (:503-506), w.r.t. time
   otherwise; RP_EDIRECTION where the reference raises for a negative longitudinal velocity (:492-494).
   The polyline / projection geometry behind these (commonroad_dc.geometry.util, pycrccosy) is this build's own
   definition; SciPy's splprep / splev and NumPy's unique / unwrap / gradient are restated. */
int rp_build_reference(int32_t n_in, const double *xy_in /* [n_in][2] */, int32_t smooth, double resample_step, int32_t cap,
                       int32_t *n_out, double *xy_out /* [cap][2] */, double *ref_pos, double *ref_theta, double *ref_curv,
                       double *ref_curv_d);
int rp_project(int32_t n, const double *ref_xy /* [n][2] */, const double *ref_pos, double proj_domain_d_limit, double x, double y,
               double *s, double *d);
int rp_initial_state(int32_t n, const double *ref_xy, const double *ref_pos, const double *ref_theta, const double *ref_curv,
                     const double *ref_curv_d, double proj_domain_d_limit, double x, double y, double orientation, double velocity,
                     double acceleration, double steering_angle, double wheelbase, int32_t low_vel_mode, double *x0_lon /* [3] */,
                     double *x0_lat /* [3] */);

/* ---- multi-GPU winner exchange on the device ----------------------------------------------------
   Candidate ranges sharded over GPUs (one process per GPU): every rank's rp_plan leaves its result block -- header +
   winner state rows -- in device memory.  rp_result_device hands out its address and size, so that the caller can pass
   it to a collective as it is (torch.distributed / RCCL all_gather_into_tensor: no host packing, no copies);
   rp_result_device waits for the context's stream first, so the block may be read by work enqueued on any stream afterwards.
   rows_valid = 0 means the winner's rows exist on the host only (large batches in non-materialising mode, where the
   winner is re-evaluated straight into host memory); the header says so, the block is gathered all the same.
   rp_combine_results takes the gathered blocks [world][bytes] (device memory), enqueues the combining kernel on
   `stream` (a hipStream_t; the stream the collective was enqueued on, so that it runs after it; NULL: the context's)
   and returns the global result: winner = lexicographic (cost, index) minimum over the ranks, counters summed, winner's
   coefficients and state rows, owner rank (-1: no winner).  *rows_valid = 0: the OWNER's rows were on its host only --
   best_states is not written, every rank sees the same flag and the caller exchanges host-packed messages for this step.
   n_collision_before_best is 0: that count depends on the global winner and is a second message
   (rp_count_collisions_before + a sum).  Replaces the fan-in of ReactivePlanner._get_optimal_trajectory's
   multiprocessing.Queue (commonroad_rp/reactive_planner.py:1084-1111). */
int rp_result_device(rp_ctx *ctx, const void **ptr, size_t *bytes, int32_t *rows_valid);
int rp_combine_results(rp_ctx *ctx, const void *d_msgs, int32_t world, void *stream, rp_result *global, double *best_states,
                       int32_t *owner_rank, int32_t *rows_valid);

/* ---- intra-node winner exchange (host only; no GPU involved) -----------------------------------
   Replaces the reference's only "communication backend", the multiprocessing.Queue fan-out of
   ReactivePlanner._get_optimal_trajectory (commonroad_rp/reactive_planner.py:1084-1111), for candidate ranges
   sharded over the GPUs of one node.  Every rank's rp_plan result already sits in host memory, so the ranks
   post it to a shared-memory region (POSIX shm mapped by every rank; the caller creates, zero-fills and maps
   it) and combine: global (cost, index) min-loc, summed counters, the winner's coefficients and state block.
   `seq` must increase by one per exchange, the same on every rank.  rp_mailbox_sum is the second pass
   (colliding candidates before the global winner).  Both spin until every rank has posted; return RP_OK, or
   RP_ESTATE when some peer has not posted within the wait budget (per peer; 30 s unless set through
   rp_mailbox_set_timeout or the environment variable RP_AMD_MAILBOX_TIMEOUT_S, read once) -- rp_mailbox_stalled_rank
   then names the peer this thread was waiting for.  After a time-out the sequence numbers of the ranks no longer
   agree: the region must not be used again (the caller tears the group down). */
size_t rp_mailbox_bytes(int32_t world, int32_t n_steps /* N + 1 */);
int rp_mailbox_exchange(void *region, int32_t world, int32_t rank, uint64_t seq, int32_t n_steps, const rp_result *local,
                        const double *local_best_states /* [14][n_steps] or NULL */, rp_result *global,
                        double *global_best_states /* [14][n_steps] or NULL */, int32_t *owner_rank);
int rp_mailbox_sum(void *region, int32_t world, int32_t rank, uint64_t seq, int32_t n_steps, int64_t value, int64_t *total);
int rp_mailbox_set_timeout(double seconds);   /* wait budget per peer, process-wide; seconds > 0 */
int rp_mailbox_stalled_rank(void);            /* peer the calling thread's last failed wait was spinning on; -1: none */

/* ---- the plan entries of ABI version 1 (rounds 1-4), as wrappers over the three above: source compatibility only, nothing exported */
static inline int rp_plan_begin(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, const rp_grids *grids, int64_t cand_begin,
                                int64_t cand_end, int32_t want_best_states) {
    return rp_plan(ctx, params, cost, grids, cand_begin, cand_end, RP_PLAN_BEGIN | (want_best_states ? RP_PLAN_ROWS : 0u), (rp_result *)0, (double *)0);
}
static inline int rp_plan_packed(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int32_t nT, int32_t nL, int32_t nD,
                                 rp_result *result, double *out /* [(RP_N_ARRAYS + 13) * (N + 1)] */) {
    rp_grids g;
    g.struct_size = (uint32_t)sizeof(rp_grids);
    g.nT = nT; g.nL = nL; g.nD = nD;
    g.T = g.L = g.D = (const double *)0; g.traj_len = (const int32_t *)0;
    return rp_plan(ctx, params, cost, &g, 0, -1, RP_PLAN_PACKED, result, out);
}
static inline int rp_plan_levels_begin(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int32_t n_levels, const rp_grids *grids,
                                       int32_t want_best_states) {
    return rp_plan_levels(ctx, params, cost, n_levels, grids, RP_PLAN_BEGIN | (want_best_states ? RP_PLAN_ROWS : 0u), (rp_result *)0, (double *)0,
                          (int32_t *)0);
}
static inline int rp_plan_levels_packed(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int32_t n_levels,
                                        const int32_t *dims /* [n_levels][3] */, rp_result *result, double *out, int32_t *level) {
    rp_grids g[64];
    int32_t k;
    if (n_levels < 1 || n_levels > 64 || !dims) return RP_EINVAL;
    for (k = 0; k < n_levels; ++k) {
        g[k].struct_size = (uint32_t)sizeof(rp_grids);
        g[k].nT = dims[3 * k]; g[k].nL = dims[3 * k + 1]; g[k].nD = dims[3 * k + 2];
        g[k].T = g[k].L = g[k].D = (const double *)0; g[k].traj_len = (const int32_t *)0;
    }
    return rp_plan_levels(ctx, params, cost, n_levels, g, RP_PLAN_PACKED, result, out, level);
}
static inline int rp_plan_coeffs_grouped(rp_ctx *ctx, const rp_params *params, const rp_cost *cost, int64_t C, const double *lon_coeffs,
                                         const double *lat_coeffs, const double *lon_T, const int32_t *traj_len, int64_t n_groups,
                                         const int32_t *group, const int32_t *group_first, rp_result *result, double *best_states) {
    return rp_plan_coeffs(ctx, params, cost, C, lon_coeffs, lat_coeffs, lon_T, traj_len, n_groups, group, group_first, result, best_states);
}

#ifdef __cplusplus
}
#endif
#endif /* RP_AMD_H */
