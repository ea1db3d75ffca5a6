/* A C host's replanning loop over the public C ABI (include/rp_amd.h): rp_plan called `steps` times on a cycle of prepared
 * inputs, the way a compiled caller of the library would (the reference's loop is run_planner.py:70-126; its hot call is
 * plan(), reactive_planner.py:566).  bench.py times its regions through this loop, so that the throughput it reports is the
 * library's and not the ctypes marshalling of the Python binding (three byref objects, a result object and a NumPy buffer
 * per call: ~6 us, a fifth of a cfg2 step).  Plain C, no HIP: the entry points come in as function pointers, the library
 * itself is not linked. */
#include <stdint.h>
#include <stddef.h>
#include "rp_amd.h"

typedef int (*rp_plan_fn)(rp_ctx *, const rp_params *, const rp_cost *, const rp_grids *, int64_t, int64_t, uint32_t, rp_result *, double *);
typedef int (*rp_last_path_fn)(const rp_ctx *);

typedef struct rp_hostloop_stats {
    int64_t feasible_sum;    /* sum of rp_result.n_feasible over the steps */
    int64_t winners;         /* steps with a winner */
    int64_t paths[4];        /* rp_last_path() histogram */
    double kernel_ms_sum;    /* rp_result.kernel_ms of the profiled steps (rp_set_profiling) */
    int64_t kernel_ms_n;
} rp_hostloop_stats;

/* steps k0 .. k0 + steps - 1 of the cycle: input k % n_inputs, candidates [cand_begin[i], cand_end[i]) (NULL: all).
 * Returns the first non-zero rp_plan status (the loop stops there), 0 otherwise. */
int rp_hostloop_run(rp_plan_fn plan, rp_last_path_fn last_path, rp_ctx *ctx, int32_t n_inputs, const rp_params *const *params,
                    const rp_cost *const *cost, const rp_grids *const *grids, const int64_t *cand_begin, const int64_t *cand_end,
                    int64_t k0, int64_t steps, rp_result *result, double *best_states, rp_hostloop_stats *st) {
    for (int64_t k = k0; k < k0 + steps; ++k) {
        const int32_t i = (int32_t)(k % n_inputs);
        const int rc = plan(ctx, params[i], cost[i], grids[i], cand_begin ? cand_begin[i] : 0, cand_end ? cand_end[i] : -1, 0u, result,
                            best_states);
        if (rc != 0) return rc;
        if (st) {
            st->feasible_sum += result->n_feasible;
            st->winners += result->best_index >= 0;
            if (result->kernel_ms > 0.0) { st->kernel_ms_sum += result->kernel_ms; st->kernel_ms_n += 1; }
            if (last_path) { const int p = last_path(ctx); if (p >= 0 && p < 4) st->paths[p] += 1; }
        }
    }
    return 0;
}

/* ---- the same loop for a rank of a sharded group (one process per GPU, ranks of one node) -------------------------------------
 * Step k: rp_plan on this rank's candidate range, then the winner exchange through the shared-memory mailbox (rp_mailbox_exchange;
 * rp_mailbox_sum for the count of colliding candidates before the GLOBAL winner, only when some rank saw a collision -- the rules of
 * commonroad_rp_amd/distributed.py: local_collisions_before).  What bench.py --gpus N times.  `seq` is the mailbox's exchange
 * counter (in: last used, out: last used), the same on every rank. */
#include <time.h>

typedef int (*rp_exchange_fn)(void *, int32_t, int32_t, uint64_t, int32_t, const rp_result *, const double *, rp_result *, double *,
                              int32_t *);
typedef int (*rp_sum_fn)(void *, int32_t, int32_t, uint64_t, int32_t, int64_t, int64_t *);
typedef int (*rp_count_before_fn)(rp_ctx *, double, int64_t, int64_t *);

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* Returns 0, or the first failing status: > 0 ... never; negative RP_* codes of rp_plan / rp_count_collisions_before, or
 * -1000 - code when it was the mailbox (time-out: the region is unusable afterwards, see rp_amd.h). */
int rp_hostloop_run_sharded(rp_plan_fn plan, rp_last_path_fn last_path, rp_exchange_fn exchange, rp_sum_fn sum,
                            rp_count_before_fn count_before, rp_ctx *ctx, void *region, int32_t world, int32_t rank, uint64_t *seq,
                            int32_t n_inputs, const rp_params *const *params, const rp_cost *const *cost,
                            const rp_grids *const *grids, const int64_t *cand_begin, const int64_t *cand_end, int64_t k0,
                            int64_t steps, rp_result *local, double *local_states, rp_result *global, double *global_states,
                            rp_hostloop_stats *st, double *exchange_seconds) {
    double t_ex = 0.0;
    for (int64_t k = k0; k < k0 + steps; ++k) {
        const int32_t i = (int32_t)(k % n_inputs);
        int rc = plan(ctx, params[i], cost[i], grids[i], cand_begin[i], cand_end[i], 0u, local, local_states);
        if (rc != 0) return rc;
        if (st) {
            st->feasible_sum += local->n_feasible;
            if (local->kernel_ms > 0.0) { st->kernel_ms_sum += local->kernel_ms; st->kernel_ms_n += 1; }
            if (last_path) { const int p = last_path(ctx); if (p >= 0 && p < 4) st->paths[p] += 1; }
        }
        const double t0 = now_s();
        const int32_t n = params[i]->N + 1;
        int32_t owner = -1;
        *seq += 1;
        rc = exchange(region, world, rank, *seq, n, local, local->best_index >= 0 ? local_states : NULL, global, global_states, &owner);
        if (rc != 0) return -1000 + rc;
        if (global->n_collision > 0) {   /* second message only when some rank saw a colliding candidate */
            int64_t before = 0, total = 0;
            if (local->n_collision == 0) before = 0;
            else if (global->best_index < 0) before = local->n_collision;
            else if (owner == rank) before = local->n_collision_before_best;
            else if (local->best_index >= 0 && local->n_collision_before_best == 0) before = 0;
            else {
                rc = count_before(ctx, global->best_cost, global->best_index, &before);
                if (rc != 0) return rc;
            }
            rc = sum(region, world, rank, *seq, n, before, &total);
            if (rc != 0) return -1000 + rc;
            global->n_collision_before_best = total;
        }
        if (st) st->winners += global->best_index >= 0;
        t_ex += now_s() - t0;
    }
    if (exchange_seconds) *exchange_seconds = t_ex;
    return 0;
}
