/* A C host's replanning loop over the public C ABI (include/rp_amd.h): rp_plan called `steps` times on a cycle of prepared
 * inputs, the way a compiled caller of the library would (the reference's loop is run_planner.py:70-126; its hot call is
 * plan(), reactive_planner.py:566).  bench.py times its regions through this loop, so that the throughput it reports is the
 * library's and not the ctypes marshalling of the Python binding (three byref objects, a result object and a NumPy buffer
 * per call: ~6 us, a fifth of a cfg2 step).  Plain C, no HIP: the entry points come in as function pointers, the library
 * itself is not linked. */
#include <stdint.h>
#include <stddef.h>
#include "rp_amd.h"

typedef int (*rp_plan_fn)(rp_ctx *, const rp_params *, const rp_cost *, const rp_grids *, int64_t, int64_t, rp_result *, double *);
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
        const int rc = plan(ctx, params[i], cost[i], grids[i], cand_begin ? cand_begin[i] : 0, cand_end ? cand_end[i] : -1, result,
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
