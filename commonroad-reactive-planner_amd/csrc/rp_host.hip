// rp_host.hip -- C ABI (include/rp_amd.h) over the HIP kernels of rp_kernels.h.
//
// Host side of the boundary: owns the device tables and work buffers of one rp_ctx, stages the
// (tiny) per-call inputs through pinned memory, launches
//     small batches:  rp_eval_kernel<.., LON_FUSED> -> rp_finalize_kernel
//     large batches:  rp_lon_kernel -> rp_eval_kernel -> [rp_fold_partials_kernel] -> rp_finalize_kernel
//                     -> [rp_count_before_kernel] -> [rp_eval_kernel<single winner>]
// on the context's stream; the result block arrives in pinned host memory straight from the kernels.  No torch types, no exceptions
// across the ABI, no CPU fallback: if HIP fails the call returns RP_EHIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "rp_kernels.h"
#include "rp_frontend.h"
#include "rp_corridor.h"
#include "rp_pool.h"

namespace {

constexpr int kBlocksPerCU = 4;           // workgroups per CU of the grid-stride helper kernels (longitudinal profiles, counts)
constexpr int kFoldPartials = 256;        // large batches: workgroup partials are folded to this many before the epilogue
constexpr int kFoldThreshold = 8192;      // (RP_AMD_FOLD_THRESHOLD overrides) every workgroup of the selection epilogue reads all partials: up to
                                          // here straight from the evaluation kernel's (cfg3: 3 906), beyond (cfg4 32 k, cfg5 64 k) folded first
constexpr size_t kAutoMaterializeBytes = 64u << 20;   // fused mode: up to this many bytes of state rows replace the winner pass
                                          // (measured: pays off from the first obstacle on, cfg4 with 5: 1.22 -> 0.89 ms)
constexpr size_t kLdsTableLimit = 65536;  // stage the reference tables in LDS up to this many bytes
// costs-only launches without the collision query (profiles/probe_chunk_threshold.py): rp_chunk_kernel from this many wavefronts per CU
// on, rp_cost_kernel from this many candidates per CU on
constexpr int kChunkMinWavesPerCU = 7;
constexpr int kCostMinPerCU = 832;
constexpr int kSweepMinPerCU = 512;   // candidates per CU from which a bounded sweep, not the eager kernel, takes over from exhausted list rounds
// cost-ordered collision stage: capacity of the candidate lists of the three rounds and their places in d_lazy_lists
constexpr int kLazyCap[RP_LAZY_LEVELS] = {1024, 4096, 16384};
constexpr int kLazyOff[RP_LAZY_LEVELS] = {0, 1024, 5120};
constexpr int kLazyListWords = 21504;
constexpr uint32_t kLazyTarget[RP_LAZY_LEVELS] = {128, 1024, 8192};   // candidates (cumulative) the levels are meant to reach

using ResultBlock = FinalizeOut;   // device -> host result block (rp_kernels.h)

// Per-context options (rp_set_option / rp_get_option; include/rp_amd.h lists them).  The environment variable of an option is read
// ONCE, in rp_create, as the context's default: no entry point that plans reads the environment.
struct Options {
    int lanes = 0;               // lanes per candidate of rp_eval_kernel: 0 = by batch (lanes_per_candidate), 16 | 32 | 64
    int eval_block = 0;          // threads per workgroup of the batch's rp_eval_kernel launch: 0 = by batch (eval_block), 64 | 256
    int cost_kernel = -1;        // rp_cost_kernel (one lane per candidate): -1 = by batch, 0 never, 1 whenever it applies
    int chunk_kernel = -1;       // rp_chunk_kernel (one lane per candidate and step block): -1 = by batch, 0 never, 1 whenever it applies
    int lazy = -1;               // cost-ordered collision stage: -1 = by the context's collision path, 0 never, 1 whenever the launch path allows it
    int sweep = -1;              // ... as a bounded sweep over the batch instead of list rounds: -1 = by batch, 0 never, 1 whenever a cost-ordered stage runs
    int fused_lon = 1;           // single-launch variant for small batches
    int fused_lon_blocks = -1;   // ... up to this many workgroups (-1: 4 per CU)
    int auto_materialize = 1;    // small batches whose winner rows are wanted write every candidate's rows
    int stage_out = 1, row_padding = 1, row_align = 0, tail_split = 1;   // layout of the state rows in device memory
    int table_window = 1;        // single-launch variant stages only the part of the reference tables a plan can touch
    int fold_threshold = kFoldThreshold;
    int lon_publish = 1, inline_grids = 1, event_bracket = 0, winner_lanes_as_batch = 0, zero_copy = 1, coeff_groups = 1;
    int winner_skip_query = 1;   // the re-evaluation of a production-mode plan's winner (its state rows) runs without the collision query
    int lazy_trace = 0, print_stamps = 0, timing = 0;   // diagnostics on stderr
    int wait_mode = RP_WAIT_SPIN;   // how the host waits for a plan's completion ticket (rp_set_wait_mode)
};
struct OptionDesc {
    const char *key;
    int Options::*field;
    const char *env;     // environment variable read by rp_create (nullptr: none)
    bool env_negates;    // the variable's presence means 0 (RP_AMD_NO_*); else its integer value
    int lo, hi;          // accepted range
};
const OptionDesc kOptionTable[] = {
    {"lanes", &Options::lanes, "RP_AMD_G", false, 0, 64},
    {"eval_block", &Options::eval_block, "RP_AMD_EVAL_BLOCK", false, 0, 256},
    {"cost_kernel", &Options::cost_kernel, "RP_AMD_COST_KERNEL", false, -1, 1},
    {"chunk_kernel", &Options::chunk_kernel, "RP_AMD_CHUNK_KERNEL", false, -1, 1},
    {"lazy", &Options::lazy, "RP_AMD_LAZY", false, -1, 1},
    {"sweep", &Options::sweep, "RP_AMD_SWEEP", false, -1, 1},
    {"fused_lon", &Options::fused_lon, "RP_AMD_NO_FUSED_LON", true, 0, 1},
    {"fused_lon_blocks", &Options::fused_lon_blocks, "RP_AMD_FUSED_LON_BLOCKS", false, -1, 1 << 20},
    {"auto_materialize", &Options::auto_materialize, "RP_AMD_NO_AUTO_MATERIALIZE", true, 0, 1},
    {"stage_out", &Options::stage_out, "RP_AMD_NO_STAGE_OUT", true, 0, 1},
    {"row_padding", &Options::row_padding, "RP_AMD_NO_ROW_PADDING", true, 0, 1},
    {"row_align", &Options::row_align, "RP_AMD_ROW_ALIGN", false, 0, 16},
    {"tail_split", &Options::tail_split, "RP_AMD_NO_TAIL_SPLIT", true, 0, 1},
    {"table_window", &Options::table_window, "RP_AMD_NO_TABLE_WINDOW", true, 0, 1},
    {"fold_threshold", &Options::fold_threshold, "RP_AMD_FOLD_THRESHOLD", false, 1, 1 << 30},
    {"lon_publish", &Options::lon_publish, "RP_AMD_NO_LON_PUBLISH", true, 0, 1},
    {"inline_grids", &Options::inline_grids, "RP_AMD_NO_INLINE_GRIDS", true, 0, 1},
    {"event_bracket", &Options::event_bracket, "RP_AMD_EVENT_BRACKET", false, 0, 1},
    {"winner_lanes_as_batch", &Options::winner_lanes_as_batch, "RP_AMD_WINNER_G_AS_BATCH", false, 0, 1},
    {"winner_skip_query", &Options::winner_skip_query, "RP_AMD_NO_WINNER_SKIP_QUERY", true, 0, 1},
    {"zero_copy", &Options::zero_copy, "RP_AMD_NO_ZERO_COPY", true, 0, 1},
    {"coeff_groups", &Options::coeff_groups, "RP_AMD_NO_COEFF_GROUPS", true, 0, 1},
    {"lazy_trace", &Options::lazy_trace, "RP_AMD_LAZY_TRACE", false, 0, 1},
    {"print_stamps", &Options::print_stamps, "RP_AMD_PRINT_STAMPS", false, 0, 1},
    {"timing", &Options::timing, "RP_AMD_TIMING", false, 0, 1},
    {"wait_mode", &Options::wait_mode, "RP_AMD_WAIT_MODE", false, RP_WAIT_SPIN, RP_WAIT_EVENT},
};
const OptionDesc *find_option(const char *key) {
    if (!key) return nullptr;
    for (const OptionDesc &d : kOptionTable)
        if (std::strcmp(d.key, key) == 0) return &d;
    return nullptr;
}
// the environment's defaults (rp_create only)
void options_from_environment(Options &o) {
    for (const OptionDesc &d : kOptionTable) {
        const char *e = d.env ? std::getenv(d.env) : nullptr;
        if (!e) continue;
        const int v = d.env_negates ? 0 : std::atoi(e);
        if (v >= d.lo && v <= d.hi) o.*(d.field) = v;
    }
}

}  // namespace

struct rp_ctx {
    Options opt;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cus = 256;
    int profiling = 0;            // 0 off, k > 0: time the evaluation kernel of every k-th rp_plan with HIP events
    unsigned long long calls = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool time_next_launch = false, timed_by_launch = false;   // profiling: the next launch_block carries ev0 / ev1 | it did

    // reference tables
    double *d_tables = nullptr;
    int n_ref = 0, search_iters = 0, n_buckets = 0, table_words = 0;
    double bucket_inv_h = 0.0;
    double proj_d_limit = 20.0;
    std::vector<double> h_pos;    // host copy of ref_pos (table window of the single-launch variant)
    rpfe::Tables h_front;         // host copy of the reference path and its tables (rp_plan_cycle: initial state of a Cartesian state)
    std::vector<double> cycle_buf;   // rp_plan_cycle: the grids of the cycle's levels as handed to the level loop
    // obstacles
    double *d_sobb = nullptr, *d_tri = nullptr, *d_circ = nullptr, *d_dyn = nullptr, *d_clus = nullptr, *d_slot = nullptr;
    // uniform grid over the static shapes, built for the ego radius of the plans (ensure_static_grid)
    unsigned long long *d_grid = nullptr;
    std::vector<double> h_sobb, h_tri, h_circ;   // host copies of the static rows (device layout) and the cluster descriptors
    std::vector<int32_t> h_clus_info;
    bool grid_valid = false;
    double grid_ego_r = 0.0;
    int32_t *d_clus_info = nullptr;
    ObsTables obs{};

    // per-call staging (pinned host + device mirror)
    char *h_stage = nullptr, *d_stage = nullptr;
    size_t cap_stage = 0;
    std::vector<char> staged;   // grids of the last rp_plan (host copy); staged_on_device: also what d_stage holds (skip the copy when unchanged)
    bool staged_on_device = false;
    // work buffers
    uint32_t *d_status = nullptr;
    double *d_cost = nullptr, *d_user = nullptr;
    size_t cap_status = 0, cap_cost = 0, cap_user = 0;
    double *d_states = nullptr;
    double *d_compact = nullptr;   // rp_fetch_states: compact copy of padded state rows
    size_t cap_compact = 0;
    size_t cap_states = 0;
    double *d_profile = nullptr, *d_profile_one = nullptr;   // longitudinal profiles (batch / rp_eval_one scratch)
    size_t cap_profile = 0, cap_profile_one = 0;
    PairHdr *d_pair_hdr = nullptr, *d_pair_hdr_one = nullptr;
    size_t cap_pair_hdr = 0;
    void *d_partials = nullptr;   // Partials (rp_kernels.h): arrays of cap_partials + kFoldPartials slots -- one per workgroup, then the folded ones
    int cap_partials = 0;
    unsigned long long *d_sel_scratch = nullptr;   // rp_select_kernel: counter totals and arrival ticket (zero between launches)
    // cost-ordered collision stage (rp_kernels.h: LazyCtl): control block, the three candidate lists, state rows of a round's candidates
    LazyCtl *d_lazy_ctl = nullptr;
    int32_t *d_lazy_lists = nullptr;
    uint32_t *d_lazy_hist = nullptr;       // [RP_LAZY_BINS + 1] histogram of the costs + arrival ticket (zero between launches)
    double *d_lazy_states = nullptr;
    size_t cap_lazy_states = 0;
    int lazy_skip = 0, lazy_penalty = 0;   // plans that go eager straight away after a lazy attempt had to fall back (doubles per failure, up to 64)
    int collision_mode = RP_COLLISION_AUTO;   // rp_set_collision_path
    int last_kernel = 0;                   // rp_last_kernel: which kernel evaluated the last plan's batch (RP_KERNEL_*)
    int last_lazy = 0;                     // 0: the last plan ran eager, 1: lazy, 2: lazy attempt + eager fallback (diagnostic, rp_last_path)
    double last_best_cost = 0.0;           // winner of the last collected plan (rp_count_collisions_before after a cost-ordered plan)
    int64_t last_best_index = -1;
    // Which of the two ways to answer the collision query is the faster one depends on the scene (how deep into the cost order
    // the first free candidate lies) and on the batch (what a collision round costs against the eager kernel's extra work): the
    // context times its own plans -- whole calls, rp_plan / rp_plan_coeffs -- per path and uses the faster one,
    // measuring the other again every 64th plan.  Results are the same either way (rp_amd.h: rp_last_path).
    double path_us[2] = {0.0, 0.0};        // running mean of the call time, [0] eager [1] cost-ordered (where the stage found the winner)
    int path_n[2] = {0, 0};
    int path_since_probe = 0, path_regime = -1;   // regime: explicit polynomials or grids, log2 of the candidate count
    bool path_adaptive = false;            // the plan in flight chose its path by this rule (not forced, not a small batch)
    bool time_whole = false;               // the plan in flight is a whole call: its duration counts
    std::chrono::steady_clock::time_point plan_t0;
    bool epilogue_dirty = false;           // a plan's chain was cut short (an error between its launches and its result): the scratch words
                                           // its epilogues keep at zero between launches (arrival tickets, totals, histogram) are cleared first
    // a plan whose kernels are on the stream and whose result has not been collected (rp_plan with RP_PLAN_BEGIN .. rp_plan_wait)
    struct Pending {
        bool active = false, cin = false, skip_eval = false, ticket = false, done = false, time_valid = false, mat = false, coeffs = false;
        unsigned long long seq = 0;
        int grid = 0;
        KArgs ka{};
        std::chrono::steady_clock::time_point tp0, tp1;
    } pending;
    // a chain of sampling levels in flight (rp_plan_levels with RP_PLAN_BEGIN .. rp_plan_wait): what every level's launch left behind -- the
    // result names the level it belongs to, whose state then becomes the context's "last plan"
    struct LevelState {
        Pending pending;
        std::vector<char> staged;
        bool staged_on_device = false, last_rows_on_device = false;
        int last_G = 0, last_block = 0, last_kernel = 0, last_lazy = 0;
        size_t last_fused_lds = 0;
        int grid_index = 0;   // index of the level in the caller's array
    };
    std::vector<LevelState> chain;          // levels of the chain in flight (empty: a plan on its own)
    int chain_total = 0;                    // levels the caller handed over (levels behind the chain run one by one in rp_plan_wait)
    unsigned long long *d_gate = nullptr;   // device word of the chain (KArgs::gate)
    unsigned long long *d_sweep = nullptr;  // bounded sweep: cost key of the cheapest free candidate found so far (KArgs::sweep_bound)
    struct ChainInputs {                    // the caller's arguments, kept for the levels behind the chain
        rp_params p; rp_cost cost; std::vector<rp_grids> grids; std::vector<std::vector<double>> T, L, D; std::vector<std::vector<int32_t>> tl;
        int want = 0;
    } chain_in;
    int last_level = 0;                     // rp_last_level: index (into the caller's grids) of the level the last result belongs to
    char *d_result = nullptr, *h_result = nullptr, *h_result_dev = nullptr;   // h_result_dev: device address of the pinned block
    size_t cap_result = 0;
    int64_t *d_single = nullptr, *h_single = nullptr;
    unsigned long long *d_debug = nullptr;   // diagnostic build only
    unsigned long long seq = 0;              // completion tickets handed to the kernels
    bool spin_wait = true;                   // the kernels hand a completion ticket over in the pinned result block (off: RP_WAIT_EVENT)
    hipEvent_t ev_done = nullptr;            // RP_WAIT_EVENT: recorded behind a plan's last launch; the host thread sleeps on it
    long long wait_fallbacks = 0;            // waits for a ticket that ended in the 200-ms fall-back (a kernel chain that did not hand it over)

    // last plan
    bool have_last = false, last_mat = false, last_coeffs = false;
    std::unordered_map<const void *, hipFunction_t> functions;   // kernel symbol -> function handle (launch_kargs)
    std::vector<double> fast_buf;   // rp_fast_buffer: grids of RP_PLAN_PACKED calls, written by the caller ([T | L | D | traj_len int32])
    // option "timing": host-side phase times of rp_plan (sums over calls, printed by rp_destroy)
    bool timing = false;
    double t_sum[6] = {0, 0, 0, 0, 0, 0};   // entry -> first launch | launches | wait for the ticket | unpack | evaluation launch | epilogue launch
    unsigned long long t_calls = 0;
    bool t_warm = false;
    std::chrono::steady_clock::time_point t_entry;
    bool last_rows_on_device = false;   // d_result holds the winner's state rows of the last plan (finalize copied them)
    int last_G = 0;              // lanes per candidate of the last plan
    int last_block = 0;          // threads per workgroup of its evaluation kernel
    size_t last_fused_lds = 0;   // LDS bytes of the single-launch variant if the last plan used it (else 0)
    KArgs last{};
    KArgsG kargs_g{};            // launch block of the evaluation kernels: KArgs + the grids of the last rp_plan (if they fit)
    KArgsGL kargs_gl{};          // launch block of rp_lon_kernel: KArgs + room for larger grids (it publishes them to d_stage for the kernels behind it)
    bool grids_pending = false;  // the grids of the last rp_plan (c->staged, h_stage) are neither in kargs_g nor in d_stage yet: run_pipeline
                                 // hands them to rp_lon_kernel's kernarg segment (two-kernel path) or copies them
    char *h_arena = nullptr, *h_arena_dev = nullptr;   // rp_coeffs_arena: pinned [lon 6 cap | lat 6 cap | traj_len cap] and its device address
    int64_t arena_cap = 0;
    const double *cin_lon = nullptr, *cin_lat = nullptr;   // host arrays the last rp_plan_coeffs read (pinned stage or arena)
    int64_t cin_groups = 0;                 // rp_plan_coeffs with groups: groups of the last such plan (0: one pair per candidate)
    const int32_t *cin_group_of = nullptr;  //   and the host view of its candidate -> group table (arena)
    int64_t cin_count = -1;   // rp_plan_coeffs: the explicit polynomials of the last such plan are the first 12 * cin_count doubles of h_stage (lon | lat)
};

namespace {

int fail(rp_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, RP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

template <typename T>
int grow(rp_ctx *c, T *&ptr, size_t &cap, size_t need) {
    if (need <= cap) return RP_OK;
    if (ptr) HIP_TRY(c, hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    size_t want = std::max(need, (size_t)1024);
    HIP_TRY(c, hipMalloc((void **)&ptr, want * sizeof(T)));
    cap = want;
    return RP_OK;
}

int upload(rp_ctx *c, double *&dptr, const std::vector<double> &host) {
    if (dptr) { HIP_TRY(c, hipFree(dptr)); dptr = nullptr; }
    if (host.empty()) return RP_OK;
    HIP_TRY(c, hipMalloc((void **)&dptr, host.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(dptr, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    return RP_OK;
}

int ensure_result(rp_ctx *c, int n) {
    size_t need = sizeof(ResultBlock) + sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n;
    if (need <= c->cap_result) return RP_OK;
    if (c->d_result) HIP_TRY(c, hipFree(c->d_result));
    if (c->h_result) HIP_TRY(c, hipHostFree(c->h_result));
    c->d_result = c->h_result = nullptr;
    c->cap_result = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_result, need));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_result, need, hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer((void **)&c->h_result_dev, c->h_result, 0));
    c->cap_result = need;
    return RP_OK;
}

int ensure_stage(rp_ctx *c, size_t need) {
    if (need <= c->cap_stage) return RP_OK;
    if (c->d_stage) HIP_TRY(c, hipFree(c->d_stage));
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->d_stage = c->h_stage = nullptr;
    c->cap_stage = 0;
    size_t want = std::max(need, (size_t)65536);
    HIP_TRY(c, hipMalloc((void **)&c->d_stage, want));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_stage, want, hipHostMallocDefault));
    c->cap_stage = want;
    c->staged.clear();
    c->staged_on_device = false;
    return RP_OK;
}

constexpr size_t kStageOutLimit = 65536;   // LDS bytes per workgroup up to which state rows are staged for linear copy-out

// Launch of a kernel whose only parameter is the KArgs block, through hipModuleLaunchKernel with the argument
// buffer handed over as it is (no per-launch symbol look-up and argument marshalling of hipLaunchKernelGGL).
void launch_block(rp_ctx *c, const void *kernel, int grid, int block, size_t lds, const void *args, size_t bytes);
void launch_kargs(rp_ctx *c, const void *kernel, int grid, int block, size_t lds, const KArgs &ka) {
    c->kargs_g.k = ka;   // (the grid part was filled by rp_plan and stays until the next one: winner re-evaluation, rp_eval_one)
    launch_block(c, kernel, grid, block, lds, &c->kargs_g, sizeof(KArgsG));
}
// (any kernel whose only parameter is one struct passed by value)
void launch_block(rp_ctx *c, const void *kernel, int grid, int block, size_t lds, const void *args_block, size_t bytes) {
    hipFunction_t f = nullptr;
    auto it = c->functions.find(kernel);
    if (it == c->functions.end()) {
        if (hipGetFuncBySymbol(&f, kernel) != hipSuccess || !f) f = nullptr;
        c->functions.emplace(kernel, f);
    } else {
        f = it->second;
    }
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args_block), HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
    if (c->time_next_launch) {   // profiling: this launch carries the two events itself -- they take the kernel's own begin / end
        c->time_next_launch = false;   // time stamps instead of bracketing it with two more commands on the stream
        if (f && hipExtModuleLaunchKernel(f, (uint32_t)grid * (uint32_t)block, 1, 1, (uint32_t)block, 1, 1, lds, c->stream, nullptr, extra,
                                          c->ev0, c->ev1, 0) == hipSuccess) {
            c->timed_by_launch = true;
            return;
        }
        (void)hipGetLastError();
    }
    if (!f || hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, c->stream, nullptr, extra) != hipSuccess) {
        (void)hipGetLastError();
        void *args[] = {const_cast<void *>(args_block)};
        (void)hipLaunchKernel(kernel, dim3(grid), dim3(block), args, lds, c->stream);   // error, if any, surfaces through hipGetLastError
    }
}

// Kernel variant by obstacle content: 0 no collision test, 1 dynamic obstacles only, 2 static shapes as well (the walks over
// static shapes are a large part of the query code; workloads without any do not carry them)
int collision_level(const KArgs &ka) {
    if (!ka.has_obstacles || (ka.flags & RP_FLAG_SKIP_COLLISION)) return 0;
    return (ka.obs.n_sobb + ka.obs.n_tri + ka.obs.n_circ) > 0 ? 2 : 1;
}

// the bounded sweep's variant of rp_eval_kernel by run-time (explicit polynomials, collision level, one step block)
const void *sweep_kernel_fn(bool cin, int coll, bool one_chunk) {
    if (cin) {
        if (coll == 2) return one_chunk ? (const void *)rp_eval_kernel<16, false, true, 2, true, false, false, RP_BLOCK, true> : (const void *)rp_eval_kernel<16, false, true, 2, false, false, false, RP_BLOCK, true>;
        return one_chunk ? (const void *)rp_eval_kernel<16, false, true, 1, true, false, false, RP_BLOCK, true> : (const void *)rp_eval_kernel<16, false, true, 1, false, false, false, RP_BLOCK, true>;
    }
    if (coll == 2) return one_chunk ? (const void *)rp_eval_kernel<16, false, false, 2, true, false, false, RP_BLOCK, true> : (const void *)rp_eval_kernel<16, false, false, 2, false, false, false, RP_BLOCK, true>;
    return one_chunk ? (const void *)rp_eval_kernel<16, false, false, 1, true, false, false, RP_BLOCK, true> : (const void *)rp_eval_kernel<16, false, false, 1, false, false, false, RP_BLOCK, true>;
}

// rp_chunk_kernel<LOW, COLL> by run-time (low-velocity mode, collision level)
const void *chunk_kernel_fn(bool low, int coll) {
    if (low) return coll == 2 ? (const void *)rp_chunk_kernel<true, 2> : (coll == 1 ? (const void *)rp_chunk_kernel<true, 1> : (const void *)rp_chunk_kernel<true, 0>);
    return coll == 2 ? (const void *)rp_chunk_kernel<false, 2> : (coll == 1 ? (const void *)rp_chunk_kernel<false, 1> : (const void *)rp_chunk_kernel<false, 0>);
}

template <int G, bool MAT, bool CIN, int COLL, bool STAGE>
void launch_eval_tcs(rp_ctx *c, const KArgs &ka, int grid, size_t lds, int block = RP_BLOCK) {
    if constexpr (G == 16 && !STAGE) {   // one wavefront per workgroup (large batches of the two-kernel path: eval_block)
        if (block == 64) {
            if (ka.N + 1 <= G) launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, true, false, false, 64>, grid, 64, lds, ka);
            else launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, false, false, false, 64>, grid, 64, lds, ka);
            return;
        }
    }
    if (ka.N + 1 <= G) launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, true, STAGE, false>, grid, RP_BLOCK, lds, ka);
    else launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, false, STAGE, false>, grid, RP_BLOCK, lds, ka);
}

// ---- single-launch variant (longitudinal profiles computed by each workgroup into LDS) ---------------------
constexpr int kFusedLonG = 16;                 // lanes per candidate of the single-launch variant
constexpr size_t kFusedLonLdsLimit = 61440;    // LDS bytes per workgroup it may use (tables + profile rows)
int fused_lon_max_blocks(const rp_ctx *c) {    // batches up to this many workgroups (16 candidates each) take it
    if (c->opt.fused_lon_blocks >= 0) return c->opt.fused_lon_blocks;
    return c->num_cus * 4;
}

// LDS bytes of the single-launch variant for this batch (0: not eligible); *pairs = profile capacity per workgroup
size_t fused_lon_lds(const rp_ctx *c, const KArgs &ka, int64_t count, int G, bool cin, bool mat, int *pairs) {
    // 16 lanes per candidate, or one wavefront per candidate for horizons of up to 64 steps (small batches at the reference's
    // default N = 60: one launch instead of rp_lon_kernel + rp_eval_kernel)
    if (count <= 0 || !(G == kFusedLonG || (G == 64 && mat && ka.N + 1 <= 64) || (G == 32 && mat && ka.N + 1 <= 32))) return 0;
    const int gpb = RP_BLOCK / G;
    if ((count + gpb - 1) / gpb > (int64_t)fused_lon_max_blocks(c)) return 0;
    // consecutive candidates of one workgroup touch at most this many (T, longitudinal sample) pairs
    const int P = cin ? gpb : std::min<int64_t>(gpb, ((int64_t)gpb + ka.nD - 2) / ka.nD + 1);
    // reference tables + profile rows + pair headers + pre-filter votes + time sample per pair
    // (+ the member slots of the static clusters, which the broad phase of the prologue reads from LDS)
    const size_t slot_bytes = ka.has_obstacles ? (size_t)ka.obs.n_clus * ka.obs.clus_per * RP_SLOT_ROW * sizeof(double) : 0;
    const size_t bytes = (size_t)ka.table_words * sizeof(double) + slot_bytes +
                         (size_t)P * ((size_t)PF_STRIDE * (size_t)(ka.N + 1) * sizeof(double) + sizeof(PairHdr) + sizeof(int) + sizeof(double)) + 16;
    if (bytes > kFusedLonLdsLimit) return 0;
    *pairs = P;
    return (bytes + 15) & ~(size_t)15;
}

template <bool MAT, bool CIN, int COLL>
void launch_eval_fused_c(rp_ctx *c, const KArgs &ka, int grid, size_t lds, int lanes) {
    if (lanes == 64) {   // (only chosen with state rows and N + 1 <= 64: one step block)
        launch_kargs(c, (const void *)rp_eval_kernel<64, true, CIN, COLL, true, false, true>, grid, RP_BLOCK, lds, ka);
        return;
    }
    if (lanes == 32) {   // (state rows, N + 1 <= 32: one step block)
        launch_kargs(c, (const void *)rp_eval_kernel<32, true, CIN, COLL, true, false, true>, grid, RP_BLOCK, lds, ka);
        return;
    }
    constexpr int G = kFusedLonG;
    if (ka.N + 1 <= G) launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, true, false, true>, grid, RP_BLOCK, lds, ka);
    else launch_kargs(c, (const void *)rp_eval_kernel<G, MAT, CIN, COLL, false, false, true>, grid, RP_BLOCK, lds, ka);
}

template <bool MAT, bool CIN>
void launch_eval_fused_t(rp_ctx *c, const KArgs &ka, int grid, size_t lds, int lanes) {
    const int coll = collision_level(ka);
    if (coll == 2) launch_eval_fused_c<MAT, CIN, 2>(c, ka, grid, lds, lanes);
    else if (coll == 1) launch_eval_fused_c<MAT, CIN, 1>(c, ka, grid, lds, lanes);
    else launch_eval_fused_c<MAT, CIN, 0>(c, ka, grid, lds, lanes);
}

void launch_eval_fused(rp_ctx *c, const KArgs &ka, int grid, bool mat, bool cin, size_t lds, int lanes) {
    if (mat) { if (cin) launch_eval_fused_t<true, true>(c, ka, grid, lds, lanes); else launch_eval_fused_t<true, false>(c, ka, grid, lds, lanes); }
    else     { if (cin) launch_eval_fused_t<false, true>(c, ka, grid, lds, kFusedLonG); else launch_eval_fused_t<false, false>(c, ka, grid, lds, kFusedLonG); }
}

// State rows leave through LDS as one linear stream (STAGE_OUT) for whole-wavefront candidates of the two-kernel path.
inline bool stage_out_applies(const rp_ctx *c, const KArgs &ka, int G, bool mat) {
    const size_t tile = (size_t)(RP_BLOCK / G) * RP_N_ARRAYS * (size_t)(ka.N + 1) * sizeof(double);
    return mat && !ka.single_index && !ka.index_list && G == 64 && tile <= kStageOutLimit && c->opt.stage_out;
}

// Doubles between two rows of a state block in device memory.  Rows stored directly (every variant but STAGE_OUT) start on
// 128-byte lines -- the line size of the L2: N + 1 rounded up to a multiple of 16, so that the 128-byte run of a group of 16
// lanes never straddles two lines (a multiple of 8 only: 200 000 candidates at N = 100 38 % of the HBM peak instead of 55 %).
// rp_fetch_states hands out compact [14][N + 1] blocks either way.
inline int state_row_stride(const rp_ctx *c, int n, bool staged) {
    if (staged || !c->opt.row_padding) return n;
    if (c->opt.row_align == 8 || c->opt.row_align == 16) return (n + c->opt.row_align - 1) & ~(c->opt.row_align - 1);
    return (n + 15) & ~15;
}
// Split tail (rp_kernels.h: state_offset): when the last step block of 16 holds at most 8 steps (N = 100: 5 of 16), the rows keep
// the full step blocks only and the partial one is stored two rows to a 128-byte line -- whole-line stores as before, without
// 11 doubles of padding per row (cfg4 / cfg5: 12.8 GB written for 11.5 GB of rows).  Two-kernel path, 16 lanes per candidate.
// Returns M (0: padded rows) and sets *ns.
inline int state_layout(const rp_ctx *c, int n, int G, bool fused, bool staged, int *ns) {
    *ns = state_row_stride(c, n, staged);
    const int r = n & 15, M = n - r;
    if (staged || fused || G != 16 || r == 0 || r > 8 || M < 16 || *ns != ((n + 15) & ~15) || !c->opt.tail_split) return 0;
    *ns = M;
    return M;
}

template <int G, bool MAT, bool CIN, int COLL>
void launch_eval_tc(rp_ctx *c, const KArgs &ka, int grid, int block) {
    const size_t tile = (size_t)(RP_BLOCK / G) * RP_N_ARRAYS * (size_t)(ka.N + 1) * sizeof(double);
    const bool stage = stage_out_applies(c, ka, G, MAT);
    if (MAT && stage) launch_eval_tcs<G, MAT, CIN, COLL, true>(c, ka, grid, tile);
    else launch_eval_tcs<G, MAT, CIN, COLL, false>(c, ka, grid, 0, block);
}

template <int G, bool MAT, bool CIN>
void launch_eval_t(rp_ctx *c, const KArgs &ka, int grid, int block) {
    const int coll = collision_level(ka);
    if (coll == 2) launch_eval_tc<G, MAT, CIN, 2>(c, ka, grid, block);
    else if (coll == 1) launch_eval_tc<G, MAT, CIN, 1>(c, ka, grid, block);
    else launch_eval_tc<G, MAT, CIN, 0>(c, ka, grid, block);
}

// Lanes per candidate.  The time axis is cut into step blocks of G lanes; fewer lanes per candidate
// mean more candidates per wavefront (fewer wavefronts for a batch that would otherwise need more
// than one residency round) at the price of a longer dependent chain per wavefront.
int lanes_per_candidate(const rp_ctx *c, int N, int64_t count, bool mat) {
    if (c->opt.lanes == 16 || c->opt.lanes == 32 || c->opt.lanes == 64) return c->opt.lanes;
    // horizons of 33 .. 64 steps (the reference's default N = 60), mid-sized batches: two step blocks of 32 lanes -- half the chain of
    // four blocks of 16, twice the wavefronts -- win between the one-wavefront-per-candidate range and the batches that fill the chip
    // anyway (profiles/probe_small_n60.py, end of round 3, 32 vs 16 lanes, production / draw: 6 000 candidates 51.6 vs 57.1 / 52.0 vs
    // 57.1 us per step, 12 500: 80.0 vs 94.9 / 74.7 vs 81.5; 2 601: as 64 lanes; 22 500: 93.3 vs 83.0 / 87.1 vs 78.0)
    if (N + 1 > 32 && N + 1 <= 64 && count > (int64_t)c->num_cus * 10 && count <= (int64_t)c->num_cus * 64) return 32;
    // measured on MI355X (profiles/r01_lanes_per_candidate.txt): without state rows 16 lanes per candidate
    // win at every horizon; with state rows long horizons want whole-wavefront rows (512-byte runs)
    if (!mat) return 16;
    // up to 32 steps (the shipped configurations: N = 20): 16 lanes, two step blocks -- except for the smallest batches, where
    // every wavefront has a SIMD to itself either way and one step block of 32 lanes is the shorter chain
    // (profiles/probe_small_n20.py, end of round 3, 32 vs 16 lanes: 630 candidates at N = 20 31.3 vs 36.0 us per step, 3 060: 34.3 vs 37.3;
    //  N = 30: 1 041 candidates 27.3 vs 30.0, 2 083: equal, 4 092: 31.1 vs 30.7, 7 440: 38.7 vs 31.1 -- one step block of 32 lanes wins
    //  up to about three wavefronts per SIMD-quarter of the chip; round 2 had the switch at 1 024 candidates)
    if (N + 1 <= 32) return (N + 1 > 16 && count <= (int64_t)c->num_cus * 12) ? 32 : 16;
    // longer horizons: one wavefront per candidate for the small batches (at most ~4 wavefronts per SIMD; up to 64 steps that is
    // the single-launch variant with one step block: 400 candidates 36 vs 43 us per step, 2 600: equal), 16 lanes beyond -- 4
    // candidates per wavefront, rows stored straight to memory in 128-byte runs on 128-byte lines (state_row_stride).  Round 1
    // had one wavefront per candidate + LDS-staged linear copy-out for the large batches: with rows that start on line boundaries
    // the direct stores win at every horizon measured (profiles/r02_lanes_per_candidate.txt: cfg3 0.210 -> 0.155 ms, cfg4
    // 1.586 -> 1.487 ms, cfg5 2.738 -> 2.513 ms; 200 000 candidates without obstacles, N = 40 .. 100: 44-56 % of the HBM peak
    // against 30-50 %)
    if (count > (int64_t)c->num_cus * 16) return 16;
    return 64;
}

template <int G>
void launch_eval_g(rp_ctx *c, const KArgs &ka, int grid, bool mat, bool cin, int block) {
    if (mat) { if (cin) launch_eval_t<G, true, true>(c, ka, grid, block); else launch_eval_t<G, true, false>(c, ka, grid, block); }
    else     { if (cin) launch_eval_t<G, false, true>(c, ka, grid, block); else launch_eval_t<G, false, false>(c, ka, grid, block); }
}

// block: threads per workgroup of the batch's launch (eval_block); every other launch -- one candidate, a round's list -- takes RP_BLOCK
void launch_eval(rp_ctx *c, const KArgs &ka, int grid, bool mat, bool cin, int G, int block = RP_BLOCK) {
    if (G == 16) launch_eval_g<16>(c, ka, grid, mat, cin, block);
    else if (G == 32) launch_eval_g<32>(c, ka, grid, mat, cin, RP_BLOCK);
    else launch_eval_g<64>(c, ka, grid, mat, cin, RP_BLOCK);
}

// Threads per workgroup of the evaluation kernel of a batch (two-kernel path).  A workgroup's slot on the CU is freed when its LAST
// wavefront is done, and the four wavefronts of a 256-thread workgroup finish apart (in-kernel stamps on cfg3: first wavefront
// 30 k cycles, workgroup 42 k): one wavefront per workgroup frees a slot as soon as ITS wavefront is done.  Measured (A/B on one
// box, profiles/r04_block_ab.txt), kernel time 256 -> 64 threads: costs-only plans cfg4 414 -> 378 us, cfg5 1 027 -> 934 us,
// cfg3 89.8 -> 88 us; plans that write state rows cfg3 102.7 -> 100.8, cfg4 1 080 -> 1 086, cfg5 2 120 -> 2 145, cfg3f 137 -> 142
// (the row stores of four wavefronts that march together fill whole DRAM pages).  Four times the block partials: a batch whose
// 256-thread grid would stay under the fold threshold pays a fold kernel for them (cfg3: +5 us for -2).  So: one wavefront per
// workgroup for costs-only plans of batches that fold anyway.  RP_AMD_EVAL_BLOCK=64|256 pins the choice.
int eval_block(const rp_ctx *c, const KArgs &ka, int64_t count, int G, bool mat) {
    if (G != 16 || stage_out_applies(c, ka, G, mat) || ka.single_index || ka.index_list) return RP_BLOCK;
    if (c->opt.eval_block == 64 || c->opt.eval_block == RP_BLOCK) return c->opt.eval_block;
    return (!mat && count > (int64_t)kFoldThreshold * (RP_BLOCK / 16)) ? 64 : RP_BLOCK;
}

template <int G, bool CIN>
void launch_lon_t(rp_ctx *c, const KArgs &ka, int grid) {
    const size_t tbytes = (size_t)ka.table_words * sizeof(double);
    c->kargs_gl.k = ka;   // (the grid part is filled by rp_plan / run_pipeline and stays until the next plan: rp_eval_one)
    if (tbytes <= kLdsTableLimit)
        launch_block(c, (const void *)rp_lon_kernel<G, CIN, true>, grid, RP_BLOCK, tbytes, &c->kargs_gl, sizeof(KArgsGL));
    else
        launch_block(c, (const void *)rp_lon_kernel<G, CIN, false>, grid, RP_BLOCK, 0, &c->kargs_gl, sizeof(KArgsGL));
}

// longitudinal profiles of the pairs [ka.pair_begin, ka.pair_begin + ka.pair_count)
void launch_lon(rp_ctx *c, const KArgs &ka, bool cin) {
    if (ka.pair_count <= 0) return;
    const int G = (ka.N + 1 <= 32) ? 32 : 64;
    const int gpb = RP_BLOCK / G;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ka.pair_count + gpb - 1) / gpb, (int64_t)c->num_cus * kBlocksPerCU));
    if (G == 32) { if (cin) launch_lon_t<32, true>(c, ka, grid); else launch_lon_t<32, false>(c, ka, grid); }
    else         { if (cin) launch_lon_t<64, true>(c, ka, grid); else launch_lon_t<64, false>(c, ka, grid); }
}

int eval_grid(const rp_ctx *c, int64_t count, int G, int block = RP_BLOCK) {
    const int gpb = block / G;
    int64_t blocks = (count + gpb - 1) / gpb;
    (void)c;
    return (int)std::max<int64_t>(1, blocks);   // the grid covers the batch: one candidate per lane group
}

void fill_common(const rp_ctx *c, const rp_params *p, const rp_cost *cost, KArgs &ka) {
    std::memset(&ka, 0, sizeof(ka));
    ka.dt = p->dt;
    ka.N = p->N; ka.factor = p->factor; ka.time_step0 = p->time_step0;
    ka.low_vel_mode = p->low_vel_mode; ka.lon_mode = p->lon_mode;
    ka.constraint_mask = p->constraint_mask; ka.flags = p->flags;
    for (int k = 0; k < 3; ++k) { ka.x0_lon[k] = p->x0_lon[k]; ka.x0_lat[k] = p->x0_lat[k]; }
    ka.x0_orientation = p->x0_orientation;
    ka.wheelbase = p->wheelbase; ka.wb_rear_axle = p->wb_rear_axle;
    ka.half_length = 0.5 * p->length; ka.half_width = 0.5 * p->width;   // reactive_planner.py:1026-1027
    ka.ego_radius = std::sqrt(ka.half_length * ka.half_length + ka.half_width * ka.half_width);
    ka.a_max = p->a_max; ka.v_switch = p->v_switch; ka.v_delta_max = p->v_delta_max;
    ka.kappa_max = std::tan(p->delta_max) / p->wheelbase;               // reactive_planner.py:985
    ka.cost_kind = cost->kind;
    if (cost->kind == RP_COST_FAILSAFE) {   // cost_function.py:82-92: w_a = 1, d* = 0, no v / s terms
        ka.w_a = 1.0; ka.desired_d = 0.0; ka.has_speed = ka.has_s = 0;
        ka.desired_speed = ka.desired_s = 0.0;
    } else {
        ka.w_a = cost->w_a; ka.desired_d = cost->desired_d;
        ka.has_speed = !std::isnan(cost->desired_speed);
        ka.has_s = !std::isnan(cost->desired_s);
        ka.desired_speed = ka.has_speed ? cost->desired_speed : 0.0;
        ka.desired_s = ka.has_s ? cost->desired_s : 0.0;
    }
    ka.tables = c->d_tables; ka.n_ref = c->n_ref; ka.search_iters = c->search_iters;
    ka.n_buckets = c->n_buckets; ka.table_words = c->table_words; ka.bucket_inv_h = c->bucket_inv_h;
    ka.pos_first = c->h_pos.empty() ? 0.0 : c->h_pos.front();
    ka.pos_last = c->h_pos.empty() ? 0.0 : c->h_pos.back();
    ka.c_yaw = 1e5 / p->dt;
    ka.c_kdot = p->dt * p->v_delta_max / p->wheelbase;
    ka.proj_d_limit = c->proj_d_limit;
    ka.obs = c->obs;
    ka.debug = c->d_debug;
    ka.has_obstacles = (c->obs.n_sobb + c->obs.n_tri + c->obs.n_circ > 0 || (c->obs.n_dyn > 0 && c->obs.n_steps > 0)) ? 1 : 0;
    ka.ticket_if_none = 1;   // (gate == nullptr: a plan on its own)
}

// Part of the reference-table block a grid plan can touch (single-launch variant: every workgroup stages the block in
// LDS -- 28 KB x 465 workgroups on cfg2 -- although one replanning cycle moves over a few dozen metres of a route of
// hundreds).  s of every valid step lies between the bounds the Hermite form of the longitudinal polynomial gives:
//   velocity keeping (quartic, v cubic from (v0, a0) to (vd, 0)):  v in [min(v0, vd) - c T |a0|, max(v0, vd) + c T |a0|], c = 4/27
//   stopping (quintic to (sf, 0, 0)):  s in [min(s0, sf), max(s0, sf)] -+ (0.2 |v0| T + 0.0173 |a0| T^2)
// An estimate, not a guarantee the kernel relies on: the kernel checks every item against [win_s_lo, win_s_hi) and stages
// the whole block when one falls outside (steps beyond the end of the route, NaN samples, ...).
void table_window(const rp_ctx *c, const rp_params *p, const rp_grids *g, KArgs &ka) {
    ka.win_n = 0;   // whole block
    const int n = c->n_ref;
    if (n < 96 || c->n_buckets <= 0 || g->nT <= 0 || g->nL <= 0 || !c->opt.table_window) return;
    double Tmax = g->T[0], Lmin = g->L[0], Lmax = g->L[0];
    for (int i = 1; i < g->nT; ++i) Tmax = std::max(Tmax, g->T[i]);
    for (int i = 1; i < g->nL; ++i) { Lmin = std::min(Lmin, g->L[i]); Lmax = std::max(Lmax, g->L[i]); }
    const double s0 = p->x0_lon[0], v0 = p->x0_lon[1], a0 = std::fabs(p->x0_lon[2]);
    double lo, hi;
    if (p->lon_mode == RP_LON_STOPPING) {
        const double over = 0.2 * std::fabs(v0) * Tmax + 0.0173 * a0 * Tmax * Tmax;
        lo = std::min(s0, Lmin) - over; hi = std::max(s0, Lmax) + over;
    } else {
        const double dv = (4.0 / 27.0) * Tmax * a0;
        lo = s0 + std::min(0.0, (std::min(v0, Lmin) - dv) * Tmax); hi = s0 + std::max(0.0, (std::max(v0, Lmax) + dv) * Tmax);
    }
    if (!(lo <= hi) || !(hi - lo < 1e300)) return;   // NaN / inf samples
    const double *pos = c->h_pos.data();
    const int seg_lo = (int)(std::upper_bound(pos, pos + n, lo) - pos) - 1, seg_hi = (int)(std::upper_bound(pos, pos + n, hi) - pos) - 1;
    // an item in segment seg reads vertices seg - 2 .. seg + 3 (bucket fix-ups, both interpolation pairs)
    int w0 = std::max(0, seg_lo - 3), w1 = std::min(n, seg_hi + 5);
    int shift = 4;
    while ((1 << shift) < w1 - w0) ++shift;
    const int wn = 1 << shift;
    if (wn * 2 > n || wn > 128) return;              // not worth a window | the kernel holds at most 5 pieces per lane
    if (w0 + wn > n) w0 = n - wn;
    ka.win_k0 = w0; ka.win_n = wn; ka.win_shift = shift;
    ka.win_s_lo = w0 == 0 ? pos[0] : pos[w0 + 2];
    ka.win_s_hi = w0 + wn == n ? pos[n - 1] : pos[w0 + wn - 4];
    // bucket entries of [win_s_lo, win_s_hi): the kernel's index is (int)((s - pos[0]) * bucket_inv_h), clamped to n_buckets - 1
    const int b0 = std::max(0, (int)((ka.win_s_lo - pos[0]) * c->bucket_inv_h) - 1);
    const int b1 = std::min(c->n_buckets - 1, (int)((ka.win_s_hi - pos[0]) * c->bucket_inv_h) + 1);
    ka.win_b0 = b0; ka.win_nb = b1 - b0 + 1;
}

// the kernels' 28 words -> the caller's rp_result (its struct_size header stays the caller's)
inline void result_from_core(rp_result *dst, const ResultCore &src) { std::memcpy(&dst->best_index, &src, sizeof(ResultCore)); }

int validate(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_result *res) {
    if (!c) return RP_EINVAL;
    if (!p || !cost || !res) return fail(c, RP_EINVAL, "null params / cost / result");
    if (p->struct_size != sizeof(rp_params) || cost->struct_size != sizeof(rp_cost) || res->struct_size != sizeof(rp_result))
        return fail(c, RP_EABI, "struct_size of rp_params / rp_cost / rp_result is not this library's (RP_ABI_VERSION " + std::to_string(RP_ABI_VERSION) +
                                ": set it with RP_*_INIT; built against another rp_amd.h?)");
    if (!c->d_tables) return fail(c, RP_ESTATE, "rp_set_reference has not been called");
    if (p->N < 1 || p->N > 4094) return fail(c, RP_EINVAL, "N out of range [1, 4094]");
    if (!(p->dt > 0.0)) return fail(c, RP_EINVAL, "dt must be positive");
    if (p->lon_mode != RP_LON_VELOCITY_KEEPING && p->lon_mode != RP_LON_STOPPING)
        return fail(c, RP_EINVAL, "unknown lon_mode");
    if (cost->kind < RP_COST_DEFAULT || cost->kind > RP_COST_EXTERNAL) return fail(c, RP_EINVAL, "unknown cost kind");
    if (!(p->wheelbase > 0.0)) return fail(c, RP_EINVAL, "wheelbase must be positive");
    if (!(p->length >= 0.0 && p->length < 1e300 && p->width >= 0.0 && p->width < 1e300))   // (NaN / inf: the grid over the static shapes has no meaning)
        return fail(c, RP_EINVAL, "length / width must be finite and non-negative");
    return RP_OK;
}

// The winner's polynomial coefficients (sampling.py:253-270, polynomial_trajectory.py:292-360) from its index and the
// grids staged by this call -- the closed forms of rp_device.h with plain divisions.
void host_winner_coeffs(const rp_ctx *c, const KArgs &ka, bool cin, rp_result *r) {
    const double nan = std::nan("");
    for (int k = 0; k < 6; ++k) r->best_lon_coeffs[k] = r->best_lat_coeffs[k] = nan;
    r->best_lat_T = nan;
    const int64_t w = r->best_index;
    if (w < 0) return;
    if (cin) {
        // (the pinned staging buffer still holds them: every plan rewrites it, and with it what this context calls its last plan)
        if (w >= c->cin_count || !c->cin_lon) return;
        const double *lon = c->cin_lon, *lat = c->cin_lat;
        for (int k = 0; k < 6; ++k) { r->best_lon_coeffs[k] = lon[6 * w + k]; r->best_lat_coeffs[k] = lat[6 * w + k]; }
        return;
    }
    const size_t need = sizeof(double) * ((size_t)ka.nT + ka.nL + ka.nD);
    if (c->staged.size() < need || ka.nL <= 0 || ka.nD <= 0) return;
    const double *T_ = reinterpret_cast<const double *>(c->staged.data()), *L_ = T_ + ka.nT, *D_ = L_ + ka.nL;
    const int64_t nLD = (int64_t)ka.nL * ka.nD;
    const int iT = (int)(w / nLD), rem = (int)(w - (int64_t)iT * nLD), iL = rem / ka.nD, iD = rem - iL * ka.nD;
    if (iT >= ka.nT) return;
    auto quintic = [](double p0, double v0, double a0, double pf, double T, double *o) {
        const double T2 = T * T, bp = pf - (p0 + v0 * T + 0.5 * a0 * T2), bv = -(v0 + a0 * T), ba = -a0;
        const double Tbv = T * bv, T2ba = T2 * ba, T3 = T2 * T;
        o[0] = p0; o[1] = v0; o[2] = 0.5 * a0;
        o[3] = (20.0 * bp - 8.0 * Tbv + T2ba) / (2.0 * T3);
        o[4] = (-30.0 * bp + 14.0 * Tbv - 2.0 * T2ba) / (2.0 * T3 * T);
        o[5] = (12.0 * bp - 6.0 * Tbv + T2ba) / (2.0 * T3 * T2);
    };
    const double T = T_[iT];
    double *lon = r->best_lon_coeffs;
    if (ka.lon_mode == RP_LON_STOPPING) {
        quintic(ka.x0_lon[0], ka.x0_lon[1], ka.x0_lon[2], L_[iL], T, lon);
    } else {   // quartic to (v_d, 0)
        const double bv = L_[iL] - ka.x0_lon[1] - ka.x0_lon[2] * T, ba = -ka.x0_lon[2];
        lon[0] = ka.x0_lon[0]; lon[1] = ka.x0_lon[1]; lon[2] = 0.5 * ka.x0_lon[2];
        lon[3] = (3.0 * bv - T * ba) / (3.0 * T * T);
        lon[4] = (T * ba - 2.0 * bv) / (4.0 * T * T * T);
        lon[5] = 0.0;
    }
    double lat_T = T;
    if (ka.low_vel_mode) {
        const double sg = ((((lon[5] * T + lon[4]) * T + lon[3]) * T + lon[2]) * T + lon[1]) * T + lon[0] - ka.x0_lon[0];
        lat_T = sg <= 0.0 ? T : sg;
    }
    quintic(ka.x0_lat[0], ka.x0_lat[1], ka.x0_lat[2], D_[iD], lat_T, r->best_lat_coeffs);
    r->best_lat_T = lat_T;
}

// Host wait for the completion ticket `seq` in the pinned result block, by the context's wait mode (rp_set_wait_mode):
//   RP_WAIT_SPIN   poll + pause            RP_WAIT_YIELD   poll + sched_yield
//   RP_WAIT_EVENT  (plans: no ticket -- see wait_plan) here as YIELD
// true: the ticket is there; false: 200 ms have passed (the caller synchronises the stream)
bool poll_ticket(const rp_ctx *c, unsigned long long seq) {
    const volatile unsigned long long *flag = &reinterpret_cast<ResultBlock *>(c->h_result)->seq;
    const bool yield = c->opt.wait_mode != RP_WAIT_SPIN;
    const auto t_start = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
        if ((spins & (yield ? 0x3Fu : 0x3FFu)) == (yield ? 0x3Fu : 0x3FFu) &&
            std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(200)) return false;
        if (yield) sched_yield();
        else __builtin_ia32_pause();
    }
}
int wait_ticket(rp_ctx *c, unsigned long long seq, hipStream_t stream = nullptr) {
    if (poll_ticket(c, seq)) return RP_OK;
    ++c->wait_fallbacks;   // (200 ms without the ticket: read-only option "wait_fallbacks")
    HIP_TRY(c, hipStreamSynchronize(stream ? stream : c->stream));
    return RP_OK;
}
// End of a plan whose kernels are on the stream: with a ticket, wait for it; without (RP_WAIT_EVENT: c->spin_wait off when the plan
// was launched) sleep on the event recorded behind the last launch
int wait_plan(rp_ctx *c, bool ticket, unsigned long long seq) {
    if (ticket) return wait_ticket(c, seq);
    if (c->ev_done) {
        HIP_TRY(c, hipEventRecord(c->ev_done, c->stream));
        HIP_TRY(c, hipEventSynchronize(c->ev_done));
        return RP_OK;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RP_OK;
}

// The cost-ordered collision stage (rp_kernels.h: LazyCtl): pass 1 without the collision query -> thresholds and candidate lists
// -> rounds of (evaluation kernel in list mode WITH the query, one-workgroup epilogue) until a round has a free candidate.
// The longitudinal profiles are in place (run_pipeline launched rp_lon_kernel).  *done: the result block in pinned host memory is
// final (winner or "no candidate survives"); otherwise the caller runs the eager kernel over the whole batch.
template <typename LaunchEval>
int run_lazy(rp_ctx *c, const KArgs &ka, bool cin, int G, const int &grid /* workgroups of pass 1: known once it is launched */, bool want_rows,
             LaunchEval &launch_main_eval, bool *done, int *rounds) {
    *done = false;
    *rounds = 0;
    (void)G;
    const int n = ka.N + 1;
    const int64_t count = ka.count;
    ResultBlock *drb = reinterpret_cast<ResultBlock *>(c->d_result);
    ResultBlock *hrb_dev = reinterpret_cast<ResultBlock *>(c->h_result_dev);
    ResultBlock *hrb_host = reinterpret_cast<ResultBlock *>(c->h_result);
    int rc;
    // -- pass 1: kinematics + cost of every candidate, no collision query; workgroup 0 clears the control block
    KArgs k1 = ka;
    k1.flags |= RP_FLAG_SKIP_COLLISION;
    k1.lazy_ctl = c->d_lazy_ctl;
    k1.sweep_init = c->d_sweep;   // (should the lists run dry, a bounded sweep takes over: pipeline_begin)
    k1.states = nullptr;
    if ((rc = launch_main_eval(k1, false)) != RP_OK) return rc;
    int n_partials = grid, p_first = 0;
    const int fold_threshold = c->opt.fold_threshold;
    if (n_partials > fold_threshold) {
        hipLaunchKernelGGL(rp_fold_partials_kernel, dim3(kFoldPartials), dim3(64), 0, c->stream, c->d_partials, ka.partials_cap, n_partials,
                           c->cap_partials);
        p_first = c->cap_partials;
        n_partials = kFoldPartials;
    }
    // -- thresholds and lists
    GatherArgs ga;
    ga.status = ka.status; ga.cost = ka.cost; ga.partials = c->d_partials; ga.lists = c->d_lazy_lists; ga.ctl = c->d_lazy_ctl;
    ga.count = count; ga.cand_begin = ka.cand_begin;
    ga.partials_cap = ka.partials_cap; ga.partials_first = p_first; ga.n_partials = n_partials; ga.pad_ = 0;
    ga.hist = c->d_lazy_hist; ga.pad2_ = 0;
    for (int l = 0; l < RP_LAZY_LEVELS; ++l) { ga.cap[l] = kLazyCap[l]; ga.off[l] = kLazyOff[l]; ga.target[l] = kLazyTarget[l]; }
    const int ggrid = (int)std::max<int64_t>(16, std::min<int64_t>((count + 2047) / 2048, (int64_t)c->num_cus));
    launch_block(c, (const void *)rp_lazy_hist_kernel, ggrid, RP_GATHER_THREADS, 0, &ga, sizeof(ga));
    launch_block(c, (const void *)rp_lazy_gather_kernel, ggrid, RP_GATHER_THREADS, 0, &ga, sizeof(ga));
    // -- rounds
    const int ns = state_row_stride(c, n, false);
    uint32_t cnt[RP_LAZY_LEVELS] = {(uint32_t)kLazyCap[0], 0, 0};   // (round 0 is launched for a full list: its size is on the device only)
    for (int l = 0; l < RP_LAZY_LEVELS; ++l) {
        if (l > 0 && cnt[l] == 0) continue;
        const int rcount = (int)std::min<uint32_t>(cnt[l], (uint32_t)kLazyCap[l]);
        const int rgrid = (rcount + RP_BLOCK / 64 - 1) / (RP_BLOCK / 64);
        if (want_rows && (rc = grow(c, c->d_lazy_states, c->cap_lazy_states, (size_t)rcount * RP_N_ARRAYS * (size_t)ns)) != RP_OK) return rc;
        KArgs kr = ka;
        kr.index_list = c->d_lazy_lists + kLazyOff[l];
        kr.list_count = &c->d_lazy_ctl->count[l];
        kr.list_cap = kLazyCap[l];
        kr.states = want_rows ? c->d_lazy_states : nullptr;
        kr.row_stride = ns;
        kr.tail_split = 0;
        kr.lazy_ctl = nullptr;
        launch_eval(c, kr, rgrid, want_rows, cin, 64);   // one wavefront per candidate
        const unsigned long long seq = ++c->seq;
        hrb_host->seq = 0;
        FinArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.status = ka.status; fa.cost = ka.cost; fa.states = kr.states; fa.partials = c->d_partials;
        fa.partials_cap = ka.partials_cap; fa.partials_first = 0;
        fa.dev_out = drb; fa.host_out = hrb_dev;
        fa.count = count; fa.cand_begin = ka.cand_begin; fa.seq = seq;
        fa.debug = c->d_debug;
        fa.N = ka.N; fa.n_partials = rgrid; fa.count_inline = 1; fa.copy_states = want_rows ? 1 : 0;
        fa.row_stride = ns; fa.tail_split = 0; fa.inv_n = (uint32_t)(0x100000000ull / (unsigned long long)n) + 1u;
        fa.scratch = c->d_sel_scratch;
        fa.list = kr.index_list; fa.list_count = kr.list_count; fa.lazy = c->d_lazy_ctl; fa.list_cap = kLazyCap[l]; fa.level = l;
        launch_block(c, (const void *)rp_finalize_kernel, 1, RP_FIN_THREADS, 0, &fa, sizeof(fa));
        HIP_TRY(c, hipGetLastError());
        if ((rc = wait_ticket(c, seq)) != RP_OK) return rc;
        // what the round left: winner, or the state of the stage (FinalizeOut.w_coeffs[0..3]: list sizes, overflow, checked, feasible)
        unsigned long long ex[4];
        std::memcpy(ex, hrb_host->w_coeffs, sizeof(ex));
        const uint32_t overflow = (uint32_t)(ex[1] >> 32), checked = (uint32_t)ex[2];
        const unsigned long long feasible = ex[3];
        cnt[0] = (uint32_t)ex[0]; cnt[1] = (uint32_t)(ex[0] >> 32); cnt[2] = (uint32_t)ex[1];
        *rounds = l + 1;
        if (c->opt.lazy_trace)   // (diagnostic: profiles/probe_lazy_trace.py)
            std::fprintf(stderr, "cost-ordered stage, round %d: list sizes %u %u %u, overflow bits %x, checked %u of %llu feasible, winner %lld\n", l,
                         cnt[0], cnt[1], cnt[2], overflow, checked, feasible, (long long)hrb_host->r.best_index);
        if (overflow & ((2u << l) - 1u)) return RP_OK;              // this round's list (or an earlier one) was incomplete: not conclusive
        if (hrb_host->r.best_index >= 0) { *done = true; return RP_OK; }
        if ((unsigned long long)checked >= feasible) { *done = true; return RP_OK; }   // every feasible candidate collides: no winner
    }
    return RP_OK;   // lists exhausted without a free candidate: the eager kernel decides
}

// eval -> finalize (-> count for huge batches) (-> winner re-evaluation when nothing was materialised).
// The result block lands in pinned host memory straight from the kernels; one stream sync per plan.
int pipeline_wait(rp_ctx *c, rp_result *result, double *best_states);
// chain_last: the level is the last one of its chain (or a plan on its own): its epilogue reports whatever it found
int pipeline_begin(rp_ctx *c, KArgs &ka, bool mat, bool cin, bool skip_eval, bool want_rows, bool chain_last = true) {
    if (c->time_whole) c->plan_t0 = std::chrono::steady_clock::now();
    c->path_adaptive = false;
    double *const best_states = want_rows ? reinterpret_cast<double *>(c) : nullptr;   // (only its being non-null matters below)
    const int n = ka.N + 1;
    int rc;
    if ((rc = ensure_result(c, n)) != RP_OK) return rc;
    ResultBlock *drb = reinterpret_cast<ResultBlock *>(c->d_result);
    ResultBlock *hrb_dev = reinterpret_cast<ResultBlock *>(c->h_result_dev);
    const int64_t count = ka.count;
    int G = lanes_per_candidate(c, ka.N, count, mat);
    // small batches: one launch computes the longitudinal profiles and evaluates (rp_eval_kernel<.., LON_FUSED>)
    int fused_pairs = 0;
    size_t fused_lds = 0;
    if (skip_eval) {
        // rp_select: nothing is evaluated again except, possibly, the winner -- on the launch path of the plan the costs
        // belong to (its profile rows exist only if that plan took the two-kernel path; same code path, same bits)
        fused_lds = c->last_fused_lds;
        fused_pairs = ka.lds_pairs;
        if (c->last_G) G = c->last_G;
    } else {
        // (grouped explicit polynomials: the two-kernel path -- the single-launch prologue counts one pair per candidate there)
        if (c->opt.fused_lon && !(cin && ka.pair_of)) fused_lds = fused_lon_lds(c, ka, count, G, cin, mat, &fused_pairs);
    }
    ka.lds_pairs = fused_pairs;
    if (!skip_eval) {
        c->last_fused_lds = fused_lds; c->last_G = G;
        ka.tail_split = state_layout(c, n, G, fused_lds != 0, !fused_lds && stage_out_applies(c, ka, G, mat), &ka.row_stride);   // (rp_select keeps the plan's)
    }
    int block = c->last_block ? c->last_block : RP_BLOCK;   // (rp_select: nothing of the batch is launched again)
    if (!skip_eval) { block = fused_lds ? RP_BLOCK : eval_block(c, ka, count, G, mat); c->last_block = block; }
    const int grid = fused_lds ? (int)((count + RP_BLOCK / G - 1) / (RP_BLOCK / G)) : eval_grid(c, count, G, block);
    // (rp_cost_kernel / rp_chunk_kernel: one partial per 64 candidates, fewer than the 16-lane kernel's; a bounded sweep keeps its pass 1's
    //  partials behind its own: room for two sets)
    const int part_grid = 2 * std::max(grid, (int)((count + 15) / 16)) + 2;
    if (std::max(part_grid, kLazyCap[RP_LAZY_LEVELS - 1] / (RP_BLOCK / 64)) > c->cap_partials) {
        if (c->d_partials) HIP_TRY(c, hipFree(c->d_partials));
        c->d_partials = nullptr;
        c->cap_partials = 0;
        int want = std::max(std::max(part_grid + part_grid / 4, c->num_cus * kBlocksPerCU), kLazyCap[RP_LAZY_LEVELS - 1] / (RP_BLOCK / 64));
        HIP_TRY(c, hipMalloc((void **)&c->d_partials, (size_t)RP_PARTIAL_BYTES * ((size_t)want + kFoldPartials)));
        c->cap_partials = want;
    }
    ka.partials = c->d_partials;
    ka.partials_cap = c->cap_partials + kFoldPartials;   // slots per array (the folded ones sit behind the workgroups')
    int n_partials = grid;
    int main_grid = grid;   // workgroups (= block partials) of the batch's evaluation launch: fewer when rp_cost_kernel takes it

    const bool small = count <= RP_FINALIZE_MAX;
    const bool copy_states = mat && best_states != nullptr && count > 0;
    const bool winner_pass = !mat && best_states != nullptr && count > 0;
    c->last_rows_on_device = copy_states || count == 0;
    const bool ticket = c->spin_wait;   // (large batches too: their selection epilogue hands the ticket over itself)
    ResultBlock *hrb_host = reinterpret_cast<ResultBlock *>(c->h_result);
    // (a level of a chain: the chain's one ticket, reset by the chain's first level)
    unsigned long long seq = ka.gate ? ka.gate_seq : ++c->seq;
    if (ticket && !ka.gate) hrb_host->seq = 0;
    unsigned long long fin_seq = (ticket && !winner_pass) ? seq : 0ull;
    // (A selection epilogue run by the evaluation kernel's last workgroup was tried and measured slower -- cfg2: eval
    //  21.6 -> 37.8 us: every workgroup then pays an agent-scope release fence, an L2 write-back, before its ticket.)
    const bool timed = c->profiling > 0 && !skip_eval && !ka.gate && (c->calls++ % (unsigned long long)c->profiling) == 0;
    bool time_valid = false;
    const auto tp0 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    bool lazy_done = false;   // the cost-ordered collision stage delivered the result (its last round handed the ticket over)
    bool swept = false;       // ... ran as a bounded sweep: the common epilogue picks the winner from the sweep's partials
    int pass1_grid = 0;       //     and takes the batch's counters from its pass 1's (pass1_grid of them, behind the sweep's)
    bool swept_totals = false;   //  ... or from the control block of the list stage the sweep took over from
    int swept_grid = 0;          //  workgroups (= block partials) of the sweep's launch
    if (c->epilogue_dirty && (!ka.gate || ka.gate_level == 1)) {   // (the last chain of kernels on this context did not reach its result: see rp_ctx::epilogue_dirty)
        HIP_TRY(c, hipMemsetAsync(c->d_sel_scratch, 0, RP_SEL_SCRATCH * sizeof(unsigned long long), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_lazy_hist, 0, (RP_LAZY_BINS + 1) * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_lazy_ctl, 0, sizeof(LazyCtl), c->stream));
    }
    c->epilogue_dirty = true;
    if (!skip_eval) {
        // grids that did not fit the evaluation kernel's launch block: in rp_lon_kernel's (two-kernel path: its workgroup 0 publishes
        // them to d_stage for the kernels behind it) or by a host-to-device copy (a 5-us blit kernel on the stream)
        bool lon_inline = ka.grids_inline != 0, lon_publish = false;
        if (c->grids_pending) {
            const size_t sbytes = c->staged.size();
            if (count > 0 && !fused_lds && !cin && sbytes + 8 <= sizeof(c->kargs_gl.grid) && c->opt.lon_publish) {
                std::memcpy(c->kargs_gl.grid, c->staged.data(), sbytes);
                lon_inline = lon_publish = true;
            } else {
                HIP_TRY(c, hipMemcpyAsync(c->d_stage, c->h_stage, sbytes, hipMemcpyHostToDevice, c->stream));
            }
            c->grids_pending = false;
            c->staged_on_device = true;
        }
        // longitudinal profiles of every (T, longitudinal sample) pair touched by the candidate range
        if (count > 0 && !fused_lds) {
            if (cin) { ka.pair_begin = 0; ka.pair_count = ka.pair_of ? c->cin_groups : count; }
            else {
                ka.pair_begin = ka.cand_begin / ka.nD;
                ka.pair_count = (ka.cand_begin + count - 1) / ka.nD + 1 - ka.pair_begin;
            }
            if ((rc = grow(c, c->d_profile, c->cap_profile, (size_t)ka.pair_count * PF_FIELDS * (size_t)n)) != RP_OK) return rc;
            if ((rc = grow(c, c->d_pair_hdr, c->cap_pair_hdr, (size_t)ka.pair_count)) != RP_OK) return rc;
            ka.profile = c->d_profile;
            ka.pair_hdr = c->d_pair_hdr;
            KArgs kl = ka;
            kl.grids_inline = lon_inline ? 1 : 0;
            kl.publish_grids = lon_publish ? 1 : 0;
            launch_lon(c, kl, cin);
        }
        // the evaluation kernel's duration: events attached to the launch itself (hipExtModuleLaunchKernel; RP_AMD_EVENT_BRACKET=1:
        // two hipEventRecord around it, which adds the dispatch and completion handling of the bracket -- ~2.5 us on a 14-us kernel)
        const bool by_launch = timed && count > 0 && !c->opt.event_bracket;
        // Launches of grid batches that keep no state rows -- production-mode plans, the first pass of the cost-ordered stage -- have
        // the time axis INSIDE the lane (rp_kernels.h):
        //   rp_cost_kernel   one lane per candidate, the lane walks the steps.  Needs wavefronts of 64 candidates for every SIMD, four
        //                    each to hide its own latency: from 262 144 candidates on (1 024 per CU).  (Round 4 took it from 61 440 on,
        //                    one wavefront per SIMD, where it was 1.1 x the 16-lane kernel: profiles/probe_cost_threshold.py.)
        //   rp_chunk_kernel  one lane per candidate and step block of 16, one wavefront per 64 candidates and step block (2 .. 7 blocks:
        //                    horizons of 17 .. 112 steps), so that a batch the size of cfg3 (62 496 candidates, N = 60: 3 906 wavefronts
        //                    of 17 iterations) fills the chip -- every batch of the two-kernel path below the cost kernel's range.
        // Options "cost_kernel" / "chunk_kernel" = 0 / 1: never / whenever the variant applies (tests, A/B); -1: by batch, as above.
        // WITH the eager collision query (no cost-ordered stage: back-off after an exhausted stage, RP_COLLISION_EAGER) the same rule:
        // thresholds from profiles/probe_chunk_threshold.py.
        const int n_steps = ka.N + 1;
        const int chunk_nb = (n_steps + RP_CHUNK_STEPS - 1) / RP_CHUNK_STEPS;   // step blocks of a candidate = wavefronts of a workgroup
        const int chunk_G = (chunk_nb >= 2 && chunk_nb <= RP_CHUNK_MAX_BLOCKS) ? chunk_nb : 0;
        const bool lane_kernel_ok = count > 0 && !fused_lds && !cin && !(ka.flags & RP_FLAG_DRAW_ALL) && !ka.single_index && !ka.index_list &&
                                    ka.cost_kind != RP_COST_EXTERNAL;
        // (rp_cost_kernel addresses the profile rows by a 32-bit byte offset from the launch's first pair: below 4 GiB of rows only --
        //  332 000 pairs at N = 100; beyond, the 16-lane kernel, whose offsets are relative to each wavefront's first pair)
        const bool cost_rows_ok = (unsigned long long)ka.pair_count * PF_FIELDS * (unsigned long long)(ka.N + 1) * 8ull < (1ull << 32);
        const bool cost_big = count >= (int64_t)c->num_cus * 1024;
        // 0: rp_eval_kernel, 1: rp_cost_kernel, 2: rp_chunk_kernel -- for a launch with collision level `coll`
        auto lane_kernel_for = [&](int coll) -> int {
            if (!lane_kernel_ok) return 0;
            if (c->opt.chunk_kernel == 1 && chunk_G) return 2;
            if (!cost_rows_ok) return (chunk_G && c->opt.chunk_kernel != 0 && coll == 0) ? 2 : 0;
            if (c->opt.cost_kernel == 1) return 1;
            // WITH the eager query inside the kernel the step blocks of a candidate want to run one after the other -- a candidate that has
            // collided is not asked again, and in traffic most collide early: the 16-lane kernel (cfg3 95 us, chunk 126 us, lane 152 us;
            // cfg3 in mostly-free traffic 113 / 128 / 182 us: profiles/probe_chunk_kernel.py), from 262 144 candidates on the lane kernel
            if (coll > 0) return (c->opt.cost_kernel != 0 && cost_big) ? 1 : 0;
            // without the query (profiles/probe_chunk_threshold.py, r05, kernel us 16-lane / lane / chunk):
            //   N = 60 (4 blocks)  24 192: 22.7 / 52.6 / 24.1   32 256: 28.7 / 52.6 / 25.0   62 496: 49.9 / 53.0 / 34.0   124 992: 97 / 78 / 59
            //                     187 488: 133 / 93 / 81       249 984: 175 / 99 / 101
            //   N = 30 (2 blocks)  29 760: 21.3 / 31.3 / 22.4   59 520: 36.0 / 32.8 / 26.9
            //   N = 100 (7 blocks) 65 024: 59 / 85 / 60        130 048: 108 / 101 / 112     512 064: 369 / 250 / 368  (no range where chunk wins)
            const int64_t chunk_waves = ((count + RP_CHUNK_BLOCK - 1) / RP_CHUNK_BLOCK) * chunk_G;
            const bool chunk_ok = c->opt.chunk_kernel != 0 && chunk_G && chunk_G <= 4 && chunk_waves >= (int64_t)c->num_cus * kChunkMinWavesPerCU;
            if (c->opt.cost_kernel != 0 && count >= (int64_t)c->num_cus * (chunk_ok ? kCostMinPerCU : (ka.N + 1 <= 64 ? 240 : 448))) return 1;
            return chunk_ok ? 2 : 0;
        };
        auto launch_main_eval = [&](const KArgs &k, bool mat_) -> int {   // the batch's evaluation kernel, timed if this step is
            c->timed_by_launch = false;
            c->time_next_launch = by_launch;
            if (timed && !by_launch) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
            const auto te0 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            const int coll = collision_level(k);
            const int which = mat_ ? 0 : lane_kernel_for(coll);
            if (which == 1) {
                main_grid = (int)((count + RP_COST_BLOCK - 1) / RP_COST_BLOCK);
                c->last_kernel = RP_KERNEL_COST;
                const void *fn = k.low_vel_mode
                    ? (coll == 2 ? (const void *)rp_cost_kernel<true, 2> : coll == 1 ? (const void *)rp_cost_kernel<true, 1> : (const void *)rp_cost_kernel<true, 0>)
                    : (coll == 2 ? (const void *)rp_cost_kernel<false, 2> : coll == 1 ? (const void *)rp_cost_kernel<false, 1> : (const void *)rp_cost_kernel<false, 0>);
                launch_kargs(c, fn, main_grid, RP_COST_BLOCK, 0, k);
            } else if (which == 2) {
                main_grid = (int)((count + RP_CHUNK_BLOCK - 1) / RP_CHUNK_BLOCK);   // 64 candidates per workgroup, one wavefront per step block
                c->last_kernel = RP_KERNEL_CHUNK;
                launch_kargs(c, chunk_kernel_fn(k.low_vel_mode != 0, coll), main_grid, RP_CHUNK_BLOCK * chunk_G, rp_chunk_lds_bytes(chunk_G), k);
            } else {
                main_grid = grid;
                c->last_kernel = RP_KERNEL_EVAL;
                if (fused_lds) launch_eval_fused(c, k, grid, mat_, cin, fused_lds, G);
                else launch_eval(c, k, grid, mat_, cin, G, block);
            }
            if (c->timing) c->t_sum[4] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - te0).count();
            c->time_next_launch = false;
            if (timed && !by_launch) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
            time_valid = timed && (!by_launch || c->timed_by_launch);   // (a launch that could not carry the events: no duration this step)
            return RP_OK;
        };
        // Production-mode plans of large batches with obstacles: costs first, collision rounds over the cheapest candidates (the
        // reference's own order of work, reactive_planner.py:1031-1062) -- see LazyCtl in rp_kernels.h.  RP_AMD_LAZY=0 never,
        // =1 whenever the launch path allows it (tests), default: batches beyond the one-workgroup epilogue, unless the last
        // attempts had to fall back to the eager kernel (a scene where nearly everything collides).
        bool lazy_possible = count > 0 && !mat && !fused_lds && collision_level(ka) > 0 && !(ka.flags & RP_FLAG_DRAW_ALL) &&
                             ka.cost_kind != RP_COST_EXTERNAL && !ka.single_index && !ka.gate;   // (a chain of levels has no host in between)
        const int lazy_env = c->opt.lazy;   // (option "lazy": the tests switch paths)
        lazy_possible = lazy_possible && lazy_env != 0 && (c->collision_mode != RP_COLLISION_EAGER || lazy_env == 1);
        const bool always = lazy_env == 1 || (lazy_env == -1 && c->collision_mode == RP_COLLISION_COST_ORDERED);
        bool lazy_try = lazy_possible && (always || (!small && c->lazy_skip == 0));
        if (lazy_possible && !lazy_try && !always && !small && c->lazy_skip > 0) --c->lazy_skip;
        // RP_COLLISION_TIMED: the context's own clock decides (what a plan labels then depends on earlier timings); the default rule
        // below depends on the sequence of plans alone
        c->path_adaptive = lazy_possible && lazy_env == -1 && !small && c->collision_mode == RP_COLLISION_TIMED;
        if (c->path_adaptive) {
            int lg = 0;
            while ((count >> (lg + 1)) != 0) ++lg;
            const int regime = 2 * lg + (cin ? 1 : 0);
            if (regime != c->path_regime) { c->path_regime = regime; c->path_n[0] = c->path_n[1] = 0; c->path_since_probe = 0; }
            if (lazy_try) {   // (not while the back-off after an exhausted stage holds the plans eager anyway)
                if (c->path_n[1] == 0) lazy_try = true;                 // first plans of a regime: the cost-ordered stage
                else if (c->path_n[0] == 0) lazy_try = c->path_n[1] < 4;   // then the eager kernel once
                else {
                    lazy_try = c->path_us[1] <= c->path_us[0];
                    if (++c->path_since_probe >= 64) { lazy_try = !lazy_try; c->path_since_probe = 0; }
                }
            }
        }
        c->last_lazy = 0;
        // The cost-ordered stage as ONE bounded sweep (no histogram, no lists, no decision of the host between the launches): pass 1 --
        // costs of every candidate, no query -- and then the eager 16-lane kernel over the whole batch, in which a lane group evaluates
        // its candidate only while no cheaper FREE candidate is known (KArgs::sweep_bound: atomic minimum of the cost keys of the
        // candidates found free).  Every candidate cheaper than the winner is looked at -- its cost is below every value the bound
        // ever has -- so the winner, the candidates labelled in front of it and infeasible_count_collision are those of the eager
        // query; how many candidates BEHIND the winner were looked at before the bound closed depends on the order the wavefronts
        // ran in (rp_result.n_collision reports the count in front of the winner).
        // WHEN: as the fallback of the list rounds on large batches.  A winner near the top of the cost order is found by the first list
        // round for less (cfg3f 91 vs 188 us, cfg4 330 vs 492: the sweep launches every workgroup of the batch, if only to find nothing to
        // do); a winner deep in the cost order -- the three lists, 21 504 candidates, all collide -- leaves the eager kernel over the whole
        // batch or the sweep: cfg4 + road boundary 880 -> 602 us, cfg3 (62 496 candidates, 97 % collide) 145 -> 159 us: from 131 072
        // candidates on (profiles/probe_chunk_kernel.py).  Option "sweep" = 1: the sweep instead of the lists, whatever the size (tests).
        const bool sweep_first = lazy_try && c->opt.sweep == 1 && !fused_lds;
        const bool sweep_fallback = lazy_possible && c->opt.sweep != 0 && !fused_lds && !small && count >= (int64_t)c->num_cus * kSweepMinPerCU &&
                                    !c->path_adaptive;
        const int sweep_grid = eval_grid(c, count, 16, RP_BLOCK);   // the sweep's own variant: 16 lanes per candidate, 256-thread workgroups
        auto launch_sweep = [&](bool pass1) -> int {
            if (pass1) {
                KArgs k1 = ka;
                k1.flags |= RP_FLAG_SKIP_COLLISION;
                k1.states = nullptr;
                k1.sweep_init = c->d_sweep;
                k1.partials_first = sweep_grid;  // (behind the sweep's own partials)
                const int r1 = launch_main_eval(k1, false);
                if (r1 != RP_OK) return r1;
                pass1_grid = main_grid;
            }
            KArgs kb = ka;
            kb.states = nullptr;
            kb.sweep_bound = c->d_sweep;
            launch_kargs(c, sweep_kernel_fn(cin, collision_level(kb), kb.N + 1 <= 16), sweep_grid, RP_BLOCK, 0, kb);
            main_grid = sweep_grid;
            swept = true;
            swept_grid = sweep_grid;
            swept_totals = !pass1;   // (behind exhausted lists: the batch's counters are in the stage's control block, summed from pass 1)
            c->last_lazy = RP_PATH_SWEEP;
            return RP_OK;
        };
        if (sweep_first) {
            if ((rc = launch_sweep(true)) != RP_OK) return rc;
            c->lazy_penalty = c->lazy_skip = 0;
        } else
        if (lazy_try) {
            int lazy_rounds = 0;
            if ((rc = run_lazy(c, ka, cin, G, main_grid, best_states != nullptr, launch_main_eval, &lazy_done, &lazy_rounds)) != RP_OK) return rc;
            if (lazy_done) {
                c->last_lazy = 1; c->last_rows_on_device = best_states != nullptr;
                // A stage that needed its third list to find the winner (three rounds and their epilogues, host decisions in
                // between) costs more than the eager kernel on batches whose query is cheap: such a scene keeps the next 1, 2, 4 ..
                // 64 plans eager, like one where the stage ran dry.  A rule over the sequence of plans, not over a clock: the same
                // calls label the same candidates in every run (RP_COLLISION_TIMED measures instead).
                if (lazy_rounds >= RP_LAZY_LEVELS && !always && !c->path_adaptive) {
                    c->lazy_penalty = std::min(64, std::max(1, c->lazy_penalty * 2));
                    c->lazy_skip = c->lazy_penalty;
                } else {
                    c->lazy_penalty = 0;
                }
            }
            else {   // the eager kernel (large batches: the bounded sweep) decides (profiles are in place); the next plans do not try again for a while
                c->lazy_penalty = std::min(64, std::max(1, c->lazy_penalty * 2));
                c->lazy_skip = c->lazy_penalty;
                c->last_lazy = 2;
                seq = ++c->seq;
                if (ticket) hrb_host->seq = 0;
                fin_seq = (ticket && !winner_pass) ? seq : 0ull;
                if (sweep_fallback && (rc = launch_sweep(false)) != RP_OK) return rc;
            }
        } else if (sweep_fallback && !always) {
            // (the back-off after an exhausted stage: the plans it keeps away from the lists)
            if ((rc = launch_sweep(true)) != RP_OK) return rc;
        }
        if (count > 0 && !lazy_done && !swept && (rc = launch_main_eval(ka, mat)) != RP_OK) return rc;
        n_partials = count == 0 ? 0 : main_grid;
    } else {
        n_partials = (int)std::max<int64_t>(1, std::min<int64_t>((count + RP_BLOCK - 1) / RP_BLOCK, c->cap_partials));
        if (count > 0)
            hipLaunchKernelGGL(rp_partials_kernel, dim3(n_partials), dim3(RP_BLOCK), 0, c->stream, ka.status, ka.cost, count,
                               ka.cand_begin, c->d_partials, ka.partials_cap);
        else
            n_partials = 0;
    }
    if (!lazy_done) {
        int fin_first = 0;
        const int fold_threshold = c->opt.fold_threshold;
        if (n_partials > fold_threshold) {   // one partial per workgroup of a very large batch: fold before the epilogue
            hipLaunchKernelGGL(rp_fold_partials_kernel, dim3(kFoldPartials), dim3(64), 0, c->stream, c->d_partials, ka.partials_cap, n_partials,
                               c->cap_partials);
            fin_first = c->cap_partials;
            n_partials = kFoldPartials;
        }
        const auto tf0 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        FinArgs fa;
        std::memset(&fa, 0, sizeof(fa));   // (list / lazy stay null: not a round of the cost-ordered stage)
        fa.status = ka.status; fa.cost = ka.cost; fa.states = ka.states; fa.partials = c->d_partials;
        fa.partials_cap = ka.partials_cap; fa.partials_first = fin_first;
        fa.dev_out = drb; fa.host_out = hrb_dev;
        fa.count = ka.count; fa.cand_begin = ka.cand_begin; fa.seq = fin_seq;
        fa.debug = c->d_debug;
        fa.N = ka.N; fa.n_partials = n_partials; fa.copy_states = copy_states ? 1 : 0;
        fa.row_stride = ka.row_stride; fa.tail_split = ka.tail_split; fa.inv_n = (uint32_t)(0x100000000ull / (unsigned long long)n) + 1u;
        fa.scratch = c->d_sel_scratch;
        fa.gate = const_cast<unsigned long long *>(ka.gate); fa.gate_seq = ka.gate_seq; fa.gate_level = ka.gate_level; fa.gate_last = chain_last ? 1 : 0;
        fa.cnt_first = (swept && !swept_totals) ? swept_grid : 0; fa.n_cnt_partials = (swept && !swept_totals) ? pass1_grid : 0;
        fa.lazy = (swept && swept_totals) ? c->d_lazy_ctl : nullptr;
        if (small && !swept) {   // one workgroup does it all (count of the colliding candidates before the winner included)
            fa.count_inline = 1;
            launch_block(c, (const void *)rp_finalize_kernel, 1, RP_FIN_THREADS, 0, &fa, sizeof(fa));
        } else {       // many workgroups, the last one to arrive writes the result and the ticket (rp_kernels.h: rp_select_kernel)
            fa.count_inline = (skip_eval || collision_level(ka) > 0) ? 1 : 0;   // (no collision test: nothing to count)
            const int sgrid = (int)std::max<int64_t>(16, std::min<int64_t>((count + RP_SEL_SLICE - 1) / RP_SEL_SLICE, (int64_t)c->num_cus));
            launch_block(c, (const void *)rp_select_kernel, sgrid, RP_SEL_THREADS, 0, &fa, sizeof(fa));
        }
        if (c->timing) c->t_sum[5] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tf0).count();
    }
    if (winner_pass && !lazy_done) {
        // nothing was materialised: re-evaluate the winner with its state block written to the host mirror
        KArgs kw = ka;
        kw.single_index = &drb->r.best_index;
        kw.count = 1;
        kw.status = &drb->pad_;   // (not w_status: its RP_WSTATUS_ROWS_ON_HOST flag tells the device-side exchange where the rows are)
        kw.cost = &drb->w_cost;
        kw.states = reinterpret_cast<double *>(hrb_dev + 1);
        kw.row_stride = n;   // the block behind the result header is compact
        kw.tail_split = 0;
        kw.coeffs = nullptr;
        kw.partials = nullptr;
        kw.host_seq = ticket ? &hrb_dev->seq : nullptr;
        kw.seq_value = seq;
        kw.ticket_if_none = (!ka.gate || chain_last) ? 1 : 0;
        // the winner IS collision-free (the selection has seen to that) and its label and cost are not read from this pass: the variant
        // without the collision query writes the same rows, reads eleven profile rows instead of sixteen and no obstacle table
        // (one workgroup, a chain of round trips: 11.1 -> 7.4 us on cfg3, behind every production-mode plan of the two-kernel path)
        if (c->opt.winner_skip_query) { kw.flags |= RP_FLAG_SKIP_COLLISION; kw.has_obstacles = 0; }   // (has_obstacles: no broad phase in a single-launch prologue either)
        if (fused_lds) launch_eval_fused(c, kw, 1, true, cin, fused_lds, G);
        // (two-kernel path: one candidate is one workgroup whatever the lanes -- a whole wavefront per candidate makes its chain
        //  of step blocks four times shorter than the batch's 16 lanes: cfg3 19 -> ~10 us.  Nothing was materialised that these
        //  rows could disagree with in their last bits.)
        else launch_eval(c, kw, 1, true, cin, c->opt.winner_lanes_as_batch ? G : 64);
    }
    HIP_TRY(c, hipGetLastError());
    rp_ctx::Pending &pd = c->pending;
    pd.active = true; pd.cin = cin; pd.skip_eval = skip_eval; pd.ticket = ticket; pd.done = lazy_done; pd.time_valid = time_valid;
    pd.seq = seq; pd.grid = grid; pd.ka = ka; pd.tp0 = tp0;
    pd.tp1 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    return RP_OK;
}

// Second half: wait for the completion ticket of the plan whose kernels pipeline_begin put on the stream, unpack the result.
int pipeline_wait(rp_ctx *c, rp_result *result, double *best_states) {
    rp_ctx::Pending &pd = c->pending;
    if (!pd.active) return fail(c, RP_ESTATE, "no plan in flight on this context");
    pd.active = false;
    const KArgs &ka = pd.ka;
    const bool cin = pd.cin, skip_eval = pd.skip_eval, ticket = pd.ticket, time_valid = pd.time_valid;
    const unsigned long long seq = pd.seq;
    const int n = ka.N + 1, grid = pd.grid;
    (void)grid;
    ResultBlock *hrb_host = reinterpret_cast<ResultBlock *>(c->h_result);
    const auto tp0 = pd.tp0, tp1 = pd.tp1;
    const bool done = pd.done;
    if (!done) {   // (with a ticket: the result block arrives ahead of the driver's completion signal)
        const int wrc = wait_plan(c, ticket, seq);
        if (wrc != RP_OK) return wrc;
    }
    (void)hrb_host;
    const auto tp2 = c->timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();

    c->epilogue_dirty = false;   // the chain ran through: its epilogues have left their scratch words at zero
    const ResultBlock *hrb = reinterpret_cast<const ResultBlock *>(c->h_result);
    result_from_core(result, hrb->r);
    result->n_collision_before_best = (int64_t)hrb->n_before;
    host_winner_coeffs(c, ka, cin, result);
    c->last_best_cost = result->best_cost; c->last_best_index = result->best_index;
    if (result->best_index >= 0 && best_states)
        std::memcpy(best_states, c->h_result + sizeof(ResultBlock), sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n);
#ifdef RP_STAMPS
    if (c->opt.print_stamps) {
        unsigned long long st[32];
        HIP_TRY(c, hipMemcpy(st, c->d_debug, sizeof(st), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "stamps (cycles since kernel start, batch kernel, one block):");
        for (int k = 1; k < 24; ++k) std::fprintf(stderr, " [%d]%lld", k, (long long)(st[k] - st[0]));
#ifdef RP_WALK_COUNT
        {
            unsigned long long wc[4];
            HIP_TRY(c, hipMemcpy(wc, c->d_debug + 40, sizeof(wc), hipMemcpyDeviceToHost));
            unsigned long long xs[4], s0;
            HIP_TRY(c, hipMemcpy(xs, c->d_debug + 44, sizeof(xs), hipMemcpyDeviceToHost));
            HIP_TRY(c, hipMemcpy(&s0, c->d_debug, sizeof(s0), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "\ncollision block: [44]%lld [45]%lld [46]%lld", (long long)(xs[0] - s0), (long long)(xs[1] - s0), (long long)(xs[2] - s0));
            unsigned long long ws[3];
            HIP_TRY(c, hipMemcpy(ws, c->d_debug + 48, sizeof(ws), hipMemcpyDeviceToHost));
            std::fprintf(stderr, " | in pose_collides: masks or-ed %lld, static walk done %lld, dynamic part reached %lld", (long long)(ws[0] - s0), (long long)(ws[1] - s0), (long long)(ws[2] - s0));
            std::fprintf(stderr, "\nstatic walk of that wavefront (accumulated over the launches so far): clusters %llu, queries %llu, exact member tests %llu", wc[0], wc[1], wc[2]);
        }
#endif
#ifndef RP_TIMELINE
        {   // rp_lon_kernel, one wavefront: kernarg lines | table copy issued | tables arrived | pair inputs | lon_step | masks | rows stored | header
            unsigned long long ls[9];
            HIP_TRY(c, hipMemcpy(ls, c->d_debug + 32, sizeof(ls), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "\nprofile kernel (rp_lon_kernel) stamps, cycles since its start:");
            for (int k = 1; k < 9; ++k) std::fprintf(stderr, " [%d]%lld", k, (long long)(ls[k] - ls[0]));
        }
#endif
        std::fprintf(stderr, "\nfinalize stamps (cycles since its start):");
        for (int k = 25; k < 31; ++k) std::fprintf(stderr, " [%d]%lld", k, (long long)(st[k] - st[24]));
        std::fprintf(stderr, "\n");
    }
#endif
#ifdef RP_TIMELINE
    if (c->opt.print_stamps) {
        std::vector<unsigned long long> tl(2 * 4096);
        HIP_TRY(c, hipMemcpy(tl.data(), c->d_debug + 32, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        const int nb = std::min(grid, 4096);
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, tl[2 * b]); t1 = std::max(t1, tl[2 * b + 1]); }
        std::vector<unsigned long long> st(nb), en(nb), du(nb);
        for (int b = 0; b < nb; ++b) { st[b] = tl[2 * b] - t0; en[b] = tl[2 * b + 1] - t0; du[b] = tl[2 * b + 1] - tl[2 * b]; }
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
        auto q = [&](const std::vector<unsigned long long> &v, double f) { return 0.01 * (double)v[(size_t)(f * (nb - 1))]; };
        std::fprintf(stderr, "timeline (us, 100 MHz clock): %d blocks, first start -> last end %.2f; starts p0/p25/p50/p55/p60/p75/p100 "
                     "%.2f %.2f %.2f %.2f %.2f %.2f %.2f; ends p0/p50/p100 %.2f %.2f %.2f; durations p0/p50/p100 %.2f %.2f %.2f\n",
                     grid, 0.01 * (double)(t1 - t0), q(st, 0), q(st, .25), q(st, .5), q(st, .55), q(st, .6), q(st, .75), q(st, 1), q(en, 0), q(en, .5),
                     q(en, 1), q(du, 0), q(du, .5), q(du, 1));
    }
#endif
    if (c->timing && !skip_eval) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        c->t_sum[0] += us(c->t_entry, tp0); c->t_sum[1] += us(tp0, tp1); c->t_sum[2] += us(tp1, tp2); c->t_sum[3] += us(tp2, tp3);
        c->t_calls++;
        if (c->t_calls == 20 && !c->t_warm) {   // the first calls carry one-time costs (module load, function lookup): start over
            c->t_warm = true;
            c->t_calls = 0;
            for (double &v : c->t_sum) v = 0.0;
        }
    }
    result->kernel_ms = 0.0;
    if (time_valid) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        if (e == hipErrorNotReady) {   // ticket arrived ahead of the driver's bookkeeping
            HIP_TRY(c, hipEventSynchronize(c->ev1));
            e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        }
        HIP_TRY(c, e);
        result->kernel_ms = ms;
    }
    return RP_OK;
}

// a whole call's duration -> the running mean of the path it took (see rp_ctx::path_us)
void note_path_time(rp_ctx *c) {
    if (!c->time_whole) return;
    c->time_whole = false;
    if (!c->path_adaptive) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c->plan_t0).count();
    if (c->last_lazy == 2) return;   // (an exhausted stage is the back-off's business -- lazy_skip --, not a sample of what the stage costs when it works)
    const int k = c->last_lazy == RP_PATH_EAGER ? 0 : 1;
    c->path_us[k] = c->path_n[k] == 0 ? us : c->path_us[k] + 0.25 * (us - c->path_us[k]);
    if (c->path_n[k] < (1 << 30)) ++c->path_n[k];
}

int run_pipeline(rp_ctx *c, KArgs &ka, bool mat, bool cin, bool skip_eval, rp_result *result, double *best_states) {
    c->time_whole = true;
    int rc = pipeline_begin(c, ka, mat, cin, skip_eval, best_states != nullptr);
    if (rc == RP_OK) rc = pipeline_wait(c, result, best_states);
    if (rc == RP_OK) note_path_time(c);
    c->time_whole = false;
    return rc;
}

}  // namespace

// -------------------------------------------------------------------------------------------------
extern "C" {

int rp_abi_version(void) { return RP_ABI_VERSION; }

#ifndef RP_SRC_HASH
#define RP_SRC_HASH "unknown"
#endif
const char *rp_source_hash(void) { return RP_SRC_HASH; }

int rp_create(rp_ctx **out, int device) {
    if (!out) return RP_EINVAL;
    *out = nullptr;
    rp_ctx *c = new (std::nothrow) rp_ctx();
    if (!c) return RP_ENOMEM;
    *out = c;   // returned even on failure so that rp_last_error works; caller destroys it
    c->device = device;
    int ndev = 0;
    HIP_TRY(c, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(c, RP_EINVAL, "no such HIP device");
    HIP_TRY(c, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreate(&c->ev0));
    HIP_TRY(c, hipEventCreate(&c->ev1));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done, hipEventBlockingSync | hipEventDisableTiming));
    HIP_TRY(c, hipMalloc((void **)&c->d_single, sizeof(int64_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_pair_hdr_one, sizeof(PairHdr)));
    HIP_TRY(c, hipMalloc((void **)&c->d_sel_scratch, RP_SEL_SCRATCH * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemset(c->d_sel_scratch, 0, RP_SEL_SCRATCH * sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc((void **)&c->d_lazy_ctl, sizeof(LazyCtl)));
    HIP_TRY(c, hipMemset(c->d_lazy_ctl, 0, sizeof(LazyCtl)));
    HIP_TRY(c, hipMalloc((void **)&c->d_lazy_lists, kLazyListWords * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_sweep, sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc((void **)&c->d_lazy_hist, (RP_LAZY_BINS + 1) * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(c->d_lazy_hist, 0, (RP_LAZY_BINS + 1) * sizeof(uint32_t)));
#if defined(RP_STAMPS) || defined(RP_TIMELINE)
    HIP_TRY(c, hipMalloc((void **)&c->d_debug, (32 + 2 * 4096) * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemset(c->d_debug, 0, (32 + 2 * 4096) * sizeof(unsigned long long)));
#endif
    HIP_TRY(c, hipHostMalloc((void **)&c->h_single, sizeof(int64_t), hipHostMallocDefault));
    options_from_environment(c->opt);   // (the only place the library reads RP_AMD_* switches of the plan path: defaults of this context)
    c->timing = c->opt.timing != 0;
    c->spin_wait = c->opt.wait_mode != RP_WAIT_EVENT;
    return RP_OK;
}

void rp_destroy(rp_ctx *c) {
    if (!c) return;
    if (c->timing && c->t_calls)
        std::fprintf(stderr, "rp_plan host phases over %llu calls after 20 warm-up calls (us): entry->launch %.2f | launches %.2f (evaluation %.2f, epilogue %.2f) | wait %.2f | unpack %.2f\n",
                     c->t_calls, c->t_sum[0] / c->t_calls, c->t_sum[1] / c->t_calls, c->t_sum[4] / c->t_calls, c->t_sum[5] / c->t_calls,
                     c->t_sum[2] / c->t_calls, c->t_sum[3] / c->t_calls);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_tables, c->d_sobb, c->d_tri, c->d_circ, c->d_dyn, c->d_clus, c->d_slot, c->d_grid, c->d_clus_info, c->d_stage, c->d_status, c->d_cost, c->d_user,
                   c->d_states, c->d_compact, c->d_partials, c->d_result, c->d_single, c->d_profile, c->d_profile_one,
                   c->d_pair_hdr, c->d_pair_hdr_one, c->d_sel_scratch, c->d_lazy_ctl, c->d_lazy_lists, c->d_lazy_hist, c->d_lazy_states, c->d_gate, c->d_sweep};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_arena) (void)hipHostFree(c->h_arena);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->h_single) (void)hipHostFree(c->h_single);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *rp_last_error(const rp_ctx *c) { return c ? c->err.c_str() : "null context"; }

int rp_last_path(const rp_ctx *c) { return c ? c->last_lazy : 0; }
int rp_last_kernel(const rp_ctx *c) { return c ? c->last_kernel : 0; }

int rp_set_collision_path(rp_ctx *c, int mode) {
    if (!c) return RP_EINVAL;
    if (mode < RP_COLLISION_AUTO || mode > RP_COLLISION_TIMED) return fail(c, RP_EINVAL, "rp_set_collision_path: unknown mode");
    if (mode != c->collision_mode) { c->lazy_skip = c->lazy_penalty = 0; c->path_regime = -1; }
    c->collision_mode = mode;
    return RP_OK;
}

int rp_set_profiling(rp_ctx *c, int enable) {
    if (!c) return RP_EINVAL;
    c->profiling = enable < 0 ? 0 : enable;   // k: every k-th call is timed (1 = every call)
    return RP_OK;
}

int rp_set_option(rp_ctx *c, const char *key, int64_t value) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_set_option: a plan is in flight on this context (rp_plan_wait first)");
    const OptionDesc *d = find_option(key);
    if (!d) return fail(c, RP_EINVAL, std::string("rp_set_option: unknown option '") + (key ? key : "(null)") + "'");
    if (value < d->lo || value > d->hi) return fail(c, RP_EINVAL, std::string("rp_set_option: value out of range for '") + key + "'");
    c->opt.*(d->field) = (int)value;
    c->timing = c->opt.timing != 0;
    c->spin_wait = c->opt.wait_mode != RP_WAIT_EVENT;
    // (a change of the launch policy starts the collision-path rules over: what the earlier plans learnt belongs to the other policy)
    c->lazy_skip = c->lazy_penalty = 0; c->path_regime = -1;
    return RP_OK;
}

int rp_get_option(const rp_ctx *c, const char *key, int64_t *value) {
    if (!c || !value) return RP_EINVAL;
    if (key && std::strcmp(key, "wait_fallbacks") == 0) { *value = c->wait_fallbacks; return RP_OK; }   // (read-only counter)
    const OptionDesc *d = find_option(key);
    if (!d) return RP_EINVAL;
    *value = c->opt.*(d->field);
    return RP_OK;
}

int rp_set_wait_mode(rp_ctx *c, int mode) { return rp_set_option(c, "wait_mode", mode); }
int rp_get_wait_mode(const rp_ctx *c) { return c ? c->opt.wait_mode : RP_WAIT_SPIN; }

int rp_set_reference(rp_ctx *c, int32_t n, const double *ref_pos, const double *ref_theta, const double *ref_curv,
                     const double *ref_curv_d, const double *ref_x, const double *ref_y, double proj_domain_d_limit) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_set_reference: a plan is in flight on this context (rp_plan_wait first)");
    if (n < 2 || !ref_pos || !ref_theta || !ref_curv || !ref_curv_d || !ref_x || !ref_y)
        return fail(c, RP_EINVAL, "rp_set_reference: need n >= 2 and six non-null tables");
    for (int i = 0; i + 1 < n; ++i)
        if (!(ref_pos[i + 1] > ref_pos[i])) return fail(c, RP_EINVAL, "rp_set_reference: ref_pos must be strictly increasing");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<double> t((size_t)TB_ROWS * (size_t)n);
    std::memcpy(&t[(size_t)TB_POS * n], ref_pos, sizeof(double) * n);
    std::memcpy(&t[(size_t)TB_THETA * n], ref_theta, sizeof(double) * n);
    std::memcpy(&t[(size_t)TB_CURV * n], ref_curv, sizeof(double) * n);
    std::memcpy(&t[(size_t)TB_CURV_D * n], ref_curv_d, sizeof(double) * n);
    std::memcpy(&t[(size_t)TB_X * n], ref_x, sizeof(double) * n);
    std::memcpy(&t[(size_t)TB_Y * n], ref_y, sizeof(double) * n);
    // vertex tangents: normalised sum of the adjacent unit segment directions; operation order is
    // the one of commonroad_rp_amd.coordinate_system.compute_vertex_tangents
    double *tx = &t[(size_t)TB_TX * n], *ty = &t[(size_t)TB_TY * n];
    std::vector<double> ux(n - 1), uy(n - 1);
    for (int i = 0; i + 1 < n; ++i) {
        double ex = ref_x[i + 1] - ref_x[i], ey = ref_y[i + 1] - ref_y[i];
        double ln = std::sqrt(ex * ex + ey * ey);
        if (!(ln > 0.0)) return fail(c, RP_EINVAL, "rp_set_reference: duplicate polyline vertices");
        ux[i] = ex / ln; uy[i] = ey / ln;
    }
    tx[0] = ux[0]; ty[0] = uy[0];
    tx[n - 1] = ux[n - 2]; ty[n - 1] = uy[n - 2];
    for (int i = 1; i + 1 < n; ++i) {
        double sx = ux[i - 1] + ux[i], sy = uy[i - 1] + uy[i];
        double tn = std::sqrt(sx * sx + sy * sy);
        tx[i] = sx / tn; ty[i] = sy / tn;
    }
    // reciprocal segment lengths; the last slot serves the wrap-around pair (k0 = n-1, k1 = 0)
    double *inv = &t[(size_t)TB_INVLEN * n];
    double hmin = ref_pos[1] - ref_pos[0];
    for (int i = 0; i + 1 < n; ++i) {
        inv[i] = 1.0 / (ref_pos[i + 1] - ref_pos[i]);
        hmin = std::min(hmin, ref_pos[i + 1] - ref_pos[i]);
    }
    inv[n - 1] = 1.0 / (ref_pos[0] - ref_pos[n - 1]);
    // uniform bucket table for the O(1) segment lookup (falls back to binary search if too fine)
    const double span = ref_pos[n - 1] - ref_pos[0];
    const double nbf = std::floor(span / hmin) + 1.0;
    int nb = (nbf >= 1.0 && nbf <= 8192.0) ? (int)nbf : 0;
    if (nb > 0) {
        t.resize(t.size() + (size_t)(nb + 1) / 2, 0.0);
        int32_t *bk = reinterpret_cast<int32_t *>(t.data() + (size_t)TB_ROWS * n);
        for (int b = 0; b < nb; ++b)
            bk[b] = (int32_t)(std::upper_bound(ref_pos, ref_pos + n, ref_pos[0] + b * hmin) - ref_pos);
    }
    if (t.size() & 1) t.push_back(0.0);   // staged with 16-byte loads
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = upload(c, c->d_tables, t);
    if (rc != RP_OK) return rc;
    c->n_buckets = nb;
    c->bucket_inv_h = 1.0 / hmin;
    c->h_pos.assign(ref_pos, ref_pos + n);
    c->h_front.ref.resize((size_t)n);
    for (int i = 0; i < n; ++i) c->h_front.ref[(size_t)i] = {ref_x[i], ref_y[i]};
    c->h_front.pos.assign(ref_pos, ref_pos + n); c->h_front.theta.assign(ref_theta, ref_theta + n);
    c->h_front.curv.assign(ref_curv, ref_curv + n); c->h_front.curv_d.assign(ref_curv_d, ref_curv_d + n);
    c->table_words = (int)t.size();
    c->n_ref = n;
    c->proj_d_limit = proj_domain_d_limit;
    int it = 1;
    while ((1 << it) < n + 1) ++it;
    c->search_iters = it + 1;
    c->have_last = false;
    return RP_OK;
}

// Uniform grid over the static shapes for an ego rectangle of bounding radius ego_r: bit k of a cell is set when a member of
// cluster k lies within ego_r (+ half a cell diagonal + a margin) of the cell's centre, i.e. can be touched by an ego
// rectangle whose centre lies anywhere in the cell.  Distances: exact for rectangles, through the bounding circle for
// triangles (an under-estimate: more bits, never fewer).  At most ~2^20 cells (8 MB); the cell grows with the map.
// Built on first use and again when a plan comes with a larger ego rectangle (a smaller one keeps the grid: still conservative).
static int ensure_static_grid(rp_ctx *c, double ego_r) {
    if (c->grid_valid && ego_r <= c->grid_ego_r) return RP_OK;
    const int n_clus = (int)(c->h_clus_info.size() / 4);
    c->obs.grid = nullptr; c->obs.gnx = c->obs.gny = 0;
    c->grid_valid = true; c->grid_ego_r = ego_r;
    if (n_clus == 0) return RP_OK;   // (no static shapes: every lookup says "nothing")
    if (!(ego_r == ego_r) || !(ego_r < 1e300)) { c->obs.gnx = c->obs.gny = -1; return RP_OK; }   // no grid for such an ego: every pose walks every cluster
    struct Shape { int clus, kind; double cx, cy, ux, uy, hl, hw; };   // triangles / circles: the circle (cx, cy), radius hl
    std::vector<Shape> shapes;
    double x0 = HUGE_VAL, y0 = HUGE_VAL, x1 = -HUGE_VAL, y1 = -HUGE_VAL;
    for (int k = 0; k < n_clus; ++k) {
        const int kind = c->h_clus_info[4 * k], first = c->h_clus_info[4 * k + 1], cnt = c->h_clus_info[4 * k + 2];
        for (int m = 0; m < cnt; ++m) {
            Shape sh{k, kind, 0, 0, 1, 0, 0, 0};
            double ex, ey;   // half extents of the axis-aligned box of the shape
            if (kind == 0) {
                const double *o = &c->h_sobb[(size_t)(first + m) * OB_ROW];
                sh.cx = o[OB_CX]; sh.cy = o[OB_CY]; sh.ux = o[OB_UX]; sh.uy = o[OB_UY]; sh.hl = o[OB_HL]; sh.hw = o[OB_HW];
                ex = std::fabs(sh.ux) * sh.hl + std::fabs(sh.uy) * sh.hw; ey = std::fabs(sh.uy) * sh.hl + std::fabs(sh.ux) * sh.hw;
            } else if (kind == 1) {
                const double *o = &c->h_tri[(size_t)(first + m) * 10];
                sh.cx = o[6]; sh.cy = o[7]; sh.hl = o[8]; ex = ey = o[8];
            } else {
                const double *o = &c->h_circ[(size_t)(first + m) * 4];
                sh.cx = o[0]; sh.cy = o[1]; sh.hl = o[2]; ex = ey = o[2];
            }
            if (!(sh.cx == sh.cx && sh.cy == sh.cy && ex == ex && ey == ey) || !(std::fabs(sh.cx) < 1e300 && std::fabs(sh.cy) < 1e300 && ex < 1e300 && ey < 1e300)) {
                c->obs.gnx = c->obs.gny = -1;   // a shape without finite extent: no grid, every pose walks every cluster
                return RP_OK;
            }
            sh.hw = kind == 0 ? sh.hw : 0.0;
            shapes.push_back(sh);
            x0 = std::min(x0, sh.cx - ex); x1 = std::max(x1, sh.cx + ex);
            y0 = std::min(y0, sh.cy - ey); y1 = std::max(y1, sh.cy + ey);
        }
    }
    if (shapes.empty()) return RP_OK;
    // cell size: 0.5 m unless that takes more than ~2^20 cells
    double cell = 0.5;
    for (int it = 0; it < 64; ++it) {
        const double R = ego_r + cell * 0.70710678118654757 + 1e-6;
        const double w = (x1 - x0) + 2.0 * (R + cell), h = (y1 - y0) + 2.0 * (R + cell);
        if ((w / cell) * (h / cell) <= 1048576.0) break;
        cell *= 1.25;
    }
    const double R = (ego_r + cell * 0.70710678118654757) * (1.0 + 1e-9) + 1e-6;
    const double gx0 = x0 - (R + cell), gy0 = y0 - (R + cell);
    const int nx = (int)std::ceil(((x1 + R + cell) - gx0) / cell) + 1, ny = (int)std::ceil(((y1 + R + cell) - gy0) / cell) + 1;
    std::vector<unsigned long long> cells((size_t)nx * (size_t)ny, 0ull);
    for (const Shape &sh : shapes) {
        const double ex = sh.kind == 0 ? std::fabs(sh.ux) * sh.hl + std::fabs(sh.uy) * sh.hw : sh.hl;
        const double ey = sh.kind == 0 ? std::fabs(sh.uy) * sh.hl + std::fabs(sh.ux) * sh.hw : sh.hl;
        const int ix0 = std::max(0, (int)std::floor((sh.cx - ex - R - gx0) / cell) - 1), ix1 = std::min(nx - 1, (int)std::floor((sh.cx + ex + R - gx0) / cell) + 1);
        const int iy0 = std::max(0, (int)std::floor((sh.cy - ey - R - gy0) / cell) - 1), iy1 = std::min(ny - 1, (int)std::floor((sh.cy + ey + R - gy0) / cell) + 1);
        const unsigned long long bit = 1ull << sh.clus;
        for (int iy = iy0; iy <= iy1; ++iy) {
            const double py = gy0 + ((double)iy + 0.5) * cell - sh.cy;
            for (int ix = ix0; ix <= ix1; ++ix) {
                const double px = gx0 + ((double)ix + 0.5) * cell - sh.cx;
                double dist;
                if (sh.kind == 0) {
                    const double lx = std::fabs(px * sh.ux + py * sh.uy) - sh.hl, ly = std::fabs(py * sh.ux - px * sh.uy) - sh.hw;
                    dist = std::hypot(std::max(lx, 0.0), std::max(ly, 0.0));
                } else {
                    dist = std::max(0.0, std::hypot(px, py) - sh.hl);
                }
                if (dist <= R) cells[(size_t)iy * nx + ix] |= bit;
            }
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_grid) { HIP_TRY(c, hipFree(c->d_grid)); c->d_grid = nullptr; }
    HIP_TRY(c, hipMalloc((void **)&c->d_grid, cells.size() * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemcpy(c->d_grid, cells.data(), cells.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    c->obs.grid = c->d_grid; c->obs.gx0 = gx0; c->obs.gy0 = gy0; c->obs.ginv = 1.0 / cell; c->obs.gnx = nx; c->obs.gny = ny;
    return RP_OK;
}

int rp_set_obstacles(rp_ctx *c, int32_t n_sobb, const double *sobb, int32_t n_tri, const double *tri, int32_t n_circ,
                     const double *circ, int32_t n_dyn, int32_t n_steps, int32_t dyn_t0, const double *dyn) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_set_obstacles: a plan is in flight on this context (rp_plan_wait first)");
    if (n_sobb < 0 || n_tri < 0 || n_circ < 0 || n_dyn < 0 || n_steps < 0 || (n_sobb && !sobb) || (n_tri && !tri) ||
        (n_circ && !circ) || (n_dyn && n_steps && !dyn))
        return fail(c, RP_EINVAL, "rp_set_obstacles: negative count or null table");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<double> a((size_t)n_sobb * OB_ROW), b((size_t)n_tri * 10), d((size_t)n_circ * 4),
        e(dyn_table_doubles(n_dyn, n_steps));
    for (int j = 0; j < n_sobb; ++j) {
        const double *o = sobb + 5 * j;
        double *r = &a[(size_t)j * OB_ROW];
        r[OB_CX] = o[0]; r[OB_CY] = o[1]; r[OB_UX] = std::cos(o[2]); r[OB_UY] = std::sin(o[2]);
        r[OB_HL] = o[3]; r[OB_HW] = o[4]; r[OB_R] = std::sqrt(o[3] * o[3] + o[4] * o[4]); r[OB_PAD] = 0.0;
    }
    for (int j = 0; j < n_tri; ++j) {
        const double *o = tri + 6 * j;
        double *r = &b[(size_t)j * 10];
        for (int k = 0; k < 6; ++k) r[k] = o[k];
        double bx = (o[0] + o[2] + o[4]) / 3.0, by = (o[1] + o[3] + o[5]) / 3.0, rr = 0.0;
        for (int k = 0; k < 3; ++k) rr = std::max(rr, std::hypot(o[2 * k] - bx, o[2 * k + 1] - by));
        r[6] = bx; r[7] = by; r[8] = rr; r[9] = 0.0;
    }
    for (int j = 0; j < n_circ; ++j) {
        const double *o = circ + 3 * j;
        double *r = &d[(size_t)j * 4];
        r[0] = o[0]; r[1] = o[1]; r[2] = o[2]; r[3] = 0.0;
    }
    const size_t plane = (size_t)n_dyn * (size_t)n_steps;
    for (int j = 0; j < n_dyn; ++j)
        for (int k = 0; k < n_steps; ++k) {
            const double *o = dyn + ((size_t)j * n_steps + k) * 5;
            const size_t at = (size_t)j * n_steps + k;
            e[at] = o[0]; e[plane + at] = o[1];
            e[2 * plane + at] = std::cos(o[2]); e[3 * plane + at] = std::sin(o[2]);
            e[4 * plane + at] = o[3]; e[5 * plane + at] = o[4];
            e[6 * plane + at] = std::sqrt(o[3] * o[3] + o[4] * o[4]);
        }
    // the broad phase of rp_lon_kernel reads a circle per (obstacle, step): centre as one 16-byte pair, radius as the obstacle's
    // largest over its steps, a scalar -- one vector load per test instead of three (rp_kernels.h: near_mask_step)
    double rmax_all = 0.0;
    {
        double *xy = &e[dyn_xy_offset(n_dyn, n_steps)], *rmax = &e[dyn_rmax_offset(n_dyn, n_steps)];
        for (int j = 0; j < n_dyn; ++j) {
            double r = 0.0;
            for (int k = 0; k < n_steps; ++k) {
                const size_t at = (size_t)j * n_steps + k;
                xy[2 * at] = e[at]; xy[2 * at + 1] = e[plane + at];
                if (e[at] == e[at] && e[6 * plane + at] > r) r = e[6 * plane + at];   // (steps where the obstacle is absent do not count)
            }
            rmax[j] = r;
            rmax_all = std::max(rmax_all, r);
        }
    }
    // clusters of consecutive static shapes of one kind (at most 63: one bit each in the (pair, step) masks)
    std::vector<double> cl;
    std::vector<int32_t> ci;
    {
        const int n_static = n_sobb + n_tri + n_circ;
        const int per = std::max(1, (n_static + 59) / 60);
        auto add = [&](int kind, int count, auto centre) {
            for (int first = 0; first < count; first += per) {
                const int cnt = std::min(per, count - first);
                double mx = 0.0, my = 0.0;
                for (int q = 0; q < cnt; ++q) { double x, y, r; centre(first + q, x, y, r); mx += x; my += y; }
                mx /= cnt; my /= cnt;
                double rad = 0.0;
                for (int q = 0; q < cnt; ++q) { double x, y, r; centre(first + q, x, y, r); rad = std::max(rad, std::hypot(x - mx, y - my) + r); }
                cl.insert(cl.end(), {mx, my, rad * (1.0 + 1e-12) + 1e-12, 0.0});
                ci.insert(ci.end(), {kind, first, cnt, 0});
            }
        };
        add(0, n_sobb, [&](int j, double &x, double &y, double &r) { x = a[(size_t)j * OB_ROW + OB_CX]; y = a[(size_t)j * OB_ROW + OB_CY]; r = a[(size_t)j * OB_ROW + OB_R]; });
        add(1, n_tri, [&](int j, double &x, double &y, double &r) { x = b[(size_t)j * 10 + 6]; y = b[(size_t)j * 10 + 7]; r = b[(size_t)j * 10 + 8]; });
        add(2, n_circ, [&](int j, double &x, double &y, double &r) { x = d[(size_t)j * 4]; y = d[(size_t)j * 4 + 1]; r = d[(size_t)j * 4 + 2]; });
    }
    const int n_clus = (int)(ci.size() / 4);
    const int clus_per = std::max(1, (n_sobb + n_tri + n_circ + 59) / 60);
    if (n_clus > 63) return fail(c, RP_EINVAL, "rp_set_obstacles: internal error, more than 63 static clusters");
    // member slots of the clusters as one flat table (rp_device.h: ObsTables::slot)
    std::vector<double> sl((size_t)n_clus * clus_per * RP_SLOT_ROW, 0.0);
    for (int cidx = 0; cidx < n_clus; ++cidx) {
        const int kind = ci[4 * cidx], first = ci[4 * cidx + 1], cnt = ci[4 * cidx + 2];
        unsigned long long bit = 1ull << cidx;
        double bit_d;
        std::memcpy(&bit_d, &bit, sizeof(bit_d));
        for (int m = 0; m < clus_per; ++m) {
            double *r = &sl[((size_t)cidx * clus_per + m) * RP_SLOT_ROW];
            r[6] = bit_d;
            if (m >= cnt) { r[2] = 1.0; r[4] = r[5] = -HUGE_VAL; continue; }
            const int j = first + m;
            r[7] = (double)(kind + 4 * j);   // what the narrow phase needs to find the shape itself
            if (kind == 0) {
                for (int q = 0; q < 6; ++q) r[q] = a[(size_t)j * OB_ROW + q];
            } else {
                const double bx = kind == 1 ? b[(size_t)j * 10 + 6] : d[(size_t)j * 4], by = kind == 1 ? b[(size_t)j * 10 + 7] : d[(size_t)j * 4 + 1];
                const double rr = kind == 1 ? b[(size_t)j * 10 + 8] : d[(size_t)j * 4 + 2];
                r[0] = bx; r[1] = by; r[2] = 1.0; r[3] = 0.0; r[4] = r[5] = rr;
            }
        }
    }
    int rc;
    if ((rc = upload(c, c->d_slot, sl)) != RP_OK) return rc;
    if ((rc = upload(c, c->d_clus, cl)) != RP_OK) return rc;
    if (c->d_clus_info) { HIP_TRY(c, hipFree(c->d_clus_info)); c->d_clus_info = nullptr; }
    if (n_clus) {
        HIP_TRY(c, hipMalloc((void **)&c->d_clus_info, ci.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpy(c->d_clus_info, ci.data(), ci.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if ((rc = upload(c, c->d_sobb, a)) != RP_OK) return rc;
    if ((rc = upload(c, c->d_tri, b)) != RP_OK) return rc;
    if ((rc = upload(c, c->d_circ, d)) != RP_OK) return rc;
    if ((rc = upload(c, c->d_dyn, e)) != RP_OK) return rc;
    c->obs.sobb = c->d_sobb; c->obs.tri = c->d_tri; c->obs.circ = c->d_circ; c->obs.dyn = c->d_dyn;
    c->obs.n_sobb = n_sobb; c->obs.n_tri = n_tri; c->obs.n_circ = n_circ;
    c->obs.n_dyn = n_dyn; c->obs.n_steps = n_steps; c->obs.dyn_t0 = dyn_t0; c->obs.dyn_rmax_all = rmax_all;
    c->obs.clus = c->d_clus; c->obs.clus_info = c->d_clus_info; c->obs.n_clus = n_clus; c->obs.clus_per = clus_per;
    c->obs.slot = c->d_slot;
    // (a replanning loop that brings new predictions of the dynamic obstacles every cycle keeps its static shapes: so does the grid)
    const bool same_static = c->grid_valid && a == c->h_sobb && b == c->h_tri && d == c->h_circ && ci == c->h_clus_info;
    if (!same_static) {
        c->h_sobb = a; c->h_tri = b; c->h_circ = d; c->h_clus_info = ci;
        c->grid_valid = false;
        c->obs.grid = nullptr; c->obs.gnx = c->obs.gny = 0; c->obs.gx0 = c->obs.gy0 = 0.0; c->obs.ginv = 0.0;
    }
    c->have_last = false;
    return RP_OK;
}

}  // extern "C"

namespace {
struct LevelGate {   // a level of a chain (rp_plan_levels): see KArgs::gate
    const unsigned long long *gate;
    unsigned long long seq;
    int level;     // 1, 2, ...
    bool last;     // last level of the chain: reports whatever it finds
};
int plan_begin_impl(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_grids *g, int64_t cand_begin, int64_t cand_end,
                    int32_t want_best_states, const LevelGate *lg);
int plan_begin(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_grids *g, int64_t cand_begin, int64_t cand_end, int32_t want_best_states);
int plan_levels_begin(rp_ctx *c, const rp_params *p, const rp_cost *cost, int32_t n_levels, const rp_grids *grids, int32_t want_best_states);
// grids of a RP_PLAN_PACKED call: the sizes in `dims`, the arrays in the context's buffer (rp_fast_buffer), level behind level
int packed_grids(rp_ctx *c, int32_t n_levels, const rp_grids *dims, rp_grids *out);
}  // namespace

extern "C" {

int rp_plan(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_grids *g, int64_t cand_begin, int64_t cand_end, uint32_t flags,
            rp_result *result, double *out) {
    if (!c) return RP_EINVAL;
    if (flags & ~(RP_PLAN_BEGIN | RP_PLAN_ROWS | RP_PLAN_PACKED)) return fail(c, RP_EINVAL, "rp_plan: unknown flag");
    rp_grids pg;
    if (flags & RP_PLAN_PACKED) {
        if (!p || !cost || !g) return fail(c, RP_EINVAL, "rp_plan: null argument");
        if (cand_begin != 0 || cand_end >= 0) return fail(c, RP_EINVAL, "rp_plan: RP_PLAN_PACKED plans whole grids (cand_begin 0, cand_end -1)");
        if (!(flags & RP_PLAN_BEGIN) && !out) return fail(c, RP_EINVAL, "rp_plan: RP_PLAN_PACKED needs the output block");
        const int grc = packed_grids(c, 1, g, &pg);
        if (grc != RP_OK) return grc;
        g = &pg;
    }
    if (flags & RP_PLAN_BEGIN) return plan_begin(c, p, cost, g, cand_begin, cand_end, ((flags & RP_PLAN_ROWS) || out) ? 1 : 0);
    if (!result) return fail(c, RP_EINVAL, "null params / cost / result");
    if (result->struct_size != sizeof(rp_result)) return fail(c, RP_EABI, "rp_plan: rp_result.struct_size is not this library's");
    c->time_whole = true;
    int rc = plan_begin(c, p, cost, g, cand_begin, cand_end, out != nullptr ? 1 : 0);
    if (rc == RP_OK) rc = rp_plan_wait(c, result, out);
    if (rc == RP_OK) note_path_time(c);
    c->time_whole = false;
    if (rc != RP_OK || !(flags & RP_PLAN_PACKED) || result->best_index < 0) return rc;
    const int n = p->N + 1;
    return rp_pack_trajectory(n, out, p->dt, p->wheelbase, p->x0_orientation, out + (size_t)RP_N_ARRAYS * n);
}

int rp_plan_wait(rp_ctx *c, rp_result *result, double *best_states) {
    if (!c) return RP_EINVAL;
    if (!result) return fail(c, RP_EINVAL, "rp_plan_wait: null result");
    if (result->struct_size != sizeof(rp_result)) return fail(c, RP_EABI, "rp_plan_wait: rp_result.struct_size is not this library's");
    if (!c->pending.active) return fail(c, RP_ESTATE, "rp_plan_wait: no plan in flight on this context");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->chain.size() > 1) {
        // a chain of levels: one ticket; the result block names the level it belongs to, whose launch state becomes the context's
        int rc = wait_plan(c, c->chain.front().pending.ticket, c->chain.front().pending.seq);   // (RP_WAIT_EVENT: no ticket was asked for)
        if (rc != RP_OK) { c->pending.active = false; c->chain.clear(); return rc; }
        const int tag = (int)reinterpret_cast<const ResultBlock *>(c->h_result)->pad_;
        if (tag < 1 || tag > (int)c->chain.size()) { c->pending.active = false; c->chain.clear(); return fail(c, RP_EHIP, "rp_plan_wait: result block of a level chain without its level"); }
        rp_ctx::LevelState &st = c->chain[(size_t)tag - 1];
        c->pending = st.pending;
        c->pending.done = true;   // (the ticket has arrived)
        c->staged = st.staged; c->staged_on_device = false; c->last_rows_on_device = st.last_rows_on_device;
        c->last_G = st.last_G; c->last_block = st.last_block; c->last_fused_lds = st.last_fused_lds;
        c->last_kernel = st.last_kernel; c->last_lazy = st.last_lazy;   // (rp_last_kernel / rp_last_path: of the level the result belongs to)
        // the grids of that level are what later launches on this context (winner re-evaluation, rp_eval_one) find in their launch blocks
        if (!c->staged.empty() && c->staged.size() <= sizeof(c->kargs_g.grid)) {
            std::memcpy(c->kargs_g.grid, c->staged.data(), c->staged.size());
            std::memcpy(c->kargs_gl.grid, c->staged.data(), c->staged.size());
        }
        c->last_level = st.grid_index;
    } else if (c->chain.size() == 1) {
        c->last_level = c->chain[0].grid_index;
    }
    int rc = pipeline_wait(c, result, best_states);
    if (rc != RP_OK) { c->chain.clear(); return rc; }
    c->last = c->pending.ka; c->have_last = true; c->last_mat = c->pending.mat; c->last_coeffs = c->pending.coeffs;
    if (!c->chain.empty()) {
        // levels behind the chain (too large to ride in it): one by one, while there is no winner (reactive_planner.py:616-636)
        int next = c->chain.back().grid_index + 1;
        const int total = c->chain_total;
        c->chain.clear();
        while (result->best_index < 0 && next < total) {
            const rp_ctx::ChainInputs &in = c->chain_in;
            const int64_t cnt = (int64_t)in.grids[next].nT * in.grids[next].nL * in.grids[next].nD;
            if (cnt > 0) {
                rc = plan_begin_impl(c, &in.p, &in.cost, &in.grids[next], 0, -1, in.want, nullptr);
                if (rc == RP_OK) rc = pipeline_wait(c, result, best_states);
                if (rc != RP_OK) return rc;
                c->last = c->pending.ka; c->have_last = true; c->last_mat = c->pending.mat; c->last_coeffs = c->pending.coeffs;
                c->last_level = next;
            }
            ++next;
        }
    }
    // The loop of plan() visits an empty level like any other (reactive_planner.py:616-636: an empty bundle has no winner and leaves
    // its counters at zero): when no level delivered and the LAST one is empty, the result is that level's -- nothing evaluated.
    if (c->chain_total > 0 && result->best_index < 0) {
        const int last = c->chain_total - 1;
        const rp_grids &gl = c->chain_in.grids[(size_t)last];
        if ((int64_t)gl.nT * gl.nL * gl.nD == 0 && c->last_level < last) {
            result->n_candidates = result->n_feasible = result->n_collision = result->n_collision_before_best = 0;
            for (int64_t &r : result->reason_counts) r = 0;
            c->last_level = last;
            c->have_last = false;   // (there is no "last plan" to fetch labels of)
        }
    }
    c->chain_total = 0;
    return RP_OK;
}

}  // extern "C"

namespace {
int plan_begin(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_grids *g, int64_t cand_begin, int64_t cand_end, int32_t want_best_states) {
    if (c) { c->chain.clear(); c->chain_total = 0; c->last_level = 0; }
    return plan_begin_impl(c, p, cost, g, cand_begin, cand_end, want_best_states, nullptr);
}

int packed_grids(rp_ctx *c, int32_t n_levels, const rp_grids *dims, rp_grids *out) {
    if (c->fast_buf.empty()) return fail(c, RP_EINVAL, "RP_PLAN_PACKED: rp_fast_buffer was never asked for");
    const double *b = c->fast_buf.data();
    size_t at = 0;   // doubles
    for (int k = 0; k < n_levels; ++k) {
        if (dims[k].struct_size != sizeof(rp_grids)) return fail(c, RP_EABI, "RP_PLAN_PACKED: rp_grids.struct_size is not this library's");
        const int nT = dims[k].nT, nL = dims[k].nL, nD = dims[k].nD;
        if (nT < 0 || nL < 0 || nD < 0) return fail(c, RP_EINVAL, "RP_PLAN_PACKED: negative grid size");
        const size_t words = (size_t)nT + nL + nD + ((size_t)nT + 1) / 2;
        if (at + words > c->fast_buf.size()) return fail(c, RP_EINVAL, "RP_PLAN_PACKED: grids do not fit the buffer of rp_fast_buffer");
        out[k].struct_size = (uint32_t)sizeof(rp_grids);
        out[k].nT = nT; out[k].nL = nL; out[k].nD = nD;
        out[k].T = b + at; out[k].L = b + at + nT; out[k].D = b + at + nT + nL;
        out[k].traj_len = reinterpret_cast<const int32_t *>(b + at + nT + nL + nD);
        at += words;
    }
    return RP_OK;
}

int plan_begin_impl(rp_ctx *c, const rp_params *p, const rp_cost *cost, const rp_grids *g, int64_t cand_begin, int64_t cand_end,
                    int32_t want_best_states, const LevelGate *lg) {
    rp_result dummy_result = RP_RESULT_INIT;
    rp_result *const result = &dummy_result;   // (validate checks the caller's in the entry point; here only params and cost)
    double *const best_states = want_best_states ? reinterpret_cast<double *>(c) : nullptr;   // (only its being non-null matters below)
    int rc = validate(c, p, cost, result);
    if (rc != RP_OK) return rc;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_plan: a plan is already in flight on this context (rp_plan_wait first)");
    c->have_last = false;   // (before anything of the last plan -- inline grids, staging area -- is overwritten: an early return leaves no half-valid plan behind)
    if (c->timing) c->t_entry = std::chrono::steady_clock::now();
    if (!g || g->nT < 0 || g->nL < 0 || g->nD < 0) return fail(c, RP_EINVAL, "rp_plan: bad grids");
    if (g->struct_size != sizeof(rp_grids)) return fail(c, RP_EABI, "rp_plan: rp_grids.struct_size is not this library's");
    const int64_t total = (int64_t)g->nT * g->nL * g->nD;
    if (total > 0x7fffffffLL) return fail(c, RP_EINVAL, "rp_plan: more than 2^31 - 1 candidates in one grid");
    if (total > 0 && (!g->T || !g->traj_len || !g->L || !g->D)) return fail(c, RP_EINVAL, "rp_plan: null grid array");
    if (cand_end < 0) cand_end = total;
    if (cand_begin < 0 || cand_begin > cand_end || cand_end > total) return fail(c, RP_EINVAL, "rp_plan: bad candidate range");
    for (int i = 0; i < g->nT; ++i)
        if (!(g->T[i] > 0.0) || g->traj_len[i] < 1) return fail(c, RP_EINVAL, "rp_plan: T must be positive, traj_len >= 1");
    const int64_t count = cand_end - cand_begin;
    const int n = p->N + 1;
    bool mat = (p->flags & RP_FLAG_MATERIALIZE_ALL) != 0 || cost->kind == RP_COST_EXTERNAL;
    // Small batches whose winner block is wanted: write every candidate's state rows anyway.  The alternative -- a
    // second launch that re-evaluates the winner -- costs 12.6 us on cfg2, the rows of 7 440 candidates 2-3 us.
    if (!mat && best_states && (size_t)count * RP_N_ARRAYS * (size_t)n * sizeof(double) <= kAutoMaterializeBytes &&
        c->opt.auto_materialize)
        mat = true;
    HIP_TRY(c, hipSetDevice(c->device));

    // stage grids: [T | L | D | traj_len]
    const size_t nd = (size_t)g->nT + g->nL + g->nD;
    const size_t sbytes = nd * sizeof(double) + (size_t)g->nT * sizeof(int32_t);
    if ((rc = ensure_stage(c, sbytes)) != RP_OK) return rc;
    c->cin_count = -1;
    double *hs = reinterpret_cast<double *>(c->h_stage);
    std::memcpy(hs, g->T, sizeof(double) * g->nT);
    std::memcpy(hs + g->nT, g->L, sizeof(double) * g->nL);
    std::memcpy(hs + g->nT + g->nL, g->D, sizeof(double) * g->nD);
    std::memcpy(hs + nd, g->traj_len, sizeof(int32_t) * g->nT);
    const bool grids_inline = sbytes <= sizeof(c->kargs_g.grid) && c->opt.inline_grids;
    c->grids_pending = false;
    if (grids_inline) {   // the grids ride in the kernarg segment of the launches: no copy on the stream
        std::memcpy(c->kargs_g.grid, c->h_stage, sbytes);
        std::memcpy(c->kargs_gl.grid, c->h_stage, sbytes);
        c->staged.assign(c->h_stage, c->h_stage + sbytes);   // (host copy: the winner's coefficients are worked out from it)
        c->staged_on_device = false;
    } else if (sbytes && (!c->staged_on_device || c->staged.size() != sbytes || std::memcmp(c->staged.data(), c->h_stage, sbytes) != 0)) {
        c->staged.assign(c->h_stage, c->h_stage + sbytes);
        c->staged_on_device = false;
        c->grids_pending = true;   // (run_pipeline knows the launch path: rp_lon_kernel's launch block, or a copy)
    }

    if ((rc = grow(c, c->d_status, c->cap_status, (size_t)count)) != RP_OK) return rc;
    if ((rc = grow(c, c->d_cost, c->cap_cost, (size_t)count)) != RP_OK) return rc;
    if (mat && (rc = grow(c, c->d_states, c->cap_states, (size_t)count * RP_N_ARRAYS * (size_t)((n + 15) & ~15))) != RP_OK) return rc;

    KArgs ka;
    {   // (the grid over the static shapes belongs to the ego rectangle of the plan; fill_common copies its descriptor)
        const int grc = ensure_static_grid(c, std::sqrt(0.25 * p->length * p->length + 0.25 * p->width * p->width));
        if (grc != RP_OK) return grc;
    }
    fill_common(c, p, cost, ka);
    const double *ds = reinterpret_cast<const double *>(c->d_stage);
    ka.nT = g->nT; ka.nL = g->nL; ka.nD = g->nD; ka.grids_inline = grids_inline ? 1 : 0;
    ka.T = ds; ka.L = ds + g->nT; ka.D = ds + g->nT + g->nL;
    ka.traj_len = reinterpret_cast<const int32_t *>(ds + nd);
    ka.cand_begin = cand_begin; ka.count = count;
    ka.status = c->d_status; ka.cost = c->d_cost;
    ka.states = mat ? c->d_states : nullptr;
    ka.coeffs = nullptr;
    // inputs of the (pair, step) broad phase of the collision query (rp_kernels.h: near_mask_step)
    ka.use_near_mask = g->nD > 0 ? 1 : 0;
    ka.lat_dmin = ka.lat_dmax = g->nD > 0 ? g->D[0] : 0.0;
    for (int i = 1; i < g->nD; ++i) { ka.lat_dmin = std::min(ka.lat_dmin, g->D[i]); ka.lat_dmax = std::max(ka.lat_dmax, g->D[i]); }
    ka.lat_abs_d = std::max(std::fabs(p->x0_lat[0]), std::max(std::fabs(ka.lat_dmin), std::fabs(ka.lat_dmax)));
    // (NaN / inf samples: pair_step_bound reports "no bound" and the masks come out all ones.  The evaluation kernels of grid
    //  plans always take the masked query: use_near_mask only tells the profile kernels whether there is anything to mask.)
    if (c->obs.n_dyn <= 0 && c->obs.n_clus == 0) ka.use_near_mask = 0;
    table_window(c, p, g, ka);
    c->have_last = false;
    if (lg) { ka.gate = lg->gate; ka.gate_seq = lg->seq; ka.gate_level = lg->level; }
    rc = pipeline_begin(c, ka, mat, false, false, best_states != nullptr, lg ? lg->last : true);
    if (rc != RP_OK) return rc;
    c->pending.mat = mat; c->pending.coeffs = false;
    return RP_OK;
}

// Can this level ride in a chain?  Its grids travel in the launches' kernarg segments (the staging buffer is one per context), its
// epilogue is the one-workgroup kernel, and the plan needs no decision of the host between its kernels.
bool level_chainable(const rp_ctx *c, const rp_params *p, const rp_grids *g) {
    if (!g || g->nT < 0 || g->nL < 0 || g->nD < 0) return false;
    const int64_t total = (int64_t)g->nT * g->nL * g->nD;
    const size_t sbytes = ((size_t)g->nT + g->nL + g->nD) * sizeof(double) + (size_t)g->nT * sizeof(int32_t);
    if (total > RP_FINALIZE_MAX || sbytes > sizeof(((KArgsG *)nullptr)->grid) || !c->opt.inline_grids) return false;
    // (plans of this size that want the winner's rows write every candidate's: no winner re-evaluation, no cost-ordered stage)
    return c->opt.auto_materialize && (size_t)total * RP_N_ARRAYS * (size_t)(p->N + 1) * sizeof(double) <= kAutoMaterializeBytes;
}

int plan_levels_begin(rp_ctx *c, const rp_params *p, const rp_cost *cost, int32_t n_levels, const rp_grids *grids, int32_t want_best_states) {
    if (!c) return RP_EINVAL;
    if (n_levels < 1 || n_levels > 64 || !grids) return fail(c, RP_EINVAL, "rp_plan_levels: need 1 .. 64 levels");
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_plan_levels: a plan is already in flight on this context (rp_plan_wait first)");
    c->chain.clear(); c->chain_total = n_levels; c->last_level = 0;
    // the caller's inputs are kept: levels behind the chain (too large for it) run one by one inside rp_plan_wait
    rp_ctx::ChainInputs &in = c->chain_in;
    if (!p || !cost) return fail(c, RP_EINVAL, "null params / cost / result");
    in.p = *p; in.cost = *cost; in.want = want_best_states;
    in.grids.assign(grids, grids + n_levels);
    in.T.resize(n_levels); in.L.resize(n_levels); in.D.resize(n_levels); in.tl.resize(n_levels);
    for (int k = 0; k < n_levels; ++k) {
        const rp_grids &g = grids[k];
        if (g.struct_size != sizeof(rp_grids)) return fail(c, RP_EABI, "rp_plan_levels: rp_grids.struct_size is not this library's");
        if (g.nT < 0 || g.nL < 0 || g.nD < 0 || ((int64_t)g.nT * g.nL * g.nD > 0 && (!g.T || !g.traj_len || !g.L || !g.D)))
            return fail(c, RP_EINVAL, "rp_plan_levels: bad grids");
        in.T[k].assign(g.T, g.T + (g.T ? g.nT : 0)); in.L[k].assign(g.L, g.L + (g.L ? g.nL : 0)); in.D[k].assign(g.D, g.D + (g.D ? g.nD : 0));
        in.tl[k].assign(g.traj_len, g.traj_len + (g.traj_len ? g.nT : 0));
        in.grids[k].T = in.T[k].data(); in.grids[k].L = in.L[k].data(); in.grids[k].D = in.D[k].data(); in.grids[k].traj_len = in.tl[k].data();
    }
    // the chain: leading levels that can ride in it (at least the first level is launched, chainable or not -- then as a plan on its own)
    int n_chain = 0;
    while (n_chain < n_levels && level_chainable(c, p, &grids[n_chain])) ++n_chain;
    if (n_chain <= 1) {   // nothing to chain: the first level as an ordinary plan, the others (if any) behind it in rp_plan_wait
        const int rc = plan_begin_impl(c, p, cost, &grids[0], 0, -1, want_best_states, nullptr);
        if (rc != RP_OK) { c->chain_total = 0; return rc; }
        rp_ctx::LevelState st;
        st.grid_index = 0;
        c->chain.push_back(st);   // (marks "a levels call": rp_plan_wait goes on with level 1 when this one has no winner)
        c->chain.back().pending = c->pending;
        return RP_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->d_gate) {
        HIP_TRY(c, hipMalloc((void **)&c->d_gate, sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_gate, 0, sizeof(unsigned long long)));
    }
    int rc = ensure_result(c, p->N + 1);
    if (rc != RP_OK) return rc;
    {   // every buffer the levels share has its final size BEFORE the first launch: a buffer that grew between two levels would be
        // freed under the kernels of the level before (and the context's "last plan" would point into the new, empty one)
        int64_t cmax = 0, pmax = 0;
        for (int k = 0; k < n_chain; ++k) {
            cmax = std::max<int64_t>(cmax, (int64_t)grids[k].nT * grids[k].nL * grids[k].nD);
            pmax = std::max<int64_t>(pmax, (int64_t)grids[k].nT * grids[k].nL);
        }
        const int n = p->N + 1;
        if ((rc = grow(c, c->d_status, c->cap_status, (size_t)cmax)) != RP_OK) return rc;
        if ((rc = grow(c, c->d_cost, c->cap_cost, (size_t)cmax)) != RP_OK) return rc;
        if ((rc = grow(c, c->d_states, c->cap_states, (size_t)cmax * RP_N_ARRAYS * (size_t)((n + 15) & ~15))) != RP_OK) return rc;
        if ((rc = grow(c, c->d_profile, c->cap_profile, (size_t)pmax * PF_FIELDS * (size_t)n)) != RP_OK) return rc;
        if ((rc = grow(c, c->d_pair_hdr, c->cap_pair_hdr, (size_t)pmax)) != RP_OK) return rc;
    }
    const unsigned long long seq = ++c->seq;
    reinterpret_cast<ResultBlock *>(c->h_result)->seq = 0;
    int launched = 0;
    for (int k = 0; k < n_chain; ++k) {
        const int64_t total = (int64_t)grids[k].nT * grids[k].nL * grids[k].nD;
        bool later = false;   // is there a later chain level with candidates?
        for (int q = k + 1; q < n_chain; ++q) later = later || (int64_t)grids[q].nT * grids[q].nL * grids[q].nD > 0;
        if (total == 0 && (later || launched > 0)) continue;   // an empty level has nothing to say (the reference's loop moves on: :616-636)
        const LevelGate lg = {c->d_gate, seq, launched + 1, !later};
        c->pending.active = false;
        if (launched > 0) c->epilogue_dirty = false;   // (the level before is on the stream: its epilogue leaves the scratch words as it found them)
        rc = plan_begin_impl(c, p, cost, &grids[k], 0, -1, want_best_states, &lg);
        if (rc != RP_OK) { c->chain.clear(); c->chain_total = 0; c->pending.active = false; return rc; }
        rp_ctx::LevelState st;
        st.pending = c->pending; st.staged = c->staged; st.staged_on_device = c->staged_on_device; st.last_rows_on_device = c->last_rows_on_device;
        st.last_G = c->last_G; st.last_block = c->last_block; st.last_fused_lds = c->last_fused_lds; st.grid_index = k;
        st.last_kernel = c->last_kernel; st.last_lazy = c->last_lazy;
        c->chain.push_back(std::move(st));
        ++launched;
    }
    c->pending.active = true;   // (one plan in flight, as far as the other entry points are concerned)
    return RP_OK;
}

}  // namespace

extern "C" {

int rp_last_level(const rp_ctx *c) { return c ? c->last_level : 0; }

int rp_plan_levels(rp_ctx *c, const rp_params *p, const rp_cost *cost, int32_t n_levels, const rp_grids *grids, uint32_t flags, rp_result *result,
                   double *out, int32_t *level) {
    if (!c) return RP_EINVAL;
    if (flags & ~(RP_PLAN_BEGIN | RP_PLAN_ROWS | RP_PLAN_PACKED)) return fail(c, RP_EINVAL, "rp_plan_levels: unknown flag");
    if (n_levels < 1 || n_levels > 64 || !grids) return fail(c, RP_EINVAL, "rp_plan_levels: need 1 .. 64 levels");
    rp_grids pg[64];
    if (flags & RP_PLAN_PACKED) {
        if (!p || !cost) return fail(c, RP_EINVAL, "rp_plan_levels: null argument");
        if (!(flags & RP_PLAN_BEGIN) && !out) return fail(c, RP_EINVAL, "rp_plan_levels: RP_PLAN_PACKED needs the output block");
        const int grc = packed_grids(c, n_levels, grids, pg);
        if (grc != RP_OK) return grc;
        grids = pg;
    }
    if (flags & RP_PLAN_BEGIN) return plan_levels_begin(c, p, cost, n_levels, grids, ((flags & RP_PLAN_ROWS) || out) ? 1 : 0);
    if (!result) return fail(c, RP_EINVAL, "null params / cost / result");
    if (result->struct_size != sizeof(rp_result)) return fail(c, RP_EABI, "rp_plan_levels: rp_result.struct_size is not this library's");
    int rc = plan_levels_begin(c, p, cost, n_levels, grids, out != nullptr ? 1 : 0);
    if (rc == RP_OK) rc = rp_plan_wait(c, result, out);
    if (rc == RP_OK && level) *level = c->last_level;
    if (rc != RP_OK || !(flags & RP_PLAN_PACKED) || result->best_index < 0) return rc;
    const int n = p->N + 1;
    return rp_pack_trajectory(n, out, p->dt, p->wheelbase, p->x0_orientation, out + (size_t)RP_N_ARRAYS * n);
}

int rp_plan_cycle(rp_ctx *c, rp_cycle *cy, const rp_params *tmpl, const rp_cost *cost, int32_t n_levels, const rp_grids *dims, rp_result *result,
                  double *out) {
    if (!c) return RP_EINVAL;
    if (!cy || !tmpl || !cost || !dims || !result || !out) return fail(c, RP_EINVAL, "rp_plan_cycle: null argument");
    if (cy->struct_size != sizeof(rp_cycle)) return fail(c, RP_EABI, "rp_plan_cycle: rp_cycle.struct_size is not this library's");
    if (n_levels < 1 || n_levels > 64) return fail(c, RP_EINVAL, "rp_plan_cycle: need 1 .. 64 levels");
    int rc = validate(c, tmpl, cost, result);
    if (rc != RP_OK) return rc;
    rp_params p = *tmpl;
    // 1. curvilinear initial state
    if (!cy->have_curvilinear) {
        const rpfe::Tables &tb = c->h_front;
        double s = 0.0, d = 0.0;
        if (!rpfe::project(tb.ref, tb.pos, c->proj_d_limit, cy->x, cy->y, s, d)) return fail(c, RP_EDOMAIN, "rp_plan_cycle: initial state outside the projection domain");
        if (rpfe::initial_state(tb, s, d, cy->orientation, cy->velocity, cy->acceleration, cy->steering_angle, p.wheelbase, cy->prev_low_vel_mode != 0,
                                cy->x0_lon, cy->x0_lat) != 0)
            return fail(c, RP_EDIRECTION, "rp_plan_cycle: negative longitudinal velocity along the reference path");
    }
    // 2. low-velocity mode of this cycle
    cy->low_vel_mode = cy->velocity < cy->low_vel_mode_threshold ? 1 : 0;
    p.low_vel_mode = cy->low_vel_mode;
    for (int k = 0; k < 3; ++k) { p.x0_lon[k] = cy->x0_lon[k]; p.x0_lat[k] = cy->x0_lat[k]; }
    p.x0_orientation = cy->orientation;
    // 3. the grids of the levels
    rp_grids staged[64], g[64];
    if ((rc = packed_grids(c, n_levels, dims, staged)) != RP_OK) return rc;
    size_t words = 0;
    for (int k = 0; k < n_levels; ++k) words += (size_t)staged[k].nT * 2 + (size_t)staged[k].nL + (size_t)staged[k].nD + 2;
    if (c->cycle_buf.size() < words) c->cycle_buf.assign(words + 64, 0.0);
    double *w = c->cycle_buf.data();
    std::vector<double> dvals;
    for (int k = 0; k < n_levels; ++k) {
        const rp_grids &sg = staged[k];
        g[k].struct_size = (uint32_t)sizeof(rp_grids);
        g[k].nT = sg.nT;
        g[k].T = w;
        std::memcpy(w, sg.T, sizeof(double) * (size_t)sg.nT); w += sg.nT;
        g[k].traj_len = reinterpret_cast<const int32_t *>(w);
        std::memcpy(w, sg.traj_len, sizeof(int32_t) * (size_t)sg.nT); w += ((size_t)sg.nT + 1) / 2;
        g[k].L = w;
        int nL = 0;
        for (int i = 0; i < sg.nL; ++i)   // stopping mode: goals behind the vehicle are no candidates (trajectories.py:545-550)
            if (p.lon_mode != RP_LON_STOPPING || sg.L[i] > cy->x0_lon[0]) w[nL++] = sg.L[i];
        g[k].nL = nL; w += nL;
        dvals.clear();
        rpco::PySetF64::from_values(sg.D, sg.nD).union_with(&cy->x0_lat[0], 1).values(dvals);   // sampling.py:226
        g[k].D = w; g[k].nD = (int32_t)dvals.size();
        std::memcpy(w, dvals.data(), sizeof(double) * dvals.size()); w += dvals.size();
    }
    // 4. the level loop, the output packing (one level: a plan on its own -- no copy of the inputs for levels behind a chain)
    rc = n_levels == 1 ? plan_begin(c, &p, cost, &g[0], 0, -1, 1) : plan_levels_begin(c, &p, cost, n_levels, g, 1);
    if (rc == RP_OK) rc = rp_plan_wait(c, result, out);
    if (rc != RP_OK) return rc;
    cy->level = c->last_level;
    if (result->best_index < 0) return RP_OK;
    const int n = p.N + 1;
    return rp_pack_trajectory(n, out, p.dt, p.wheelbase, p.x0_orientation, out + (size_t)RP_N_ARRAYS * n);
}

int rp_coeffs_arena(rp_ctx *c, int64_t cap, double **lon_coeffs, double **lat_coeffs, int32_t **traj_len) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_coeffs_arena: a plan is in flight on this context (rp_plan_wait first)");
    if (cap <= 0 || !lon_coeffs || !lat_coeffs || !traj_len) return fail(c, RP_EINVAL, "rp_coeffs_arena: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (cap > c->arena_cap) {
        if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_arena) HIP_TRY(c, hipHostFree(c->h_arena));
        c->h_arena = c->h_arena_dev = nullptr; c->arena_cap = 0;
        c->have_last = false;   // (the last plan's polynomials may have lived in the old arena)
        c->cin_lon = c->cin_lat = nullptr; c->cin_count = -1;
        const size_t bytes = (size_t)cap * (12 * sizeof(double) + 3 * sizeof(int32_t));   // lon | lat | traj_len | group | group_first
        if (hipHostMalloc((void **)&c->h_arena, bytes, hipHostMallocDefault) != hipSuccess) { c->h_arena = nullptr; return fail(c, RP_ENOMEM, "rp_coeffs_arena: pinned host memory"); }
        if (hipHostGetDevicePointer((void **)&c->h_arena_dev, c->h_arena, 0) != hipSuccess) { (void)hipHostFree(c->h_arena); c->h_arena = c->h_arena_dev = nullptr; return fail(c, RP_EHIP, "rp_coeffs_arena: device address of the pinned arrays"); }
        c->arena_cap = cap;
    }
    double *base = reinterpret_cast<double *>(c->h_arena);
    *lon_coeffs = base; *lat_coeffs = base + 6 * c->arena_cap;
    *traj_len = reinterpret_cast<int32_t *>(base + 12 * c->arena_cap);
    return RP_OK;
}

int rp_coeffs_arena_groups(rp_ctx *c, int32_t **group, int32_t **group_first) {
    if (!c) return RP_EINVAL;
    if (!c->h_arena || !group || !group_first) return fail(c, RP_ESTATE, "rp_coeffs_arena_groups: no arena (rp_coeffs_arena first) / null output");
    int32_t *tl = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(c->h_arena) + 12 * c->arena_cap);
    *group = tl + c->arena_cap;
    *group_first = tl + 2 * c->arena_cap;
    return RP_OK;
}

int rp_plan_coeffs(rp_ctx *c, const rp_params *p, const rp_cost *cost, int64_t C, const double *lon_coeffs,
                   const double *lat_coeffs, const double *lon_T, const int32_t *traj_len, int64_t n_groups, const int32_t *group,
                   const int32_t *group_first, rp_result *result, double *best_states) {
    int rc = validate(c, p, cost, result);
    if (rc != RP_OK) return rc;
    if (c->timing) c->t_entry = std::chrono::steady_clock::now();
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_plan_coeffs: a plan is in flight on this context (rp_plan_wait first)");
    c->have_last = false;
    (void)lon_T;
    if (C < 0 || (C > 0 && (!lon_coeffs || !lat_coeffs || !traj_len))) return fail(c, RP_EINVAL, "rp_plan_coeffs: bad arrays");
    const int n = p->N + 1;
    const bool mat = (p->flags & RP_FLAG_MATERIALIZE_ALL) != 0 || cost->kind == RP_COST_EXTERNAL;
    HIP_TRY(c, hipSetDevice(c->device));
    // the candidates' arrays -> device.  Arrays handed out by rp_coeffs_arena are pinned already and go as they are; anything
    // else is copied into the pinned staging buffer first (2.4 MB at 25 536 candidates: a quarter of a millisecond of memcpy)
    const double *d_lon = nullptr, *d_lat = nullptr;
    const int32_t *d_tl = nullptr;
    const bool no_zero_copy = !c->opt.zero_copy;   // (A/B: arena rows through the staging copy)
    const bool in_arena = !no_zero_copy && c->h_arena && C <= c->arena_cap && lon_coeffs == reinterpret_cast<const double *>(c->h_arena) &&
                          lat_coeffs == reinterpret_cast<const double *>(c->h_arena) + 6 * c->arena_cap &&
                          traj_len == reinterpret_cast<const int32_t *>(reinterpret_cast<const double *>(c->h_arena) + 12 * c->arena_cap);
    if (in_arena) {
        // the kernels read the pinned arrays themselves (each value once per plan or round: rp_lon_kernel the longitudinal
        // polynomials, the evaluation kernel the lateral ones): three host-to-device copies of 28 + 27 + 7 us with 8-us gaps in
        // front of the first kernel cost more than the reads over PCIe inside it (25 536 candidates: 0.336 -> 0.311 ms)
        const size_t o_lat = sizeof(double) * 6 * (size_t)c->arena_cap, o_tl = 2 * o_lat;
        d_lon = reinterpret_cast<const double *>(c->h_arena_dev);
        d_lat = reinterpret_cast<const double *>(c->h_arena_dev + o_lat);
        d_tl = reinterpret_cast<const int32_t *>(c->h_arena_dev + o_tl);
        c->cin_lon = lon_coeffs; c->cin_lat = lat_coeffs;
    } else {
        const size_t sbytes = (size_t)C * (12 * sizeof(double) + sizeof(int32_t));
        if ((rc = ensure_stage(c, sbytes)) != RP_OK) return rc;
        double *hs = reinterpret_cast<double *>(c->h_stage);
        std::memcpy(hs, lon_coeffs, sizeof(double) * 6 * C);
        std::memcpy(hs + 6 * C, lat_coeffs, sizeof(double) * 6 * C);
        std::memcpy(hs + 12 * C, traj_len, sizeof(int32_t) * C);
        if (sbytes) HIP_TRY(c, hipMemcpyAsync(c->d_stage, c->h_stage, sbytes, hipMemcpyHostToDevice, c->stream));
        const double *ds = reinterpret_cast<const double *>(c->d_stage);
        d_lon = ds; d_lat = ds + 6 * C; d_tl = reinterpret_cast<const int32_t *>(ds + 12 * C);
        c->cin_lon = hs; c->cin_lat = hs + 6 * C;
    }
    c->staged.clear();
    c->staged_on_device = false;
    c->cin_count = C;
    if ((rc = grow(c, c->d_status, c->cap_status, (size_t)C)) != RP_OK) return rc;
    if ((rc = grow(c, c->d_cost, c->cap_cost, (size_t)C)) != RP_OK) return rc;
    if (mat && (rc = grow(c, c->d_states, c->cap_states, (size_t)C * RP_N_ARRAYS * (size_t)((n + 15) & ~15))) != RP_OK) return rc;
    KArgs ka;
    {   // (the grid over the static shapes belongs to the ego rectangle of the plan; fill_common copies its descriptor)
        const int grc = ensure_static_grid(c, std::sqrt(0.25 * p->length * p->length + 0.25 * p->width * p->width));
        if (grc != RP_OK) return grc;
    }
    fill_common(c, p, cost, ka);
    ka.lon_coeffs = d_lon; ka.lat_coeffs = d_lat;
    ka.traj_len_c = d_tl;
    ka.cand_begin = 0; ka.count = C;
    // candidates in groups with a common longitudinal polynomial: one profile per group (pair = group).  Out of the arena only -- the
    // kernels read the group tables where the sampling space wrote them, as they read the polynomials.
    c->cin_groups = 0; c->cin_group_of = nullptr;
    const bool no_groups = !c->opt.coeff_groups;   // (A/B: one profile per candidate)
    // (a level small enough for the single-launch variant keeps one pair per candidate: one launch less beats the shared profiles there)
    int fused_pairs_q = 0;
    const bool single_launch = c->opt.fused_lon &&
                               fused_lon_lds(c, ka, C, lanes_per_candidate(c, ka.N, C, mat), true, mat, &fused_pairs_q) != 0;
    if (n_groups > 0 && group && group_first && !no_zero_copy) {   // (RP_AMD_NO_ZERO_COPY: the A/B path copies the rows, the group tables stay behind)
        int32_t *tl_h = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(c->h_arena) + 12 * c->arena_cap);
        if (!in_arena || group != tl_h + c->arena_cap || group_first != tl_h + 2 * c->arena_cap || n_groups > C)
            return fail(c, RP_EINVAL, "rp_plan_coeffs: groups come with candidates out of the context's arena (rp_coeffs_arena, rp_coeffs_arena_groups)");
        // (the kernels index profiles and polynomials through these tables: every entry is looked at -- ~10 us at 25 000 candidates)
        bool tables_ok = C > 0 && group[0] == 0 && group_first[0] == 0 && group[C - 1] == (int32_t)(n_groups - 1);
        for (int64_t i = 1; tables_ok && i < C; ++i) {
            const int32_t step = group[i] - group[i - 1];
            tables_ok = step == 0 || (step == 1 && group_first[group[i]] == (int32_t)i);
        }
        if (!tables_ok)
            return fail(c, RP_EINVAL, "rp_plan_coeffs: groups must be adjacent and numbered 0 .. n_groups - 1 in order, group_first their first candidates");
        if (!no_groups && !single_launch) {   // (tables checked either way: what a call accepts does not depend on the batch size)
            const int32_t *tl_d = reinterpret_cast<const int32_t *>(reinterpret_cast<const double *>(c->h_arena_dev) + 12 * c->arena_cap);
            ka.pair_of = tl_d + c->arena_cap;
            ka.group_first = tl_d + 2 * c->arena_cap;
            c->cin_groups = n_groups; c->cin_group_of = group;
        }
    }
    ka.status = c->d_status; ka.cost = c->d_cost;
    ka.states = mat ? c->d_states : nullptr;
    c->have_last = false;
    rc = run_pipeline(c, ka, mat, true, false, result, best_states);
    if (rc != RP_OK) return rc;
    c->last = ka; c->have_last = true; c->last_mat = mat; c->last_coeffs = true;
    return RP_OK;
}

int rp_fetch_status(rp_ctx *c, int64_t first, int64_t count, uint32_t *status, double *cost) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_fetch_status: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last) return fail(c, RP_ESTATE, "rp_fetch_status: no plan on this context");
    if (first < 0 || count < 0 || first + count > c->last.count) return fail(c, RP_EINVAL, "rp_fetch_status: range");
    HIP_TRY(c, hipSetDevice(c->device));
    if (count == 0) return RP_OK;
    if (status) HIP_TRY(c, hipMemcpy(status, c->d_status + first, sizeof(uint32_t) * count, hipMemcpyDeviceToHost));
    if (cost) HIP_TRY(c, hipMemcpy(cost, c->d_cost + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    return RP_OK;
}

int rp_fetch_states(rp_ctx *c, int64_t first, int64_t count, double *states) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_fetch_states: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last || !c->last_mat) return fail(c, RP_ESTATE, "rp_fetch_states: last plan did not materialise states");
    if (first < 0 || count < 0 || first + count > c->last.count || (count && !states))
        return fail(c, RP_EINVAL, "rp_fetch_states: range");
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = c->last.N + 1, ns = c->last.row_stride, tail = c->last.tail_split;
    const size_t blk = (size_t)RP_N_ARRAYS * (size_t)n;
    if (!count) return RP_OK;
    if (ns == n && !tail) {
        HIP_TRY(c, hipMemcpy(states, c->d_states + blk * first, sizeof(double) * blk * count, hipMemcpyDeviceToHost));
        return RP_OK;
    }
    // padded rows: compact on the device (a copy at memory speed), then a linear transfer -- in pieces of at most 256 MB, so that
    // fetching every block of a 10^6-candidate batch does not need a second 11-GB buffer
    const int64_t piece = std::max<int64_t>(1, (int64_t)((256u << 20) / (blk * sizeof(double))));
    int rc;
    if ((rc = grow(c, c->d_compact, c->cap_compact, blk * (size_t)std::min<int64_t>(piece, count))) != RP_OK) return rc;
    const uint32_t inv_n = (uint32_t)(0x100000000ull / (unsigned long long)n) + 1u;
    for (int64_t at = 0; at < count; at += piece) {
        const int64_t cnt = std::min<int64_t>(piece, count - at);
        const size_t total = blk * (size_t)cnt;
        const int grid = (int)std::min<size_t>((total + 255) / 256, (size_t)c->num_cus * 16);
        hipLaunchKernelGGL(rp_compact_rows_kernel, dim3(grid), dim3(256), 0, c->stream,
                           c->d_states + (size_t)state_block_doubles(ns, tail) * (size_t)(first + at), c->d_compact, n, ns, tail, inv_n,
                           (unsigned long long)total);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(states + blk * (size_t)at, c->d_compact, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));   // (the piece buffer is reused)
    }
    return RP_OK;
}

int rp_eval_one(rp_ctx *c, int64_t index, double *states, uint32_t *status, double *cost) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_eval_one: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last) return fail(c, RP_ESTATE, "rp_eval_one: no plan on this context");
    const KArgs &l = c->last;
    const int64_t total = c->last_coeffs ? l.count : (int64_t)l.nT * l.nL * l.nD;
    if (index < 0 || index >= total) return fail(c, RP_EINVAL, "rp_eval_one: index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = l.N + 1;
    int rc;
    if ((rc = ensure_result(c, n)) != RP_OK) return rc;
    ResultBlock *drb = reinterpret_cast<ResultBlock *>(c->d_result);
    *c->h_single = index;
    HIP_TRY(c, hipMemcpyAsync(c->d_single, c->h_single, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    KArgs kw = l;
    kw.single_index = c->d_single;
    kw.count = 1;
    kw.status = &drb->pad_;
    kw.cost = &drb->w_cost;
    kw.states = reinterpret_cast<double *>(c->d_result + sizeof(ResultBlock));
    kw.row_stride = n;
    kw.tail_split = 0;
    kw.coeffs = nullptr;
    kw.partials = nullptr;
    // the candidate's pair may lie outside the last plan's shard: give it a one-pair profile of its own
    if ((rc = grow(c, c->d_profile_one, c->cap_profile_one, (size_t)PF_FIELDS * (size_t)n)) != RP_OK) return rc;
    kw.pair_begin = c->last_coeffs ? (l.pair_of && c->cin_group_of ? (int64_t)c->cin_group_of[index] : index) : index / l.nD;
    kw.pair_count = 1;
    kw.profile = c->d_profile_one;
    kw.pair_hdr = c->d_pair_hdr_one;
    const int G1 = c->last_G ? c->last_G : lanes_per_candidate(c, kw.N, 1, true);   // the batch's choice: same code path, same bits
    if (c->last_fused_lds) {   // same code path as the batch: results agree bit for bit
        launch_eval_fused(c, kw, 1, true, c->last_coeffs, c->last_fused_lds, G1);
    } else {
        launch_lon(c, kw, c->last_coeffs);
        launch_eval(c, kw, 1, true, c->last_coeffs, G1);
    }
    const size_t bytes = sizeof(ResultBlock) + sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n;
    HIP_TRY(c, hipMemcpyAsync(c->h_result, c->d_result, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    const ResultBlock *hrb = reinterpret_cast<const ResultBlock *>(c->h_result);
    if (status) *status = hrb->pad_;
    if (cost) *cost = hrb->w_cost;
    if (states) std::memcpy(states, c->h_result + sizeof(ResultBlock), sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n);
    return RP_OK;
}

int rp_count_collisions_before(rp_ctx *c, double cost, int64_t index, int64_t *count) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_count_collisions_before: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last || !count) return fail(c, RP_ESTATE, "rp_count_collisions_before: no plan / null output");
    // after a cost-ordered plan the labels are complete up to the plan's own winner (all of them without one): a key behind it
    // would count candidates nobody looked at
    if ((c->last_lazy == RP_PATH_LAZY || c->last_lazy == RP_PATH_SWEEP) && c->last_best_index >= 0 && !(cost < c->last_best_cost || (cost == c->last_best_cost && index <= c->last_best_index)))
        return fail(c, RP_ESTATE, "rp_count_collisions_before: (cost, index) sorts behind the winner of a plan that answered the collision "
                                  "query in cost order (rp_last_path() == RP_PATH_LAZY)");
    HIP_TRY(c, hipSetDevice(c->device));
    ResultBlock *drb = reinterpret_cast<ResultBlock *>(c->d_result);
    const KArgs &l = c->last;
    *count = 0;
    if (l.count == 0) return RP_OK;
    HIP_TRY(c, hipMemsetAsync(&drb->n_before, 0, sizeof(unsigned long long), c->stream));
    const int cgrid = (int)std::max<int64_t>(1, std::min<int64_t>((l.count + RP_BLOCK - 1) / RP_BLOCK, (int64_t)c->num_cus * 4));
    hipLaunchKernelGGL(rp_count_before_kernel, dim3(cgrid), dim3(RP_BLOCK), 0, c->stream, l.status, l.cost, l.count,
                       l.cand_begin, &drb->r, cost, index, 1, &drb->n_before);
    unsigned long long *hn = reinterpret_cast<unsigned long long *>(c->h_single);
    HIP_TRY(c, hipMemcpyAsync(hn, &drb->n_before, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    *count = (int64_t)*hn;
    return RP_OK;
}

int rp_select(rp_ctx *c, const double *costs, int64_t count, rp_result *result, double *best_states) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_select: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last) return fail(c, RP_ESTATE, "rp_select: no plan on this context");
    if (!result || count != c->last.count || (count && !costs)) return fail(c, RP_EINVAL, "rp_select: count mismatch");
    if (result->struct_size != sizeof(rp_result)) return fail(c, RP_EABI, "rp_select: rp_result.struct_size is not this library's");
    // a cost-ordered plan (rp_last_path() == RP_PATH_LAZY) answered the collision query for the candidates it had to look at only:
    // the others keep RP_LABEL_FEASIBLE, and a selection with other costs could crown a colliding one
    if (c->last_lazy == RP_PATH_LAZY || c->last_lazy == RP_PATH_SWEEP)
        return fail(c, RP_ESTATE, "rp_select: the last plan answered the collision query in cost order (rp_last_path() == RP_PATH_LAZY); "
                                  "plan with RP_FLAG_MATERIALIZE_ALL / RP_COST_EXTERNAL, which take the eager query");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, c->d_user, c->cap_user, (size_t)count)) != RP_OK) return rc;
    if (count) {
        HIP_TRY(c, hipMemcpyAsync(c->d_user, costs, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((count + RP_BLOCK - 1) / RP_BLOCK, 1024));
        hipLaunchKernelGGL(rp_apply_costs_kernel, dim3(grid), dim3(RP_BLOCK), 0, c->stream, c->d_user, c->d_cost, c->d_status, count);
    }
    KArgs ka = c->last;
    return run_pipeline(c, ka, c->last_mat, c->last_coeffs, true, result, best_states);
}

int rp_result_device(rp_ctx *c, const void **ptr, size_t *bytes, int32_t *rows_valid) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_result_device: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last || !ptr || !bytes) return fail(c, RP_ESTATE, "rp_result_device: no plan / null output");
    // The caller reads the block on ANOTHER stream (the collective's).  rp_plan returns on the host ticket, which says nothing
    // about the device copy as seen from other streams: wait for the context's stream here (its kernels have delivered their
    // ticket already: ~1 us), after which the block is visible to work enqueued anywhere.
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *ptr = c->d_result;
    *bytes = sizeof(ResultBlock) + sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)(c->last.N + 1);
    if (rows_valid) *rows_valid = c->last_rows_on_device ? 1 : 0;
    return RP_OK;
}

int rp_combine_results(rp_ctx *c, const void *d_msgs, int32_t world, void *stream, rp_result *global, double *best_states,
                       int32_t *owner_rank, int32_t *rows_valid) {
    if (!c) return RP_EINVAL;
    if (!c->have_last || !d_msgs || !global) return fail(c, RP_ESTATE, "rp_combine_results: no plan / null argument");
    if (world < 1 || world > RP_COMBINE_MAX_WORLD) return fail(c, RP_EINVAL, "rp_combine_results: world size out of range [1, 64]");
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = c->last.N + 1;
    int rc;
    if ((rc = ensure_result(c, n)) != RP_OK) return rc;
    const size_t msg_bytes = sizeof(ResultBlock) + sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n;
    ResultBlock *hrb_host = reinterpret_cast<ResultBlock *>(c->h_result);
    const unsigned long long seq = ++c->seq;
    hrb_host->seq = 0;
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
    hipLaunchKernelGGL(rp_combine_kernel, dim3(1), dim3(RP_COMBINE_THREADS), 0, st, reinterpret_cast<const char *>(d_msgs), (int)world,
                       msg_bytes, n, reinterpret_cast<ResultBlock *>(c->h_result_dev), seq);
    HIP_TRY(c, hipGetLastError());
    // (the all-gather ahead of the kernel can take a while on a cold communicator: poll with a generous fall-back)
    if (!poll_ticket(c, seq)) {
        HIP_TRY(c, hipStreamSynchronize(st));
        if (__atomic_load_n(&hrb_host->seq, __ATOMIC_ACQUIRE) != seq) return fail(c, RP_EHIP, "rp_combine_results: no completion ticket");
    }
    if (global->struct_size != sizeof(rp_result)) return fail(c, RP_EABI, "rp_combine_results: rp_result.struct_size");
    result_from_core(global, hrb_host->r);
    global->n_collision_before_best = 0;
    global->kernel_ms = 0.0;
    KArgs ka = c->last;
    host_winner_coeffs(c, ka, c->last_coeffs, global);
    if (owner_rank) *owner_rank = (int32_t)hrb_host->pad_;
    const bool rows_ok = !(hrb_host->w_status & RP_WSTATUS_ROWS_ON_HOST);
    if (rows_valid) *rows_valid = rows_ok ? 1 : 0;
    if (global->best_index >= 0 && best_states && rows_ok)
        std::memcpy(best_states, c->h_result + sizeof(ResultBlock), sizeof(double) * (size_t)RP_N_ARRAYS * (size_t)n);
    return RP_OK;
}

int rp_cost_range(rp_ctx *c, double *min_cost, double *max_cost, int64_t *n_out) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_cost_range: a plan is in flight on this context (rp_plan_wait first)");
    if (!c->have_last || !min_cost || !max_cost || !n_out) return fail(c, RP_ESTATE, "rp_cost_range: no plan / null output");
    *min_cost = *max_cost = std::nan("");
    *n_out = 0;
    const KArgs &l = c->last;
    if (l.count == 0) return RP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, c->d_user, c->cap_user, (size_t)4)) != RP_OK) return rc;
    unsigned long long *d_out = reinterpret_cast<unsigned long long *>(c->d_user);
    static const unsigned long long init[3] = {~0ull, 0ull, 0ull};
    HIP_TRY(c, hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((l.count + RP_BLOCK * 8 - 1) / (RP_BLOCK * 8), (int64_t)c->num_cus * 4));
    hipLaunchKernelGGL(rp_cost_range_kernel, dim3(grid), dim3(RP_BLOCK), 0, c->stream, (const uint32_t *)c->d_status, (const double *)c->d_cost,
                       l.count, d_out);
    HIP_TRY(c, hipGetLastError());
    unsigned long long h[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h[2] == 0) return RP_OK;
    auto unkey = [](unsigned long long k) {
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double d;
        std::memcpy(&d, &b, sizeof(d));
        return d;
    };
    *min_cost = unkey(h[0]);
    *max_cost = unkey(h[1]);
    *n_out = (int64_t)h[2];
    return RP_OK;
}

int rp_check_swept(rp_ctx *c, const rp_params *p, int32_t n_poses, const double *x, const double *y, const double *theta,
                   int32_t *first_hit, double *boxes) {
    if (!c) return RP_EINVAL;
    if (c->pending.active) return fail(c, RP_ESTATE, "rp_check_swept: a plan is in flight on this context (rp_plan_wait first)");
    if (!p || !first_hit || n_poses < 0 || (n_poses > 0 && (!x || !y || !theta))) return fail(c, RP_EINVAL, "rp_check_swept: arguments");
    *first_hit = -1;
    if (n_poses < 2) return RP_OK;   // no segment
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)n_poses;
    int rc;
    // poses and results go through the pinned, device-visible mirror of the result block (free between plans: rp_plan has
    // copied its result out before it returned): 9 n doubles behind the header fit where 14 n are reserved
    if ((rc = ensure_result(c, n_poses)) != RP_OK) return rc;
    ResultBlock *hrb_host = reinterpret_cast<ResultBlock *>(c->h_result);
    double *h_poses = reinterpret_cast<double *>(hrb_host + 1);
    std::memcpy(h_poses, x, sizeof(double) * n);
    std::memcpy(h_poses + n, y, sizeof(double) * n);
    std::memcpy(h_poses + 2 * n, theta, sizeof(double) * n);
    const unsigned long long seq = ++c->seq;
    hrb_host->seq = 0;
    hipLaunchKernelGGL(rp_swept_kernel, dim3(1), dim3(RP_SWEPT_THREADS), 0, c->stream, c->obs, (int)n_poses, p->wb_rear_axle, 0.5 * p->length,
                       0.5 * p->width, (int)p->time_step0, reinterpret_cast<ResultBlock *>(c->h_result_dev), boxes ? 1 : 0, seq);
    HIP_TRY(c, hipGetLastError());
    if (c->opt.wait_mode == RP_WAIT_EVENT || !poll_ticket(c, seq)) HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned int hit = (unsigned int)hrb_host->n_before;
    *first_hit = hit < (unsigned int)(n_poses - 1) ? (int32_t)hit : -1;
    if (boxes) std::memcpy(boxes, h_poses + 3 * n, sizeof(double) * 6 * (n - 1));
    return RP_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------
// Reference-path front end (host only; rp_frontend.h)
// ------------------------------------------------------------------------------------------------
extern "C" {

int rp_fast_buffer(rp_ctx *c, void **ptr, size_t *bytes) {
    if (!c || !ptr || !bytes) return RP_EINVAL;
    if (c->fast_buf.empty()) c->fast_buf.assign(4096, 0.0);   // (never resized: the caller keeps views of it for the life of the context)
    *ptr = c->fast_buf.data();
    *bytes = c->fast_buf.size() * sizeof(double);
    return RP_OK;
}

int rp_pack_trajectory(int32_t n, const double *states, double dt, double wheelbase, double x0_orientation, double *out) {
    if (n < 1 || !states || !out || !(dt > 0.0)) return RP_EINVAL;
    const double *x = states + (size_t)RP_X * n, *y = states + (size_t)RP_Y * n, *th = states + (size_t)RP_THETA * n;
    const double *kappa = states + (size_t)RP_KAPPA * n, *s_ = states + (size_t)RP_S * n, *d_ = states + (size_t)RP_D * n;
    const double *sd = states + (size_t)RP_S_DOT * n, *sdd = states + (size_t)RP_S_DDOT * n;
    const double *dd = states + (size_t)RP_D_DOT * n, *ddd = states + (size_t)RP_D_DDOT * n;
    const double lo = x0_orientation - M_PI, hi = x0_orientation + M_PI;
    for (int i = 0; i < n; ++i) {
        double *o = out + (size_t)i * 13;
        o[0] = x[i]; o[1] = y[i]; o[2] = s_[i]; o[3] = d_[i];
        o[4] = s_[i]; o[5] = sd[i]; o[6] = sdd[i];
        o[7] = d_[i]; o[8] = dd[i]; o[9] = ddd[i];
        double t = th[i];   // shift_orientation (utility/general.py:49-55): into [x_0.orientation - pi, x_0.orientation + pi]
        if (t == t && std::fabs(t) < 1e300) {
            while (t < lo) t += 2.0 * M_PI;
            while (t > hi) t -= 2.0 * M_PI;
        }
        o[10] = t;
        o[11] = std::atan2(wheelbase * kappa[i], 1.0);                 // reactive_planner.py:539
        o[12] = i > 0 ? (th[i] - th[i - 1]) / dt : 0.0;               // :535 (state 0 carries x_0.yaw_rate: the caller's)
    }
    return RP_OK;
}

int rp_pyset_order(int32_t n, const double *values, int32_t union_zero, double *out, int32_t *n_out) {
    if (n < 0 || (n && !values) || !out || !n_out) return RP_EINVAL;
    const rpco::PySetF64 base = rpco::PySetF64::from_values(values, n);
    std::vector<double> v;
    const double zero = 0.0;
    if (union_zero) base.union_with(&zero, 1).values(v);
    else base.values(v);
    *n_out = (int32_t)v.size();
    std::memcpy(out, v.data(), v.size() * sizeof(double));   // (at most n + 1 values: the caller's room)
    return RP_OK;
}

int rp_corridor_coeffs(int32_t nT, const double *T, const int32_t *traj_len, const double *v_low, const double *v_up, const int32_t *box_off,
                       const double *boxes, int32_t n_samples, const double *x0_lon, const double *x0_lat, int64_t cap, int64_t *count,
                       double *lon_coeffs, double *lat_coeffs, double *lon_T, int32_t *traj_len_out, double *lon_end, double *lat_end) {
    return rp_corridor_coeffs_grouped(nT, T, traj_len, v_low, v_up, box_off, boxes, n_samples, x0_lon, x0_lat, cap, count, lon_coeffs, lat_coeffs,
                                      lon_T, traj_len_out, lon_end, lat_end, nullptr, nullptr, nullptr);
}

int rp_corridor_coeffs_grouped(int32_t nT, const double *T, const int32_t *traj_len, const double *v_low, const double *v_up,
                               const int32_t *box_off, const double *boxes, int32_t n_samples, const double *x0_lon, const double *x0_lat,
                               int64_t cap, int64_t *count, double *lon_coeffs, double *lat_coeffs, double *lon_T, int32_t *traj_len_out,
                               double *lon_end, double *lat_end, int32_t *group, int32_t *group_first, int64_t *n_groups) {
    if (nT < 0 || n_samples < 1 || !count || !x0_lon || !x0_lat || (nT && (!T || !traj_len || !v_low || !v_up || !box_off))) return RP_EINVAL;
    if (nT && box_off[nT] > 0 && !boxes) return RP_EINVAL;
    // one part per time sample, worked out on the library's host threads (rp_pool.h), put together in the reference's order: a level
    // of 25 536 candidates is ~1 ms of set emulation and coefficient solves on one core -- three times the device's share of the level
    const rpco::Box *bx = reinterpret_cast<const rpco::Box *>(boxes);
    // (the parts keep their room between calls -- a level is ~5 MB of vectors, and memory handed back to the system at the end of
    //  every call comes back as page faults in the next: 2.0 -> 3.8 ms in the build container -- so concurrent callers take turns)
    static std::mutex call_mutex;
    static std::vector<rpco::Candidates> parts;
    std::lock_guard<std::mutex> turn(call_mutex);
    if (parts.size() < (size_t)nT) parts.resize((size_t)nT);
    rppool::Pool &pool = rppool::Pool::get();
    // One pass: a part is worked out, waits for the end offset of the part before it (items are handed out in ascending order, so
    // that part was started earlier), publishes its own and copies itself into the output -- no second wake-up of the pool for
    // the copies.  Parts that would not fit are not copied; *count is the room needed.
    static std::vector<std::atomic<int64_t>> ends, gends;   // end offsets of the parts: candidates, groups
    if (ends.size() < (size_t)nT + 1) {
        std::vector<std::atomic<int64_t>> grown((size_t)nT + 1), ggrown((size_t)nT + 1);
        ends.swap(grown); gends.swap(ggrown);
    }
    for (int k = 0; k <= nT; ++k) { ends[(size_t)k].store(k == 0 ? 0 : -1, std::memory_order_relaxed); gends[(size_t)k].store(0, std::memory_order_relaxed); }
    const bool have_out = lon_coeffs && lat_coeffs && lon_T && traj_len_out && lon_end && lat_end;
    std::atomic<bool> part_failed{false};
    const bool items_ok = pool.parallel_for(nT, [&](int k) {
        rpco::Candidates &c = parts[(size_t)k];
        bool ok = true;
        try {   // (the vectors grow: std::bad_alloc must neither leave a worker thread nor the chain of end offsets unpublished)
            c.lon.clear(); c.lat.clear(); c.T.clear(); c.v_end.clear(); c.d_end.clear(); c.traj_len.clear(); c.group.clear(); c.first.clear();
            rpco::corridor_candidates(k, k + 1, T, traj_len, v_low, v_up, box_off, bx, n_samples, x0_lon, x0_lat, c);
        } catch (...) {
            ok = false;
            part_failed.store(true);
        }
        const size_t m = ok ? c.T.size() : 0, mg = ok ? c.first.size() : 0;
        int64_t at;
        while ((at = ends[(size_t)k].load(std::memory_order_acquire)) < 0) __builtin_ia32_pause();
        const int64_t gat = gends[(size_t)k].load(std::memory_order_relaxed);   // (written before ends[k], read behind it)
        gends[(size_t)k + 1].store(gat + (int64_t)mg, std::memory_order_relaxed);
        ends[(size_t)k + 1].store(at + (int64_t)m, std::memory_order_release);
        if (!m || !have_out || at + (int64_t)m > cap) return;
        if (group) for (size_t i = 0; i < m; ++i) group[(size_t)at + i] = (int32_t)gat + c.group[i];
        if (group_first) for (size_t g = 0; g < mg; ++g) group_first[(size_t)gat + g] = (int32_t)at + c.first[g];   // (groups <= candidates <= cap)
        std::memcpy(lon_coeffs + 6 * (size_t)at, c.lon.data(), sizeof(double) * 6 * m);
        std::memcpy(lat_coeffs + 6 * (size_t)at, c.lat.data(), sizeof(double) * 6 * m);
        std::memcpy(lon_T + at, c.T.data(), sizeof(double) * m);
        std::memcpy(traj_len_out + at, c.traj_len.data(), sizeof(int32_t) * m);
        std::memcpy(lon_end + at, c.v_end.data(), sizeof(double) * m);
        std::memcpy(lat_end + at, c.d_end.data(), sizeof(double) * m);
    });
    const int64_t C = ends[(size_t)nT].load(std::memory_order_acquire);
    if (!items_ok || part_failed.load()) { *count = 0; return RP_ENOMEM; }
    *count = C;
    if (n_groups) *n_groups = gends[(size_t)nT].load(std::memory_order_relaxed);
    if (C > cap || (C && !have_out)) return RP_ENOMEM;   // *count: room needed
    return RP_OK;
}

int rp_build_reference(int32_t n_in, const double *xy_in, int32_t smooth, double resample_step, int32_t cap, int32_t *n_out,
                       double *xy_out, double *ref_pos, double *ref_theta, double *ref_curv, double *ref_curv_d) {
    if (n_in < 2 || !xy_in || !n_out || !(resample_step > 0.0)) return RP_EINVAL;
    std::vector<rpfe::Pt> in((size_t)n_in);
    for (int i = 0; i < n_in; ++i) {
        in[i] = {xy_in[2 * i], xy_in[2 * i + 1]};
        if (!(std::isfinite(in[i].x) && std::isfinite(in[i].y))) return RP_EINVAL;
    }
    rpfe::Tables tb;
    if (rpfe::build_reference(in, smooth != 0, resample_step, tb) != 0) return RP_EINVAL;
    const int n = (int)tb.ref.size();
    *n_out = n;
    if (n > cap || !xy_out || !ref_pos || !ref_theta || !ref_curv || !ref_curv_d) return RP_ENOMEM;   // *n_out tells how much room is needed
    for (int i = 0; i < n; ++i) {
        xy_out[2 * i] = tb.ref[i].x; xy_out[2 * i + 1] = tb.ref[i].y;
        ref_pos[i] = tb.pos[i]; ref_theta[i] = tb.theta[i]; ref_curv[i] = tb.curv[i]; ref_curv_d[i] = tb.curv_d[i];
    }
    return RP_OK;
}

int rp_project(int32_t n, const double *ref_xy, const double *ref_pos, double proj_domain_d_limit, double x, double y, double *s, double *d) {
    if (n < 2 || !ref_xy || !ref_pos || !s || !d) return RP_EINVAL;
    std::vector<rpfe::Pt> ref((size_t)n);
    for (int i = 0; i < n; ++i) ref[i] = {ref_xy[2 * i], ref_xy[2 * i + 1]};
    const std::vector<double> pos(ref_pos, ref_pos + n);
    return rpfe::project(ref, pos, proj_domain_d_limit, x, y, *s, *d) ? RP_OK : RP_EDOMAIN;
}

int rp_initial_state(int32_t n, const double *ref_xy, const double *ref_pos, const double *ref_theta, const double *ref_curv,
                     const double *ref_curv_d, double proj_domain_d_limit, double x, double y, double orientation, double velocity,
                     double acceleration, double steering_angle, double wheelbase, int32_t low_vel_mode, double *x0_lon, double *x0_lat) {
    if (n < 2 || !ref_xy || !ref_pos || !ref_theta || !ref_curv || !ref_curv_d || !x0_lon || !x0_lat || !(wheelbase > 0.0)) return RP_EINVAL;
    rpfe::Tables tb;
    tb.ref.resize((size_t)n);
    for (int i = 0; i < n; ++i) tb.ref[i] = {ref_xy[2 * i], ref_xy[2 * i + 1]};
    tb.pos.assign(ref_pos, ref_pos + n); tb.theta.assign(ref_theta, ref_theta + n);
    tb.curv.assign(ref_curv, ref_curv + n); tb.curv_d.assign(ref_curv_d, ref_curv_d + n);
    double s = 0.0, d = 0.0;
    if (!rpfe::project(tb.ref, tb.pos, proj_domain_d_limit, x, y, s, d)) return RP_EDOMAIN;
    return rpfe::initial_state(tb, s, d, orientation, velocity, acceleration, steering_angle, wheelbase, low_vel_mode != 0, x0_lon, x0_lat) == 0
               ? RP_OK : RP_EDIRECTION;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Intra-node winner exchange through a shared-memory mailbox (include/rp_amd.h).  Slot layout per
// (parity, rank), 64-byte aligned: [seq][7 pad][rp_result][14 n doubles][cnt_seq][cnt][pad].
// Two parities alternate between consecutive exchanges: a rank can run at most one exchange ahead of the
// slowest reader, so the slot it overwrites two exchanges later has been read by everyone.
// ------------------------------------------------------------------------------------------------
namespace {
struct MailboxLayout {
    size_t slot_bytes, result_off, states_off, cnt_off;
};
MailboxLayout mailbox_layout(int n_steps) {
    MailboxLayout l;
    l.result_off = 64;
    l.states_off = l.result_off + ((sizeof(rp_result) + 63) & ~(size_t)63);
    l.cnt_off = l.states_off + ((sizeof(double) * RP_N_ARRAYS * (size_t)n_steps + 63) & ~(size_t)63);
    l.slot_bytes = l.cnt_off + 64;
    return l;
}
inline char *mailbox_slot(void *region, const MailboxLayout &l, int world, uint64_t seq, int rank) {
    return static_cast<char *>(region) + ((size_t)(seq & 1) * (size_t)world + (size_t)rank) * l.slot_bytes;
}
// wait budget per peer (seconds; rp_mailbox_set_timeout, or RP_AMD_MAILBOX_TIMEOUT_S read once) and the rank the last
// failed wait of this thread was spinning on
double g_mailbox_timeout_s = -1.0;
thread_local int t_mailbox_stalled = -1;
double mailbox_timeout_s() {
    if (g_mailbox_timeout_s < 0.0) {
        const char *e = std::getenv("RP_AMD_MAILBOX_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        g_mailbox_timeout_s = v > 0.0 ? v : 30.0;
    }
    return g_mailbox_timeout_s;
}
bool mailbox_wait(void *region, const MailboxLayout &l, int world, uint64_t seq, size_t off) {
    const auto budget = std::chrono::duration<double>(mailbox_timeout_s());
    t_mailbox_stalled = -1;
    for (int r = 0; r < world; ++r) {
        const auto t0 = std::chrono::steady_clock::now();   // every peer gets the whole budget
        const uint64_t *w = reinterpret_cast<const uint64_t *>(mailbox_slot(region, l, world, seq, r) + off);
        for (unsigned spins = 0; __atomic_load_n(w, __ATOMIC_ACQUIRE) != seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > budget) {
                t_mailbox_stalled = r;
                return false;
            }
        }
    }
    return true;
}
}  // namespace

extern "C" {

size_t rp_mailbox_bytes(int32_t world, int32_t n_steps) {
    if (world < 1 || n_steps < 1) return 0;
    return 2 * (size_t)world * mailbox_layout(n_steps).slot_bytes;
}

int rp_mailbox_exchange(void *region, int32_t world, int32_t rank, uint64_t seq, int32_t n_steps, const rp_result *local,
                        const double *local_best_states, rp_result *global, double *global_best_states, int32_t *owner_rank) {
    if (!region || !local || !global || world < 1 || rank < 0 || rank >= world || n_steps < 1 || seq == 0) return RP_EINVAL;
    const MailboxLayout l = mailbox_layout(n_steps);
    const size_t sbytes = sizeof(double) * RP_N_ARRAYS * (size_t)n_steps;
    char *mine = mailbox_slot(region, l, world, seq, rank);
    std::memcpy(mine + l.result_off, local, sizeof(rp_result));
    if (local->best_index >= 0 && local_best_states) std::memcpy(mine + l.states_off, local_best_states, sbytes);
    __atomic_store_n(reinterpret_cast<uint64_t *>(mine), seq, __ATOMIC_RELEASE);
    if (!mailbox_wait(region, l, world, seq, 0)) return RP_ESTATE;
    if (local->struct_size != sizeof(rp_result) || global->struct_size != sizeof(rp_result)) return RP_EABI;
    rp_result g;
    std::memset(&g, 0, sizeof(g));
    g.struct_size = (uint32_t)sizeof(rp_result);
    g.best_index = -1;
    g.best_cost = std::nan("");
    g.best_lat_T = std::nan("");
    for (int k = 0; k < 6; ++k) g.best_lon_coeffs[k] = g.best_lat_coeffs[k] = std::nan("");
    int owner = -1;
    for (int r = 0; r < world; ++r) {
        rp_result q;
        std::memcpy(&q, mailbox_slot(region, l, world, seq, r) + l.result_off, sizeof(q));
        g.n_candidates += q.n_candidates;
        g.n_feasible += q.n_feasible;
        g.n_collision += q.n_collision;
        for (int k = 0; k < 8; ++k) g.reason_counts[k] += q.reason_counts[k];
        if (q.best_index >= 0 && (owner < 0 || q.best_cost < g.best_cost || (q.best_cost == g.best_cost && q.best_index < g.best_index))) {
            owner = r;
            g.best_index = q.best_index;
            g.best_cost = q.best_cost;
            std::memcpy(g.best_lon_coeffs, q.best_lon_coeffs, sizeof(g.best_lon_coeffs));
            std::memcpy(g.best_lat_coeffs, q.best_lat_coeffs, sizeof(g.best_lat_coeffs));
            g.best_lat_T = q.best_lat_T;
        }
    }
    g.kernel_ms = local->kernel_ms;
    *global = g;
    if (owner >= 0 && global_best_states) std::memcpy(global_best_states, mailbox_slot(region, l, world, seq, owner) + l.states_off, sbytes);
    if (owner_rank) *owner_rank = owner;
    return RP_OK;
}

int rp_mailbox_set_timeout(double seconds) {
    if (!(seconds > 0.0)) return RP_EINVAL;
    g_mailbox_timeout_s = seconds;
    return RP_OK;
}

int rp_mailbox_stalled_rank(void) { return t_mailbox_stalled; }

int rp_mailbox_sum(void *region, int32_t world, int32_t rank, uint64_t seq, int32_t n_steps, int64_t value, int64_t *total) {
    if (!region || !total || world < 1 || rank < 0 || rank >= world || n_steps < 1 || seq == 0) return RP_EINVAL;
    const MailboxLayout l = mailbox_layout(n_steps);
    char *mine = mailbox_slot(region, l, world, seq, rank);
    *reinterpret_cast<int64_t *>(mine + l.cnt_off + 8) = value;
    __atomic_store_n(reinterpret_cast<uint64_t *>(mine + l.cnt_off), seq, __ATOMIC_RELEASE);
    if (!mailbox_wait(region, l, world, seq, l.cnt_off)) return RP_ESTATE;
    int64_t t = 0;
    for (int r = 0; r < world; ++r) t += *reinterpret_cast<const int64_t *>(mailbox_slot(region, l, world, seq, r) + l.cnt_off + 8);
    *total = t;
    return RP_OK;
}

}  // extern "C"
