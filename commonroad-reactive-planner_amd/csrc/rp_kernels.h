// rp_kernels.h -- the kernels of one replanning step.  Per candidate they replace the loop body of
// ReactivePlanner._check_kinematics (commonroad_rp/reactive_planner.py:731-960) together with
// sampling (sampling.py:218-241), cost (cost_function.py:51-71) and the collision query
// (reactive_planner.py:1031-1046); citations inline.
//
//   rp_lon_kernel       one group of lanes per (T, longitudinal sample) PAIR: everything of a step that
//                       does not depend on the lateral sample d -- s, s_dot, s_ddot, the reference-path
//                       segment lookup and the quantities interpolated on it (theta_ref, k_r, k_r',
//                       foot point, unit normal) -- written once as a "longitudinal profile" and shared
//                       by the nD candidates of the pair (the reference recomputes it nD times).
//   rp_eval_kernel      one group of lanes per CANDIDATE: lateral polynomial, Frenet -> Cartesian,
//                       constraints, horizon extension, cost, collision, state rows, block partial.
//   rp_finalize_kernel  selection epilogue.
#pragma once

#include "rp_device.h"

#define RP_BLOCK 256  // 4 wavefronts per workgroup
// linear copy-out of the LDS-staged state rows: non-temporal (streaming) 16-byte stores -- the rows are written once
// and never read by the kernels again; measured cfg5 draw 3.59 -> 2.88 ms, cfg4 2.15 -> 1.87 ms
#ifndef RP_COPY_OUT_STORE
#define RP_COPY_OUT_STORE(p, v)                                                                                   \
    __builtin_nontemporal_store(*reinterpret_cast<const __attribute__((ext_vector_type(2))) double *>(&(v)),        \
                                reinterpret_cast<__attribute__((ext_vector_type(2))) double *>(p))
#endif
#ifndef RP_WRITE_THROUGH
#ifndef RP_BRANCHFREE_CONSTRAINTS
#define RP_BRANCHFREE_CONSTRAINTS 1
#endif
#define RP_WRITE_THROUGH 1   // single-launch variant: agent-scope write-through stores of the state rows (see st_row)
#endif
#ifndef RP_LATE_KERNARGS
#define RP_LATE_KERNARGS 1   // single-launch variant: kernel arguments of the step loop are loaded after the prologue
#endif
#ifndef RP_WAVES_PER_SIMD
#define RP_WAVES_PER_SIMD 3  // register budget of the evaluation kernel: 512 / 3 -> 168 VGPRs.  One wavefront issues
                             // an instruction only every ~10-13 cycles (profiles/r01_instruction_costs.txt), so the
                             // vector ALU needs >= 3 resident wavefronts per SIMD to fill; 4 would spill to scratch
#endif

// Everything the evaluation kernel needs; passed by value (kernarg segment).
struct KArgs {
    // problem
    double dt;
    int32_t N, factor, time_step0, low_vel_mode, lon_mode;
    uint32_t constraint_mask, flags;
    double x0_lon[3], x0_lat[3], x0_orientation;
    double wheelbase, wb_rear_axle, half_length, half_width, ego_radius;
    double a_max, v_switch, v_delta_max, kappa_max;
    double c_yaw;   // 1e5 / dt
    double c_kdot;  // dt * v_delta_max / wheelbase
    // cost
    int32_t cost_kind, has_speed, has_s, partials_cap;   // partials_cap: slots per array of the block partials (see Partials)
    double w_a, desired_speed, desired_d, desired_s;
    // grids (device pointers) or explicit polynomials
    int32_t nT, nL, nD, grids_inline;   // grids_inline: [T | L | D | traj_len] travel in the kernarg segment (KArgsG), not in device memory
    const double *T;
    const int32_t *traj_len;
    const double *L;
    const double *D;
    const double *lon_coeffs;  // [C][6]  (COEFFS_IN)
    const double *lat_coeffs;  // [C][6]
    const int32_t *traj_len_c; // [C]
    // COEFFS_IN, optional: candidates that share their longitudinal polynomial (all lateral samples of one (T, v) sample of a corridor
    // level) form a group -- one longitudinal profile per group instead of one per candidate.  pair_of[C]: group of a candidate
    // (ascending, adjacent); group_first[groups]: its first candidate, whose lon_coeffs / traj_len_c rows stand for the group.
    const int32_t *pair_of;
    const int32_t *group_first;
    // candidate range: global indices [cand_begin, cand_begin + count)
    int64_t cand_begin, count;
    const int64_t *single_index;  // != nullptr: evaluate exactly this one (global) candidate -> slot 0
    // list mode (a round of the cost-ordered collision stage; two-kernel path, one wavefront per candidate): slot s evaluates the
    // global candidate index_list[s] for s < min(*list_count, list_cap); status goes to the candidate's own slot of `status`, the
    // cost there (written by the first pass) is what the block partial carries, state rows go to slot s of `states`
    const int32_t *index_list;
    const uint32_t *list_count;
    int32_t list_cap;
    int32_t tail_split;           // 0: state rows of row_stride doubles each.  M > 0 (a multiple of 16, M < N + 1 <= M + 8; two-kernel path, 16 lanes
                                  // per candidate): rows hold the first M steps (row_stride == M), the last N + 1 - M steps of rows 2q and
                                  // 2q + 1 share 128-byte line q behind the rows -- see state_offset
    struct LazyCtl *lazy_ctl;     // first pass of the cost-ordered collision stage: workgroup 0 clears the stage's control block
    // Bounded collision sweep (rp_host.hip: run_sweep): pass 1 -- costs without the query -- opens the bound (workgroup 0 stores ~0 to
    // *sweep_init); the sweep -- the eager 16-lane kernel over the whole batch -- evaluates only candidates that are FEASIBLE by pass 1 and
    // whose cost key is <= *sweep_bound, and every candidate it finds free lowers the bound (atomic min of its cost key)
    unsigned long long *sweep_init;
    unsigned long long *sweep_bound;
    int32_t partials_first, pad_sweep_;   // first slot of this launch's block partials (the sweep's and its pass 1's share one allocation)
    // tables
    const double *tables;  // [TB_ROWS][n_ref]
    int32_t n_ref, search_iters, n_buckets, table_words;
    double bucket_inv_h;
    double proj_d_limit;
    // single-launch variant: the part of the table block a launch can touch (rp_host.hip: table_window), staged instead of the
    // whole block; win_n == 0: the whole block.  Items whose s lies in [win_s_lo, win_s_hi) read staged entries only; any other
    // item makes its workgroup stage the whole block and start over.
    int32_t win_k0, win_n, win_shift, win_b0, win_nb;   // vertices [win_k0, win_k0 + win_n), win_n = 1 << win_shift; bucket entries [win_b0, win_b0 + win_nb)
    int32_t publish_grids;   // rp_lon_kernel: workgroup 0 copies the grids of its kernarg segment to T (device memory) for the kernels behind it
    double win_s_lo, win_s_hi;
    double pos_first, pos_last;   // ref_pos[0], ref_pos[n_ref - 1]
    ObsTables obs;         // obstacle table descriptor, by value: a pointer to a device copy costs a dependent
                           // memory round trip before the first obstacle row can be requested
    int32_t has_obstacles, row_stride;   // row_stride: doubles between the rows of `states` (>= N + 1; rp_host.hip: state_row_stride)
    // longitudinal profiles (rp_lon_kernel -> rp_eval_kernel): pairs [pair_begin, pair_begin + pair_count)
    double *profile;            // [pair_count][PF_FIELDS][N+1]
    struct PairHdr *pair_hdr;   // [pair_count]
    int64_t pair_begin, pair_count;
    int32_t lds_pairs, use_near_mask;   // LON_FUSED: capacity (pairs) of the workgroup's LDS profile area | PF_NEAR masks are bounds
    double lat_abs_d;           // max(|d0|, max |D|): largest lateral offset a lateral polynomial starts or ends at
    double lat_dmin, lat_dmax;  // range of the lateral samples D
    // outputs
    uint32_t *status;  // [count]
    double *cost;      // [count]
    double *states;    // [count][14][row_stride] (MAT) or nullptr
    double *coeffs;    // [count][13] or nullptr (lon 6, lat 6, lat_T)
    void *partials;    // block partials (Partials, partials_cap slots per array; slot = blockIdx.x) or nullptr
    unsigned long long *debug;      // diagnostic build (-DRP_STAMPS) only: s_memtime stamps of block 0 / wave 0
    unsigned long long *host_seq;   // winner re-evaluation only: completion ticket in the host mirror
    unsigned long long seq_value;
    // Chain of sampling levels on one stream (rp_plan_levels: the level loop of plan(), reactive_planner.py:616-636, without a host
    // round trip per level): `gate` is a device word the epilogue of the level that finds a winner sets to (gate_seq << 8) | its
    // number (1, 2, ...); the kernels of every LATER level find it set and leave at once.  nullptr: a plan on its own.
    const unsigned long long *gate;
    unsigned long long gate_seq;
    int32_t gate_level;       // number (1, 2, ...) of the level this launch belongs to
    int32_t ticket_if_none;   // winner re-evaluation: hand the completion ticket over also when there is no winner (not in a chain's earlier levels)
};

// true: an earlier level of the chain this launch belongs to has found its winner -- nothing of this level is wanted any more
__device__ __forceinline__ bool level_gate_closed(const unsigned long long *gate, unsigned long long seq, int level) {
    if (!gate) return false;
    const unsigned long long g = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (g >> 8) == seq && (int)(g & 0xffull) < level;
}

// What the evaluation and profile kernels take: the KArgs block and, behind it, room for the sample grids of the launch
// [T (nT doubles) | L (nL) | D (nD) | traj_len (nT int32)].  A replanning cycle changes the grids every time (the velocity
// samples follow the current speed, the current offset joins the lateral samples), and staging ~0.5 KB through a
// host-to-device copy put a 5-us blit kernel in front of every step; the kernarg segment is written by the launch anyway.
// Grids that do not fit (RP_GRID_INLINE doubles) are staged in device memory as before (KArgs.T / L / D / traj_len): those are the
// large batches, where a 5-us copy does not show; the room is kept small because the launch cost of a kernel grows with its
// kernarg segment (measured: 3.8 KB instead of 0.6 KB put ~1 us between the dispatch and the first wavefront).
#define RP_GRID_INLINE 96
struct KArgsG {
    KArgs k;
    double grid[RP_GRID_INLINE];
};
static_assert(sizeof(KArgsG) <= 4096 - 256, "kernarg segment: at most 4 KB including the hidden arguments");
// rp_lon_kernel -- the first launch of the two-kernel path of a large batch -- takes room for larger grids (cfg3: 126 doubles +
// 31 int32, cfg5: 301 + 99): a 100-us step does not feel the microsecond a 3.8-KB kernarg segment adds to its first launch, a 5-us
// host-to-device copy of the grids in front of every step it does.  Its workgroup 0 also writes them to device memory
// (KArgs.publish_grids), where the kernels behind it -- whose launches stay small -- read them.
constexpr int RP_GRID_INLINE_L = (int)((4096 - 256 - sizeof(KArgs)) / 8);   // what is left of a 4-KB segment behind the hidden arguments
static_assert(RP_GRID_INLINE_L >= 360, "cfg5's grids (301 doubles + 99 int32) ride in rp_lon_kernel's launch block");
struct KArgsGL {
    KArgs k;
    double grid[RP_GRID_INLINE_L];
};
static_assert(sizeof(KArgsGL) <= 4096 - 256, "kernarg segment: at most 4 KB including the hidden arguments");

// base of the launch's grids: [T | L | D | traj_len]
__device__ __forceinline__ const double *grid_base(const KArgs &a) {
    typedef const char __attribute__((address_space(4))) *kchar;
    typedef const double __attribute__((address_space(1))) *gdouble;
    const gdouble inl = (gdouble)(uintptr_t)((kchar)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(KArgs));
    const gdouble staged = (gdouble)(uintptr_t)a.T;
    return (const double *)(a.grids_inline ? inl : staged);
}

// fields of one step of a longitudinal profile
enum { PF_S = 0, PF_SD, PF_SDD, PF_INV_SD, PF_TH_REF, PF_KR, PF_KRD, PF_PX, PF_PY, PF_NX, PF_NY, PF_INDOM, PF_NEAR, PF_NEAR_S, PF_COS_REF, PF_SIN_REF,
       PF_FIELDS };
#define PF_STRIDE (PF_FIELDS + 1)   // item-major profile rows in LDS: [pair][step][PF_STRIDE] (see load_profile_item)
// PF_COS_REF / PF_SIN_REF: cos / sin of theta_ref -- with cos / sin of theta_cl known algebraically (atan branch) the heading's
// cos / sin (ego rectangle of the collision query, direction of the horizon extension) need no transcendental per candidate
// PF_NEAR: bit pattern of a 64-bit mask, bit j = dynamic obstacle j can touch SOME candidate of the pair at this step
// (bit 63: one of the obstacles 63, 64, ... can); PF_NEAR_S: the same for the clusters of static shapes; see near_mask_step

struct PairHdr {
    double lat_T;        // delta_tau of the lateral polynomials of this pair (sampling.py:229-237)
    double s0;           // s at t = 0 (low-velocity mode samples the lateral motion over s - s0)
    int32_t pre_reason;  // pre-filter verdict of reactive_planner.py:796-805 (RP_REASON_* or 0)
    int32_t L;           // number of valid steps (traj_len), clamped to [1, N+1]
};

// Where element (row, step i) of a candidate's state block lies inside the block in device memory, and the block's size.
//   rows of `ns` doubles (tail == 0): N + 1 rounded up to whole 128-byte lines where rows are stored straight to memory -- every
//       store instruction of a 16-lane group then writes whole lines, at the price of the padding: 112 doubles for the 101 steps of
//       N = 100, 11 % of the HBM stream of cfg4 / cfg5;
//   split tail (tail == M): the step blocks that are full, [14][M], then the partial one -- at most 8 steps -- two rows to a line:
//       line q = row 2q (doubles 0 .. 7) | row 2q + 1 (doubles 8 .. 15).  N = 100: 14 * 96 + 7 * 16 = 1 456 doubles for 1 414.
// rp_fetch_states / the result block hand out compact [14][N + 1] blocks either way.
__host__ __device__ __forceinline__ int state_block_doubles(int ns, int tail) { return tail ? RP_N_ARRAYS * tail + (RP_N_ARRAYS / 2) * 16 : RP_N_ARRAYS * ns; }
__host__ __device__ __forceinline__ int state_offset(int row, int i, int ns, int tail) {
    if (tail == 0 || i < tail) return row * ns + i;
    return RP_N_ARRAYS * tail + (row >> 1) * 16 + (row & 1) * 8 + (i - tail);
}

// One partial per workgroup of the evaluation kernel: its best (cost, index) and its counters.  Three arrays in one allocation
// (cap slots each) instead of an array of structs: the selection epilogue of a large batch runs in many workgroups, each of which
// reads ALL (cost, index) pairs -- 16 contiguous bytes per partial -- but only its own slice of the counters.
//   cost [cap] f64 | index [cap] i64 (global index, -1 none) | counters [cap][RP_PARTIAL_CNT] u32: n_feasible, n_collision, reasons[8]
#define RP_PARTIAL_CNT 10
struct Partials {
    double *cost;
    long long *idx;
    uint32_t *cnt;
};
__host__ __device__ __forceinline__ Partials partials_at(void *base, int cap, int first = 0) {
    Partials p;
    p.cost = reinterpret_cast<double *>(base) + first;
    p.idx = reinterpret_cast<long long *>(reinterpret_cast<double *>(base) + cap) + first;
    p.cnt = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(base) + 2 * (size_t)cap) + (size_t)first * RP_PARTIAL_CNT;
    return p;
}
#define RP_PARTIAL_BYTES (16 + 4 * RP_PARTIAL_CNT)   // per slot

// ------------------------------------------------------------------------------------------------
// Cost-ordered ("lazy") collision stage of production-mode plans on large batches -- the reference's own order of work
// (ReactivePlanner._check_collisions, reactive_planner.py:1031-1062: candidates in ascending cost, stop at the first one that is
// free; TrajectoryBundle.sort, trajectories.py:502-510):
//   pass 1   rp_eval_kernel without the collision query: kinematics + cost of every candidate;
//   gather   rp_lazy_hist_kernel + rp_lazy_gather_kernel: three cost thresholds e_0 < e_1 < e_2 and, for each, the list of the
//            feasible candidates with e_(l-1) <= cost < e_l (exact threshold sets, any order);
//   round l  rp_eval_kernel in list mode over list l WITH the query (state rows kept), rp_finalize_kernel picks the cheapest free
//            candidate of the list: every cheaper feasible candidate is in lists 0 .. l and has been found colliding, so it is the
//            winner, and infeasible_count_collision = candidates of the earlier rounds + colliding ones before it in this list.
//            No free candidate: next round; lists exhausted, or a list over its capacity: the eager kernel decides (rp_host.hip).
// Thresholds: a histogram of the costs over bins that are fine next to the cheapest cost c* and coarse far from it -- bin of a
// cost c = position of k(c) - k(c*) on a logarithmic scale with 16 steps per octave, k = the order-preserving integer image of a
// double (cost_key), so that a bin is an exact integer interval of keys -- built by every workgroup of rp_lazy_hist_kernel over its
// slice of the batch; the last workgroup to arrive takes the prefix sums and the first bin edges behind 128, 1 024 and 8 192
// candidates.  rp_lazy_gather_kernel appends every feasible candidate below an edge to the list of its level.
// ------------------------------------------------------------------------------------------------
#define RP_LAZY_LEVELS 3
#define RP_LAZY_BINS 1024
struct LazyCtl {   // 20 eight-byte words, cleared by pass 1
    unsigned long long totals[RP_PARTIAL_CNT];   // counters of pass 1 (n_feasible, -, reasons[8]), summed by the histogram kernel's workgroups
    uint32_t count[RP_LAZY_LEVELS];              // candidates appended to the list of each level
    uint32_t overflow;                           // bit l: the candidates of level l do not fit its list (the level is not conclusive)
    uint32_t checked, found;                     // candidates of the rounds run so far / colliding ones among them
    unsigned long long edge[RP_LAZY_LEVELS];     // level l: feasible candidates with edge[l-1] <= k(cost) - key_min < edge[l]
    unsigned long long key_min;                  // k(c*), c* = cheapest feasible cost of the batch
    uint32_t expect[RP_LAZY_LEVELS], pad_[3];    // candidates of each level by the histogram
};
static_assert(sizeof(LazyCtl) == 160, "LazyCtl layout");

struct GatherArgs {
    const uint32_t *status;      // [count]
    const double *cost;          // [count]
    void *partials;              // Partials of pass 1
    int32_t *lists;              // list of level l at lists + off[l], cap[l] entries (global candidate indices)
    LazyCtl *ctl;
    uint32_t *hist;              // [RP_LAZY_BINS + 1] global histogram + arrival ticket, zero between launches
    int64_t count, cand_begin;
    int32_t partials_cap, partials_first, n_partials, pad_;
    int32_t cap[RP_LAZY_LEVELS], off[RP_LAZY_LEVELS];
    uint32_t target[RP_LAZY_LEVELS], pad2_;   // cumulative number of candidates each level is meant to reach
};

// The 28 eight-byte words of rp_result behind its struct_size header (include/rp_amd.h): what the kernels write.  The result blocks
// in device and pinned host memory hold these; the host copies them into the caller's rp_result (rp_host.hip: result_from_core).
struct ResultCore {
    int64_t best_index;
    double best_cost;
    int64_t n_candidates, n_feasible, n_collision_before_best, n_collision;
    int64_t reason_counts[8];
    double best_lon_coeffs[6], best_lat_coeffs[6];
    double best_lat_T, kernel_ms;
};
static_assert(sizeof(ResultCore) == 28 * 8 && sizeof(rp_result) == 8 + sizeof(ResultCore) && offsetof(rp_result, best_index) == 8 &&
              offsetof(rp_result, kernel_ms) == 8 + offsetof(ResultCore, kernel_ms), "rp_result = struct_size header + ResultCore");

// In-kernel stamps (diagnostic build only; see cdna_hip_programming.md section 7).  Values go to a
// buffer nothing else reads.
#ifdef RP_STAMPS
#ifndef RP_STAMP_BLOCK
#define RP_STAMP_BLOCK 0
#endif
#ifndef RP_STAMP_THREAD
#define RP_STAMP_THREAD 0
#endif
#define RP_STAMP(k)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (a.debug && !a.single_index && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) a.debug[(k)] = t_;                   \
    } while (0)
#else
#define RP_STAMP(k) do { } while (0)
#endif
// Per-block timeline (diagnostic build -DRP_TIMELINE): start / end of every workgroup on the 100 MHz constant clock
// (s_memrealtime: the same counter on every XCD, unlike s_memtime).
#ifdef RP_TIMELINE
#define RP_TL(slot)                                                                               \
    do {                                                                                          \
        if (a.debug && !a.single_index && threadIdx.x == 0) {                                     \
            unsigned long long t_;                                                                \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            a.debug[32 + 2 * blockIdx.x + (slot)] = t_;                                           \
        }                                                                                         \
    } while (0)
#else
#define RP_TL(slot) do { } while (0)
#endif

// stamps of rp_lon_kernel (diagnostic build -DRP_STAMPS without -DRP_TIMELINE: slots 32 .. of the debug buffer; one wavefront)
#if defined(RP_STAMPS) && !defined(RP_TIMELINE)
#define RP_LSTAMP(k)                                                                             \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (a.debug && !a.single_index && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) a.debug[32 + (k)] = t_;  \
    } while (0)
#else
#define RP_LSTAMP(k) do { } while (0)
#endif

// The kernarg segment (528 bytes of KArgs + the hidden launch arguments) is written by the host right before
// the launch and is cold in every cache.  The compiler fetches its fields in many small scalar loads spread
// over the prologue, each first touch of a 64-byte line a full memory round trip in series with the others.
// One batch of loads that touches every line up front turns these into a single round trip; the scattered
// loads that follow hit the scalar cache.
template <int LINES>
__device__ __forceinline__ void touch_kernargs() {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    static_assert(LINES >= 1 && LINES <= 10, "kernarg lines");
    unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm volatile(
        "s_load_dword %0, %10, 0x0\n\t"
        "s_load_dword %1, %10, %11\n\t"
        "s_load_dword %2, %10, %12\n\t"
        "s_load_dword %3, %10, %13\n\t"
        "s_load_dword %4, %10, %14\n\t"
        "s_load_dword %5, %10, %15\n\t"
        "s_load_dword %6, %10, %16\n\t"
        "s_load_dword %7, %10, %17\n\t"
        "s_load_dword %8, %10, %18\n\t"
        "s_load_dword %9, %10, %19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)
        : "s"(kp), "n"(LINES > 1 ? 0x40 : 0), "n"(LINES > 2 ? 0x80 : 0), "n"(LINES > 3 ? 0xc0 : 0), "n"(LINES > 4 ? 0x100 : 0),
          "n"(LINES > 5 ? 0x140 : 0), "n"(LINES > 6 ? 0x180 : 0), "n"(LINES > 7 ? 0x1c0 : 0), "n"(LINES > 8 ? 0x200 : 0),
          "n"(LINES > 9 ? 0x240 : 0)
        : "memory");
}

// The same for `lines` more 64-byte lines from byte offset `first` on (the inline grids behind the KArgs block): scalar
// loads whose results nobody reads -- they pull the lines into L2 / the scalar cache while the prologue computes indices,
// so that the per-lane vector loads of grid values that follow do not go to memory one by one.
__device__ __forceinline__ void touch_kernarg_lines(int first, int lines) {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    for (int k = 0; k < lines; k += 4) {   // (wave-uniform trip count; the last group may touch up to 3 lines beyond: same 4 KB segment)
        unsigned t0, t1, t2, t3;
        const int off = first + k * 64;
        asm volatile(
            "s_load_dword %0, %4, %5\n\t"
            "s_load_dword %1, %4, %6\n\t"
            "s_load_dword %2, %4, %7\n\t"
            "s_load_dword %3, %4, %8"
            : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3)
            : "s"(kp), "s"(off), "s"(off + 64), "s"(off + 128), "s"(off + 192)
            : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ bool better(double c, int64_t i, double bc, int64_t bi) {
    return bi < 0 || c < bc || (c == bc && i < bi);
}

// ------------------------------------------------------------------------------------------------
// The fused kernel.  G lanes per candidate; MAT: write the 14 state rows; COEFFS_IN: polynomials
// are given; LDS_TABLES: reference-path tables staged in LDS (else read through L1/L2).
//
// Arithmetic notes (all within the 1e-6 state tolerance, typically ~1e-13):
//   * theta_cl = atan(d') on the moving branch, hence cos(theta_cl) = 1/sqrt(1 + d'^2) and
//     tan(theta_cl) = d' without a second and third transcendental; only standstill lanes
//     (reactive_planner.py:864-873) evaluate sin/cos, behind a wave-uniform branch.
//   * divisions by quantities reused several times (s_dot, 1 - k_r d, segment length) become one
//     Newton-refined reciprocal each; threshold tests are rearranged to multiplications.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// One-launch epilogue for batches up to RP_FINALIZE_MAX candidates: reduce the block partials,
// count the colliding samples that precede the winner (lazy semantics of _check_collisions),
// fetch the winner's state block from the materialised states (if any) and its polynomial
// coefficients, and write the whole result block to device memory AND to the pinned host mirror
// (no separate memset / count / copy operations on the stream).
// ------------------------------------------------------------------------------------------------
#ifndef RP_FIN_THREADS
#define RP_FIN_THREADS 512   // (256: the count of colliding candidates before the winner took four dependent trips of eight loads on cfg2-sized batches; profiles/r04_fin_threads_ab.txt)
#endif
#define RP_FINALIZE_MAX (1 << 14)   // above: the count of colliding samples before the winner runs as its own many-workgroup
                                    // kernel (one workgroup walking 60 000 status words took 139 us on cfg3)

#define RP_WSTATUS_ROWS_ON_HOST 0x100u   // FinalizeOut.w_status: the winner's state rows are NOT behind this header in device memory
                                         // (non-materialising plans re-evaluate the winner straight into the host mirror)
struct FinalizeOut {          // layout shared with the host (rp_host.hip: ResultBlock)
    ResultCore r;
    unsigned long long n_before;
    uint32_t w_status, pad_;
    double w_cost;
    double w_coeffs[13];
    unsigned long long seq;   // completion ticket: written to the host mirror after everything else
    // followed by best_states[14][n]
};

// lexicographic (cost, index) minimum over a wavefront, result in every lane.  DPP row shifts / broadcasts (plain
// VALU) instead of a ds_bpermute butterfly: 24 LDS-crossbar round trips in series were ~3 000 cycles.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void min_pair_step(double &c, long long &i) {
    int cl = __double2loint(c), ch = __double2hiint(c);
    int il = (int)(unsigned int)((unsigned long long)i & 0xffffffffull), ih = (int)(unsigned int)((unsigned long long)i >> 32);
    const int ocl = __builtin_amdgcn_update_dpp(cl, cl, CTRL, ROW_MASK, 0xf, false);   // lanes without a source keep their own
    const int och = __builtin_amdgcn_update_dpp(ch, ch, CTRL, ROW_MASK, 0xf, false);
    const int oil = __builtin_amdgcn_update_dpp(il, il, CTRL, ROW_MASK, 0xf, false);
    const int oih = __builtin_amdgcn_update_dpp(ih, ih, CTRL, ROW_MASK, 0xf, false);
    const double oc = __hiloint2double(och, ocl);
    const long long oi = (long long)(((unsigned long long)(unsigned int)oih << 32) | (unsigned long long)(unsigned int)oil);
    if (oi >= 0 && better(oc, (int64_t)oi, c, (int64_t)i)) { c = oc; i = oi; }
}
__device__ __forceinline__ void wave_min_pair(double &c, long long &i) {
    min_pair_step<DPP_ROW_SHR1, 0xf>(c, i);
    min_pair_step<DPP_ROW_SHR2, 0xf>(c, i);
    min_pair_step<DPP_ROW_SHR4, 0xf>(c, i);
    min_pair_step<DPP_ROW_SHR8, 0xf>(c, i);        // lane 15 of every row: minimum of the row
    min_pair_step<DPP_ROW_BCAST15, 0xa>(c, i);     // rows 1, 3 also see row 0, 2
    min_pair_step<DPP_ROW_BCAST31, 0xc>(c, i);     // rows 2, 3 also see rows 0 + 1: lane 63 holds the wavefront's minimum
    const int cl = __builtin_amdgcn_readlane(__double2loint(c), 63), ch = __builtin_amdgcn_readlane(__double2hiint(c), 63);
    const unsigned int il = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)i & 0xffffffffull), 63);
    const unsigned int ih = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)i >> 32), 63);
    c = __hiloint2double(ch, cl);
    i = (long long)(((unsigned long long)ih << 32) | il);
}

// Arguments of rp_finalize_kernel: the handful of KArgs fields it reads (the whole 584-byte block by value was nine cold
// kernarg lines to fetch for one workgroup that lives 4 us).
struct FinArgs {
    const uint32_t *status;      // [count]
    const double *cost;          // [count]
    const double *states;        // [count][14][N+1] or nullptr
    void *partials;              // Partials of partials_cap slots per array, read from slot partials_first on
    FinalizeOut *dev_out, *host_out;
    int64_t count, cand_begin;
    unsigned long long seq;      // completion ticket for the host (0: none)
    int32_t N, n_partials, count_inline, copy_states;
    int32_t row_stride;          // doubles between the rows of `states`
    int32_t tail_split;          // layout of the state blocks (state_offset)
    uint32_t inv_n;              // floor(2^32 / (N + 1)) + 1: k / (N + 1) == umulhi(k, inv_n) for k < 14 (N + 1)
    int32_t partials_cap, partials_first;
    unsigned long long *debug;   // diagnostic build (-DRP_STAMPS): s_memtime stamps of the phases, slots 26..31
    unsigned long long *scratch; // rp_select_kernel: [RP_SEL_SCRATCH] zero between launches (counter totals, count, arrival ticket)
    // a round of the cost-ordered collision stage (rp_finalize_kernel; see LazyCtl): the candidates of the round, its control block
    const int32_t *list;
    const uint32_t *list_count;
    struct LazyCtl *lazy;
    int32_t list_cap, level;
    // a level of a chain (see KArgs::gate): the device word, the chain's sequence number, this level's number, and whether it is the
    // chain's last level (which hands the completion ticket over whatever it found)
    unsigned long long *gate;
    unsigned long long gate_seq;
    int32_t gate_level, gate_last;
    // epilogue of a bounded sweep (rp_select_kernel): the winner comes from the sweep's partials (`partials`), the counters of the batch
    // from its pass 1's -- slots cnt_first .. cnt_first + n_cnt_partials - 1 of the same allocation (n_cnt_partials == 0: one set of partials)
    int32_t cnt_first, n_cnt_partials;
};
#ifdef RP_STAMPS
#define RP_FSTAMP(k)                                                                             \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (a.debug && threadIdx.x == 0) a.debug[(k)] = t_;                                      \
    } while (0)
#else
#define RP_FSTAMP(k) do { } while (0)
#endif

// order-preserving integer image of a double (sign flip): integer atomics then order costs
__device__ __forceinline__ unsigned long long cost_key(double c) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(c);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// system-scope store into the pinned host mirror of the result block: past the L2, no write-back needed afterwards
__device__ __forceinline__ void result_host_store(void *p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Result header, one 8-byte word per lane (lanes 0 .. 43 of the calling workgroup), to the device block and the host mirror:
// rp_result (28 words) | n_before | w_status, pad | w_cost | w_coeffs[13] (coefficients, lat_T: filled in by the host)
__device__ __forceinline__ void store_result_header(FinalizeOut *dev_out, FinalizeOut *host_out, int tid, int64_t widx, double wcost, int64_t count,
                                                    const unsigned int *c32 /* [10] n_feasible, n_collision, reasons[8] */,
                                                    unsigned long long before, bool rows_here,
                                                    const unsigned long long *extra = nullptr /* [4]: words 31 .. 34 (w_coeffs[0..3]) */,
                                                    int level_tag = 0) {
    constexpr int words = (int)(offsetof(FinalizeOut, seq) / 8);
    static_assert(words == 44 && sizeof(ResultCore) == 28 * 8, "FinalizeOut layout");
    if (tid < words) {
        const int k = tid;
        const bool have = widx >= 0;
        const unsigned long long nanb = 0x7ff8000000000000ull;
        const unsigned long long costb = have ? (unsigned long long)__double_as_longlong(wcost) : nanb;
        const int ci = k == 3 ? 0 : (k == 5 ? 1 : k - 4);   // counter behind words 3, 5, 6..13
        unsigned long long v = nanb;
        if (k == 0) v = (unsigned long long)widx;
        else if (k == 1 || k == 30) v = costb;
        else if (k == 2) v = (unsigned long long)count;
        else if (k == 3 || k == 5 || (k >= 6 && k < 14)) v = (unsigned long long)c32[ci];
        else if (k == 4 || k == 28) v = before;
        else if (k == 27) v = 0ull;                                  // kernel_ms
        else if (k == 29) v = (have ? (unsigned long long)(RP_LABEL_FEASIBLE | (rows_here ? 0u : RP_WSTATUS_ROWS_ON_HOST)) : 0ull) |
                              ((unsigned long long)(unsigned int)level_tag << 32);   // w_status | pad (pad: number of the chain's level, 0 outside a chain)
        else if (extra && k >= 31 && k < 35) v = extra[k - 31];
        reinterpret_cast<unsigned long long *>(dev_out)[k] = v;
        result_host_store(reinterpret_cast<unsigned long long *>(host_out) + k, v);
    }
}

// sum of a 32-bit value over the lanes of a wavefront (DPP adds), valid in lane 63
__device__ __forceinline__ unsigned int wave_sum_u32_lane63(unsigned int t) {
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR1, 0xf, 0xf, true);
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR2, 0xf, 0xf, true);
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR4, 0xf, 0xf, true);
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR8, 0xf, 0xf, true);      // lane 15 of every row: row total
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST15, 0xa, 0xf, true);   // rows 1, 3 += row 0, 2
    t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST31, 0xc, 0xf, true);   // lane 63: wavefront total
    return t;
}

// Body of rp_finalize_kernel (blockDim.x == RP_FIN_THREADS).  One workgroup that lives a few microseconds: what it costs is
// (a) memory round trips -- the winner's state rows are requested as soon as the winner is known and travel while the
// colliding candidates before it are counted -- and (b) the LENGTH of its instruction chains (a lone wavefront issues an
// instruction every ~10 cycles): only the (cost, index) reduction stands between the partials and the request for the rows,
// the ten counters are summed while the rows travel; the result header is produced one word per lane instead of by one
// thread; the host copy of the result leaves through system-scope stores followed by a plain wait -- a system-scope
// release fence also writes the L2 back, 1.1 us for nothing here.  In-kernel stamps before / after:
// profiles/r02_finalize_stamps.txt.
__device__ __forceinline__ void finalize_body(const FinArgs &a) {
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;   // (uniform) an earlier level of the chain has delivered
    __shared__ unsigned long long sh_before;
    __shared__ unsigned int sh_c32[10];
    const int tid = threadIdx.x;
    const int n_partials = a.n_partials;
    RP_FSTAMP(24);
    double bc = 0.0;
    long long bi = -1;
    unsigned int cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (a thread holds a few partials of < 2^23 candidates each)
    const Partials pp = partials_at(a.partials, a.partials_cap, a.partials_first);
    for (int k = tid; k < n_partials; k += RP_FIN_THREADS) {
        const double pc = pp.cost[k];
        const long long pi = pp.idx[k];
        const uint2 *pq = reinterpret_cast<const uint2 *>(pp.cnt + (size_t)k * RP_PARTIAL_CNT);   // (40-byte rows: 8-byte aligned)
        const uint2 q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3], q4 = pq[4];
        if (pi >= 0 && better(pc, (int64_t)pi, bc, (int64_t)bi)) { bc = pc; bi = pi; }
        cnt[0] += q0.x; cnt[1] += q0.y; cnt[2] += q1.x; cnt[3] += q1.y; cnt[4] += q2.x;
        cnt[5] += q2.y; cnt[6] += q3.x; cnt[7] += q3.y; cnt[8] += q4.x; cnt[9] += q4.y;
    }
    if (tid == 0) sh_before = 0;
    if (tid < 10) sh_c32[tid] = 0;
    RP_FSTAMP(25);     // partials have arrived
    // the winner first -- its state rows are the next round trip; the counters are reduced while the rows travel.
    // (LDS atomics on one address serialise lane by lane: 2 816 of them for the counters alone took 17 k cycles.)
    wave_min_pair(bc, bi);
    __shared__ double sh_cost[RP_FIN_THREADS / 64];
    __shared__ long long sh_idx[RP_FIN_THREADS / 64];
    if ((tid & 63) == 0) { sh_cost[tid >> 6] = bc; sh_idx[tid >> 6] = bi; }
    __syncthreads();
    double wcost = sh_cost[0];
    long long widx_ = sh_idx[0];
#pragma unroll
    for (int w = 1; w < RP_FIN_THREADS / 64; ++w)
        if (sh_idx[w] >= 0 && better(sh_cost[w], (int64_t)sh_idx[w], wcost, (int64_t)widx_)) { wcost = sh_cost[w]; widx_ = sh_idx[w]; }
    const int64_t widx = (int64_t)widx_;
    const int n = a.N + 1;
    // the winner's state rows are requested now: their round trip overlaps the counter sums and the count below
    constexpr int kRowsPerThread = 4;   // covers 14 (N+1) <= 1024 doubles in registers; longer blocks loop at the end
    double wrow[kRowsPerThread];
    const bool want_rows = a.copy_states && widx >= 0;
    const int ns = a.row_stride;
    const bool lazy = a.list != nullptr;   // a round of the cost-ordered collision stage: the candidates are the round's list, their
                                           // state rows lie by list slot (found in the count loop below, requested behind it)
    __shared__ int sh_wslot;
    if (tid == 0) sh_wslot = 0;
    const int blk = state_block_doubles(ns, a.tail_split);
    const double *wsrc = (want_rows && !lazy) ? a.states + (size_t)(widx - a.cand_begin) * (size_t)blk : nullptr;
    auto padded = [&](int k) -> int {   // element k of the compact [14][N + 1] block inside the block in device memory
        const int row = (int)__umulhi((uint32_t)k, a.inv_n);
        return state_offset(row, k - row * n, ns, a.tail_split);
    };
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
        const int k = tid + q * RP_FIN_THREADS;
        wrow[q] = (want_rows && !lazy && k < RP_N_ARRAYS * n) ? wsrc[padded(k)] : 0.0;
    }
    // counters: wavefront sums through 32-bit DPP adds, one LDS add per wavefront and counter
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned int t = cnt[r];
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR1, 0xf, 0xf, true);
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR2, 0xf, 0xf, true);
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR4, 0xf, 0xf, true);
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR8, 0xf, 0xf, true);      // lane 15 of every row: row total
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST15, 0xa, 0xf, true);   // rows 1, 3 += row 0, 2
        t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST31, 0xc, 0xf, true);   // lane 63: wavefront total
        if ((tid & 63) == 63 && t) atomicAdd(&sh_c32[r], t);
    }
    __syncthreads();
    RP_FSTAMP(26);   // winner known, rows requested
    // colliding feasible samples that sort before the winner (reactive_planner.py:1031-1046)
    int lcount = 0;
    if (lazy) { const uint32_t lc = *a.list_count; lcount = (int)(lc < (uint32_t)a.list_cap ? lc : (uint32_t)a.list_cap); }
    const int ccount = lazy ? lcount : (a.count_inline ? (int)a.count : 0);   // count_inline: at most RP_FINALIZE_MAX
    if (ccount > 0 && (lazy || sh_c32[1] > 0)) {
        // status and cost of kUnroll candidates per lane are requested together (both unconditionally, 12 B per candidate):
        // one candidate per trip with the cost load behind the label test was a chain of dependent round trips
        // (sixteen: batches of up to 8 192 candidates in ONE trip -- a trip to words the evaluation kernel has just written is ~3 500
        //  cycles, cfg2 + road boundary took two)
        constexpr int kUnroll = 16;
        int nloc = 0;
        for (int i0 = tid; i0 < ccount; i0 += RP_FIN_THREADS * kUnroll) {
            uint32_t st[kUnroll];
            double cs[kUnroll];
            int64_t gi[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int i = i0 + u * RP_FIN_THREADS, ic = i < ccount ? i : ccount - 1;
                const int64_t fs = lazy ? (int64_t)a.list[ic] - a.cand_begin : (int64_t)ic;
                gi[u] = a.cand_begin + fs;
                st[u] = a.status[fs];
                cs[u] = a.cost[fs];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int i = i0 + u * RP_FIN_THREADS;
                const bool coll = i < ccount && RP_STATUS_LABEL(st[u]) == RP_LABEL_INFEASIBLE_COLLISION;
                nloc += (coll && (widx < 0 || cs[u] < wcost || (cs[u] == wcost && gi[u] < widx))) ? 1 : 0;
                if (lazy && i < ccount && gi[u] == widx) sh_wslot = i;   // (list entries are distinct)
            }
        }
        // (one LDS add per wavefront: adds from every lane to the one word serialise lane by lane -- in-kernel stamps on cfg1 / cfg2 +
        //  road boundary, where most lanes have something to add: 9 400 / 7 000 cycles for this block, profiles/r04_finalize_stamps.txt)
        const unsigned int nw = wave_sum_u32_lane63((unsigned int)nloc);
        if ((tid & 63) == 63 && nw) atomicAdd(&sh_before, (unsigned long long)nw);
        __syncthreads();   // (uniform: sh_c32 is final)
    }
    __shared__ unsigned long long sh_extra[4];
    if (lazy) {
        // counters of the batch come from pass 1 (the control block); the collisions are the ones FOUND so far, and the candidates
        // of the earlier rounds -- all colliding -- sort before this round's winner
        LazyCtl *const lz = a.lazy;
        __syncthreads();
        const unsigned int round_coll = sh_c32[1];
        const unsigned int checked0 = lz->checked, found0 = lz->found;
        __syncthreads();
        if (tid < RP_PARTIAL_CNT) sh_c32[tid] = tid == 1 ? found0 + round_coll : (unsigned int)lz->totals[tid];
        if (tid == 0) {
            sh_before += checked0;
            const unsigned int checked1 = checked0 + (unsigned int)lcount, found1 = found0 + round_coll;
            if (widx < 0) { lz->checked = checked1; lz->found = found1; }   // this round failed: its candidates all collide
            sh_extra[0] = (unsigned long long)lz->count[0] | ((unsigned long long)lz->count[1] << 32);
            sh_extra[1] = (unsigned long long)lz->count[2] | ((unsigned long long)lz->overflow << 32);
            sh_extra[2] = (unsigned long long)checked1 | ((unsigned long long)found1 << 32);
            sh_extra[3] = lz->totals[0];
        }
        __syncthreads();
        if (want_rows) {   // the winner's rows, by its slot in the round's list
            wsrc = a.states + (size_t)sh_wslot * (size_t)blk;
#pragma unroll
            for (int q = 0; q < kRowsPerThread; ++q) {
                const int k = tid + q * RP_FIN_THREADS;
                wrow[q] = k < RP_N_ARRAYS * n ? wsrc[padded(k)] : 0.0;
            }
        }
    }
    RP_FSTAMP(27);   // count done
    FinalizeOut *const dev_out = a.dev_out, *const host_out = a.host_out;
    auto host_store = [](void *p, unsigned long long v) { result_host_store(p, v); };
    store_result_header(dev_out, host_out, tid, widx, wcost, a.count, sh_c32, sh_before, want_rows, lazy ? sh_extra : nullptr, a.gate_level);
    RP_FSTAMP(28);   // header stored
    if (want_rows) {   // winner's state block straight from the materialised states
        double *d1 = reinterpret_cast<double *>(dev_out + 1), *d2 = reinterpret_cast<double *>(host_out + 1);
#pragma unroll
        for (int q = 0; q < kRowsPerThread; ++q) {
            const int k = tid + q * RP_FIN_THREADS;
            if (k < RP_N_ARRAYS * n) { d1[k] = wrow[q]; host_store(d2 + k, (unsigned long long)__double_as_longlong(wrow[q])); }
        }
        for (int k = tid + kRowsPerThread * RP_FIN_THREADS; k < RP_N_ARRAYS * n; k += RP_FIN_THREADS) {
            const double v = wsrc[padded(k)];
            d1[k] = v;
            host_store(d2 + k, (unsigned long long)__double_as_longlong(v));
        }
    }
    RP_FSTAMP(29);   // rows stored
    // a level of a chain: a winner closes the gate for the levels behind; without one only the last level reports
    if (a.gate && widx >= 0 && tid == 0)
        __hip_atomic_store(a.gate, (a.gate_seq << 8) | (unsigned long long)a.gate_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.seq && (!a.gate || widx >= 0 || a.gate_last)) {   // completion ticket for the spinning host thread: after every wavefront's result stores have been acknowledged
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's stores have left for the host
        __syncthreads();
        RP_FSTAMP(30);   // stores acknowledged
        if (tid == 0) __hip_atomic_store(&host_out->seq, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(RP_FIN_THREADS) void rp_finalize_kernel(const FinArgs a) {
    touch_kernargs<3>();
    finalize_body(a);
}

// this thread's share of the lexicographic (cost, index) minimum over the block partials k = tid, tid + nthreads, ...: eight
// partials' (cost, index) pairs are requested together -- one pair per trip, with the comparison between the trips, was a chain of
// memory round trips (cfg3: 3 906 partials over 256 threads = 16 trips; rp_select_kernel 17 us against 10.6 us on cfg5's 256)
// (sixteen at a time: cfg3's 3 906 partials over 256 threads in ONE trip)
__device__ __forceinline__ void partials_min(const Partials &pp, int n_partials, int tid, int nthreads, double &bc, long long &bi) {
    constexpr int kBatch = 16;
    for (int k0 = tid; k0 < n_partials; k0 += nthreads * kBatch) {
        double pc[kBatch];
        long long pi[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int k = k0 + u * nthreads, kc = k < n_partials ? k : n_partials - 1;
            pc[u] = pp.cost[kc];
            pi[u] = k < n_partials ? pp.idx[kc] : -1;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (pi[u] >= 0 && better(pc[u], (int64_t)pi[u], bc, (int64_t)bi)) { bc = pc[u]; bi = pi[u]; }
    }
}

// ------------------------------------------------------------------------------------------------
// Selection epilogue of a LARGE batch (more than RP_FINALIZE_MAX candidates): the same result as rp_finalize_kernel, in one
// launch of many workgroups instead of fold -> one-workgroup epilogue -> count kernel -> copy of the count -> stream sync.
//   every workgroup   reduces ALL (cost, index) partials -- 16 contiguous bytes each -- to the winner (the same in all of them),
//                     sums the counters of ITS slice of the partials and counts the colliding candidates of ITS slice of the
//                     batch that sort before the winner (lazy semantics of _check_collisions, reactive_planner.py:1031-1046):
//                     device-scope atomic adds into a[scratch];
//   workgroup 0       also copies the winner's state rows to the result block (device + host mirror);
//   the LAST one      to arrive (ticket = device-scope atomic counter behind a release fence: the "last block" pattern -- no
//                     workgroup ever waits for another) reads the totals, writes the result header and the completion ticket the
//                     host spins on, and clears the scratch words for the next launch.
// ------------------------------------------------------------------------------------------------
#define RP_SEL_THREADS 256
#define RP_SEL_SCRATCH 16   // 8-byte words: [0..9] counter totals | [10] colliding before the winner | [11] arrival ticket
#define RP_SEL_SLICE 2048   // candidates per workgroup (one trip of the count loop: 8 per lane, requested together with the partials)
// What passes between the workgroups are the totals, and those are device-scope atomic adds: complete at the device's point of
// coherence once acknowledged.  A workgroup therefore only WAITS for its adds (s_waitcnt) before it takes its ticket -- no
// release / acquire fences (each an L2 write-back or invalidate, a microsecond or two on a chain that is nothing but round trips).
__global__ __launch_bounds__(RP_SEL_THREADS) void rp_select_kernel(const FinArgs a) {
    touch_kernargs<3>();
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;   // (uniform over the grid) an earlier level of the chain has delivered
    __shared__ double sh_cost[RP_SEL_THREADS / 64];
    __shared__ long long sh_idx[RP_SEL_THREADS / 64];
    __shared__ unsigned int sh_c32[RP_PARTIAL_CNT];
    __shared__ unsigned int sh_nb, sh_ticket;
    __shared__ unsigned long long sh_before;
    const int tid = threadIdx.x, nwg = (int)gridDim.x, w = (int)blockIdx.x;
    const int n_partials = a.n_partials;
    const Partials pp = partials_at(a.partials, a.partials_cap, a.partials_first);
    // -- everything this workgroup reads is requested up front, one round trip: its slice of the batch (status + cost: the count of
    //    the colliding candidates before the winner needs the winner only for the comparison), all (cost, index) partials
    constexpr int kUnroll = RP_SEL_SLICE / RP_SEL_THREADS;
    const int ccount = a.count_inline ? (int)a.count : 0;   // (< 2^31: checked by the host)
    const int per_c = (ccount + nwg - 1) / nwg, i_lo = w * per_c, i_hi = i_lo + per_c < ccount ? i_lo + per_c : ccount;
    uint32_t st[kUnroll];
    double cs[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const int i = i_lo + tid + u * RP_SEL_THREADS, ic = i < i_hi ? i : (i_hi > 0 ? i_hi - 1 : 0);
        st[u] = ccount > 0 ? a.status[ic] : 0u;
        cs[u] = ccount > 0 ? a.cost[ic] : 0.0;
    }
    double bc = 0.0;
    long long bi = -1;
    partials_min(pp, n_partials, tid, RP_SEL_THREADS, bc, bi);
    // counters of this workgroup's slice of the partials (after a bounded sweep: of its pass 1's partials -- the sweep's own count only
    // the candidates it looked at)
    unsigned int cnt[RP_PARTIAL_CNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool swept_totals = a.lazy != nullptr;   // (the sweep took over from exhausted list rounds: the counters are in their control block)
    const bool swept = a.n_cnt_partials > 0 || swept_totals;
    const int n_cp = swept_totals ? 0 : (swept ? a.n_cnt_partials : n_partials);
    const Partials pcn = swept ? partials_at(a.partials, a.partials_cap, a.cnt_first) : pp;
    const int per_p = (n_cp + nwg - 1) / nwg, k0 = w * per_p, k1 = k0 + per_p < n_cp ? k0 + per_p : n_cp;
    for (int k = k0 + tid; k < k1; k += RP_SEL_THREADS) {
        const uint2 *pq = reinterpret_cast<const uint2 *>(pcn.cnt + (size_t)k * RP_PARTIAL_CNT);
        const uint2 q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3], q4 = pq[4];
        cnt[0] += q0.x; cnt[1] += q0.y; cnt[2] += q1.x; cnt[3] += q1.y; cnt[4] += q2.x;
        cnt[5] += q2.y; cnt[6] += q3.x; cnt[7] += q3.y; cnt[8] += q4.x; cnt[9] += q4.y;
    }
    if (tid < RP_PARTIAL_CNT) sh_c32[tid] = 0;
    if (tid == 0) sh_nb = 0;
    // -- the winner: the same in every workgroup
    wave_min_pair(bc, bi);
    if ((tid & 63) == 0) { sh_cost[tid >> 6] = bc; sh_idx[tid >> 6] = bi; }
    __syncthreads();
    double wcost = sh_cost[0];
    long long widx_ = sh_idx[0];
#pragma unroll
    for (int q = 1; q < RP_SEL_THREADS / 64; ++q)
        if (sh_idx[q] >= 0 && better(sh_cost[q], (int64_t)sh_idx[q], wcost, (int64_t)widx_)) { wcost = sh_cost[q]; widx_ = sh_idx[q]; }
    const int64_t widx = (int64_t)widx_;
    const int n = a.N + 1, ns = a.row_stride;
    // -- the winner's state rows travel while the slices are summed -- in EVERY workgroup (a few KB from the L2 each): whichever
    //    arrives last has them at hand and sends rows and header to the host in one batch, with one wait for the acknowledgements
    //    (rows by workgroup 0 ahead of its ticket + header by the last one: two waits on the same chain)
    constexpr int kRowsPerThread = 4;
    double wrow[kRowsPerThread];
    const bool want_rows = a.copy_states && widx >= 0;
    const bool my_rows = want_rows;
    const double *const wsrc = want_rows ? a.states + (size_t)(widx - a.cand_begin) * (size_t)state_block_doubles(ns, a.tail_split) : nullptr;
    auto padded = [&](int k) -> int { const int row = (int)__umulhi((uint32_t)k, a.inv_n); return state_offset(row, k - row * n, ns, a.tail_split); };
#pragma unroll
    for (int q = 0; q < kRowsPerThread; ++q) {
        const int k = tid + q * RP_SEL_THREADS;
        wrow[q] = (my_rows && k < RP_N_ARRAYS * n) ? wsrc[padded(k)] : 0.0;
    }
    if (__any(k0 + tid < k1)) {   // (wave-uniform: wavefronts without a partial skip the sums)
#pragma unroll
        for (int r = 0; r < RP_PARTIAL_CNT; ++r) {
            unsigned int t = cnt[r];
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR1, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR2, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR4, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR8, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST15, 0xa, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST31, 0xc, 0xf, true);
            if ((tid & 63) == 63 && t) atomicAdd(&sh_c32[r], t);
        }
    }
    // -- colliding feasible candidates of this workgroup's slice of the batch that sort before the winner (reactive_planner.py:1031-1046)
    {
        int nloc = 0;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = i_lo + tid + u * RP_SEL_THREADS;
            const bool coll = i < i_hi && RP_STATUS_LABEL(st[u]) == RP_LABEL_INFEASIBLE_COLLISION;
            nloc += (coll && (widx < 0 || cs[u] < wcost || (cs[u] == wcost && a.cand_begin + i < widx))) ? 1 : 0;
        }
        for (int i = i_lo + tid + RP_SEL_SLICE; i < i_hi; i += RP_SEL_THREADS) {   // (slices beyond RP_SEL_SLICE: the grid was capped)
            const bool coll = RP_STATUS_LABEL(a.status[i]) == RP_LABEL_INFEASIBLE_COLLISION;
            const double c = a.cost[i];
            nloc += (coll && (widx < 0 || c < wcost || (c == wcost && a.cand_begin + i < widx))) ? 1 : 0;
        }
        const unsigned int nw = wave_sum_u32_lane63((unsigned int)nloc);   // (one LDS add per wavefront, as in finalize_body)
        if ((tid & 63) == 63 && nw) atomicAdd(&sh_nb, nw);
    }
    __syncthreads();
    if (tid < RP_PARTIAL_CNT && sh_c32[tid])
        __hip_atomic_fetch_add(&a.scratch[tid], (unsigned long long)sh_c32[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == RP_PARTIAL_CNT && sh_nb) __hip_atomic_fetch_add(&a.scratch[10], (unsigned long long)sh_nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FinalizeOut *const dev_out = a.dev_out, *const host_out = a.host_out;
    // -- arrival: this workgroup's adds have been acknowledged before its ticket is taken
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) sh_ticket = (unsigned int)__hip_atomic_fetch_add(&a.scratch[11], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_ticket != (unsigned int)(nwg - 1)) return;   // (workgroup-uniform)
    // -- the last workgroup: totals -> result header -> completion ticket; scratch cleared for the next launch
    if (tid < RP_PARTIAL_CNT) sh_c32[tid] = (unsigned int)__hip_atomic_load(&a.scratch[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == RP_PARTIAL_CNT) sh_before = __hip_atomic_load(&a.scratch[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // (after a bounded sweep "colliding candidates found" is the number in front of the winner: which candidates BEHIND it the sweep
    //  looked at before the bound closed depends on the order its wavefronts ran in -- not a property of the plan)
    if (swept_totals && tid < RP_PARTIAL_CNT) sh_c32[tid] = (unsigned int)a.lazy->totals[tid];
    __syncthreads();
    if (swept && tid == 0) sh_c32[1] = (unsigned int)sh_before;
    __syncthreads();
    if (tid < 12) __hip_atomic_store(&a.scratch[tid], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_result_header(dev_out, host_out, tid, widx, wcost, a.count, sh_c32, sh_before, want_rows, nullptr, a.gate_level);
    if (a.gate && widx >= 0 && tid == 0)
        __hip_atomic_store(a.gate, (a.gate_seq << 8) | (unsigned long long)a.gate_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (want_rows) {
        double *d1 = reinterpret_cast<double *>(dev_out + 1), *d2 = reinterpret_cast<double *>(host_out + 1);
#pragma unroll
        for (int q = 0; q < kRowsPerThread; ++q) {
            const int k = tid + q * RP_SEL_THREADS;
            if (k < RP_N_ARRAYS * n) { d1[k] = wrow[q]; result_host_store(d2 + k, (unsigned long long)__double_as_longlong(wrow[q])); }
        }
        for (int k = tid + kRowsPerThread * RP_SEL_THREADS; k < RP_N_ARRAYS * n; k += RP_SEL_THREADS) {
            const double v = wsrc[padded(k)];
            d1[k] = v;
            result_host_store(d2 + k, (unsigned long long)__double_as_longlong(v));
        }
    }
    if (a.seq && (!a.gate || widx >= 0 || a.gate_last)) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&host_out->seq, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// bin of a key difference d = k(c) - k(c*): d itself below 16, then 16 bins per octave; lazy_bin_lower = smallest d of a bin
__device__ __forceinline__ int lazy_bin(unsigned long long d) {
    if (d < 16ull) return (int)d;
    const int msb = 63 - __clzll((long long)d);
    return (msb - 3) * 16 + (int)((d >> (msb - 4)) & 15ull);
}
__device__ __forceinline__ unsigned long long lazy_bin_lower(int b) {
    if (b < 16) return (unsigned long long)b;
    if (b >= 61 * 16) return ~0ull;   // (behind the last bin)
    return (unsigned long long)(16 + (b & 15)) << (b / 16 - 1);
}

#define RP_GATHER_THREADS 256
// pass 1 -> counters, histogram, thresholds
__global__ __launch_bounds__(RP_GATHER_THREADS) void rp_lazy_hist_kernel(const GatherArgs a) {
    touch_kernargs<2>();
    __shared__ unsigned int sh_hist[RP_LAZY_BINS];
    __shared__ unsigned int sh_c32[RP_PARTIAL_CNT];
    __shared__ double sh_cost[RP_GATHER_THREADS / 64];
    __shared__ long long sh_idx[RP_GATHER_THREADS / 64];
    __shared__ unsigned int sh_ticket;
    const int tid = threadIdx.x, nwg = (int)gridDim.x, w = (int)blockIdx.x;
    const int n_partials = a.n_partials;
    const Partials pp = partials_at(a.partials, a.partials_cap, a.partials_first);
    // -- the cheapest feasible cost (every workgroup, the same) and the counters of this workgroup's slice of the partials
    double bc = 0.0;
    long long bi = -1;
    partials_min(pp, n_partials, tid, RP_GATHER_THREADS, bc, bi);
    unsigned int cnt[RP_PARTIAL_CNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int per_p = (n_partials + nwg - 1) / nwg, k0 = w * per_p, k1 = k0 + per_p < n_partials ? k0 + per_p : n_partials;
    for (int k = k0 + tid; k < k1; k += RP_GATHER_THREADS) {
        const uint2 *pq = reinterpret_cast<const uint2 *>(pp.cnt + (size_t)k * RP_PARTIAL_CNT);
        const uint2 q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3], q4 = pq[4];
        cnt[0] += q0.x; cnt[1] += q0.y; cnt[2] += q1.x; cnt[3] += q1.y; cnt[4] += q2.x;
        cnt[5] += q2.y; cnt[6] += q3.x; cnt[7] += q3.y; cnt[8] += q4.x; cnt[9] += q4.y;
    }
    for (int b = tid; b < RP_LAZY_BINS; b += RP_GATHER_THREADS) sh_hist[b] = 0;
    if (tid < RP_PARTIAL_CNT) sh_c32[tid] = 0;
    wave_min_pair(bc, bi);
    if ((tid & 63) == 0) { sh_cost[tid >> 6] = bc; sh_idx[tid >> 6] = bi; }
    __syncthreads();
    double wcost = sh_cost[0];
    long long widx = sh_idx[0];
#pragma unroll
    for (int q = 1; q < RP_GATHER_THREADS / 64; ++q)
        if (sh_idx[q] >= 0 && better(sh_cost[q], (int64_t)sh_idx[q], wcost, (int64_t)widx)) { wcost = sh_cost[q]; widx = sh_idx[q]; }
    const unsigned long long key_min = widx >= 0 ? cost_key(wcost) : 0ull;
    if (__any(k0 + tid < k1)) {
#pragma unroll
        for (int r = 0; r < RP_PARTIAL_CNT; ++r) {
            unsigned int t = cnt[r];
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR1, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR2, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR4, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_SHR8, 0xf, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST15, 0xa, 0xf, true);
            t += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST31, 0xc, 0xf, true);
            if ((tid & 63) == 63 && t) atomicAdd(&sh_c32[r], t);
        }
    }
    // -- histogram of this workgroup's slice of the batch (eight candidates per lane requested together: one round trip per 2 048)
    if (widx >= 0) {
        const int ccount = (int)a.count;
        const int per_c = (ccount + nwg - 1) / nwg, i_lo = w * per_c, i_hi = i_lo + per_c < ccount ? i_lo + per_c : ccount;
        constexpr int kUnroll = 8;
        for (int i0 = i_lo + tid; i0 < i_hi; i0 += RP_GATHER_THREADS * kUnroll) {
            uint32_t st[kUnroll];
            double cs[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int i = i0 + u * RP_GATHER_THREADS, ic = i < i_hi ? i : i_hi - 1;
                st[u] = a.status[ic];
                cs[u] = a.cost[ic];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (i0 + u * RP_GATHER_THREADS < i_hi && RP_STATUS_LABEL(st[u]) == RP_LABEL_FEASIBLE && cs[u] == cs[u])
                    atomicAdd(&sh_hist[lazy_bin(cost_key(cs[u]) - key_min)], 1u);
        }
    }
    __syncthreads();
    for (int b = tid; b < RP_LAZY_BINS; b += RP_GATHER_THREADS)
        if (sh_hist[b]) __hip_atomic_fetch_add(&a.hist[b], sh_hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < RP_PARTIAL_CNT && sh_c32[tid])
        __hip_atomic_fetch_add(&a.ctl->totals[tid], (unsigned long long)sh_c32[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);   // this workgroup's adds have been acknowledged before it takes its ticket (see rp_select_kernel)
    __syncthreads();
    if (tid == 0) sh_ticket = __hip_atomic_fetch_add(&a.hist[RP_LAZY_BINS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_ticket != (unsigned int)(nwg - 1)) return;
    // -- the last workgroup: prefix sums over the bins, the edges behind the levels' targets; histogram cleared for the next launch
    unsigned int h[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = __hip_atomic_load(&a.hist[4 * tid + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run += h[q];
    }
    __syncthreads();   // (sh_hist is reused)
    sh_hist[tid] = run;
    __syncthreads();
    for (int d = 1; d < RP_GATHER_THREADS; d <<= 1) {   // inclusive scan of the 256 four-bin sums
        const unsigned int v = tid >= d ? sh_hist[tid - d] : 0u;
        __syncthreads();
        sh_hist[tid] += v;
        __syncthreads();
    }
    unsigned int before = tid ? sh_hist[tid - 1] : 0u;   // candidates in the bins ahead of this thread's four
    __shared__ int sh_bin[RP_LAZY_LEVELS];
    __shared__ unsigned int sh_cum[RP_LAZY_LEVELS];
    if (tid < RP_LAZY_LEVELS) { sh_bin[tid] = RP_LAZY_BINS - 1; sh_cum[tid] = sh_hist[RP_GATHER_THREADS - 1]; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned int after = before + h[q];
#pragma unroll
        for (int l = 0; l < RP_LAZY_LEVELS; ++l)
            if (before < a.target[l] && after >= a.target[l]) { sh_bin[l] = 4 * tid + q; sh_cum[l] = after; }   // (exactly one bin per level, if any)
        before = after;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) __hip_atomic_store(&a.hist[4 * tid + q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(&a.hist[RP_LAZY_BINS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        LazyCtl *const z = a.ctl;
        z->key_min = key_min;
        unsigned int prev = 0, ov = 0;
        for (int l = 0; l < RP_LAZY_LEVELS; ++l) {
            z->edge[l] = widx >= 0 ? lazy_bin_lower(sh_bin[l] + 1) : 0ull;
            const unsigned int ex = widx >= 0 ? sh_cum[l] - prev : 0u;
            z->expect[l] = ex;
            if (ex > (unsigned int)a.cap[l]) ov |= 1u << l;
            prev = widx >= 0 ? sh_cum[l] : 0u;
        }
        z->overflow = ov;
    }
}

// thresholds -> the candidate lists of the levels (any order inside a list)
__global__ __launch_bounds__(RP_GATHER_THREADS) void rp_lazy_gather_kernel(const GatherArgs a) {
    touch_kernargs<2>();
    const int tid = threadIdx.x, lane = tid & 63, nwg = (int)gridDim.x, w = (int)blockIdx.x;
    const LazyCtl *const z = a.ctl;
    const unsigned long long key_min = z->key_min, e0 = z->edge[0], e1 = z->edge[1], e2 = z->edge[2];
    const int ccount = (int)a.count;
    const int per_c = (ccount + nwg - 1) / nwg, i_lo = w * per_c, i_hi = i_lo + per_c < ccount ? i_lo + per_c : ccount;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    constexpr int kUnroll = 8;   // eight candidates per lane requested together: one round trip per 2 048 candidates of the slice
    for (int base0 = i_lo + (tid & ~63); base0 < i_hi; base0 += RP_GATHER_THREADS * kUnroll) {   // wave-uniform: ballots inside
        uint32_t stv[kUnroll];
        double csv[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = base0 + u * RP_GATHER_THREADS + lane, ic = i < i_hi ? i : i_hi - 1;
            stv[u] = a.status[ic];
            csv[u] = a.cost[ic];
        }
        // level of each of this lane's eight candidates (3: none)
        int lvl[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = base0 + u * RP_GATHER_THREADS + lane;
            const double c = csv[u];
            const unsigned long long d = cost_key(c) - key_min;
            const bool feas = i < i_hi && RP_STATUS_LABEL(stv[u]) == RP_LABEL_FEASIBLE && c == c && d < e2;
            lvl[u] = !feas ? 3 : (d < e0 ? 0 : (d < e1 ? 1 : 2));
        }
        // one reservation per level and wavefront for all eight rounds of ballots (the good candidates cluster in a few wavefronts:
        // a reservation per ballot was a chain of up to 24 device-scope round trips there -- 18 us for a 10-us kernel)
        unsigned int total[RP_LAZY_LEVELS] = {0, 0, 0};
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int l = 0; l < RP_LAZY_LEVELS; ++l) total[l] += (unsigned int)__popcll(__ballot(lvl[u] == l));
        unsigned int at[RP_LAZY_LEVELS] = {0, 0, 0};
#pragma unroll
        for (int l = 0; l < RP_LAZY_LEVELS; ++l)
            if (total[l] && lane == 0) at[l] = atomicAdd(&a.ctl->count[l], total[l]);   // (the three are in flight together)
#pragma unroll
        for (int l = 0; l < RP_LAZY_LEVELS; ++l) at[l] = (unsigned int)__builtin_amdgcn_readfirstlane((int)at[l]);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = base0 + u * RP_GATHER_THREADS + lane;
#pragma unroll
            for (int l = 0; l < RP_LAZY_LEVELS; ++l) {
                const unsigned long long bm = __ballot(lvl[u] == l);
                const unsigned int pos = at[l] + (unsigned int)__popcll(bm & lt_mask);
                if (lvl[u] == l && pos < (unsigned int)a.cap[l]) a.lists[a.off[l] + (int)pos] = (int32_t)(a.cand_begin + i);
                at[l] += (unsigned int)__popcll(bm);
            }
        }
    }
}

// Frenet -> Cartesian curvature and acceleration (reactive_planner.py:883-896) with the fused multiply-adds SPELLED OUT.  Sums of two
// products -- k_r' d + k_r d', (..) (q q) + q k_r, s'' f + (..)(..) -- can be contracted either way round, and the compiler's choice
// depends on the code around the expression: two kernels that inline the same source line may then differ in the last bit (seen
// between rp_eval_kernel and rp_cost_kernel: 44 of 186 968 costs of cfg4 one ulp apart).  These are the forms rp_eval_kernel has
// always compiled to.
__device__ __forceinline__ double frenet_kterm(double k_r_d, double d, double k_r, double dp) { return __builtin_fma(k_r_d, d, k_r * dp); }
__device__ __forceinline__ double frenet_kappa(double dpp, double kterm, double tanT, double cosT, double q, double k_r) {
    return __builtin_fma(k_r, q, (__builtin_fma(kterm, tanT, dpp) * cosT) * (q * q));
}
// cos / sin of the heading theta_ref + theta_cl from cos / sin of both (sums of two products: spelled out, see above)
__device__ __forceinline__ double heading_cos(double cos_ref, double sin_ref, double cosT, double sinT) { return __builtin_fma(cos_ref, cosT, -(sin_ref * sinT)); }
__device__ __forceinline__ double heading_sin(double cos_ref, double sin_ref, double cosT, double sinT) { return __builtin_fma(sin_ref, cosT, cos_ref * sinT); }
__device__ __forceinline__ double frenet_acc(double sdd, double f, double sd, double secT, double oneKrD, double tanT, double kappa, double k_r,
                                             double kterm) {
    return __builtin_fma(sdd, f, (sd * sd * secT) * __builtin_fma(oneKrD * tanT, __builtin_fma(kappa, f, -k_r), -kterm));
}

// Per-group LDS scratch of the evaluation kernel.  Values that are identical in all lanes of a group
// and live for the whole candidate are parked here and re-read (broadcast reads) where they are used.
//   poly[0..14]  lateral polynomial: c0..c5 | c1, 2c2, 3c3, 4c4, 5c5 | 2c2, 6c3, 12c4, 20c5
//   last[0..15]  x y theta v a kappa kappa_dot s d theta_cl s_dot s_ddot d_dot d_ddot cos(theta) sin(theta)
struct GroupScratch {
    double poly[16];
    double last[16];
};

__device__ __forceinline__ void park_poly(double *o, const Poly &c) {
    o[0] = c.c0; o[1] = c.c1; o[2] = c.c2; o[3] = c.c3; o[4] = c.c4; o[5] = c.c5;
    o[6] = c.c1; o[7] = 2.0 * c.c2; o[8] = 3.0 * c.c3; o[9] = 4.0 * c.c4; o[10] = 5.0 * c.c5;
    o[11] = 2.0 * c.c2; o[12] = 6.0 * c.c3; o[13] = 12.0 * c.c4; o[14] = 20.0 * c.c5;
}
// PolynomialTrajectory.calc_position / calc_velocity / calc_acceleration (polynomial_trajectory.py:240-271), Horner
__device__ __forceinline__ double poly_pos(const double *o, double t) {
    return ((((o[5] * t + o[4]) * t + o[3]) * t + o[2]) * t + o[1]) * t + o[0];
}
__device__ __forceinline__ double poly_vel(const double *o, double t) {
    return (((o[10] * t + o[9]) * t + o[8]) * t + o[7]) * t + o[6];
}
__device__ __forceinline__ double poly_acc(const double *o, double t) { return ((o[14] * t + o[13]) * t + o[12]) * t + o[11]; }

// ------------------------------------------------------------------------------------------------
// Longitudinal profiles.  Everything of a step that does not depend on the lateral sample d.
// ------------------------------------------------------------------------------------------------
struct RefTab {   // rows of the reference-path table block (rp_device.h: TB_*), in LDS or global memory
    const double *pos, *theta, *curv, *curv_d, *x, *y, *tx, *ty, *inv;
    const int *bucket;
    int n_ref;
    double pos_first, pos_last;
};

__device__ __forceinline__ RefTab ref_tab(const double *tab, int n_ref) {
    RefTab r;
    r.pos = tab + TB_POS * n_ref; r.theta = tab + TB_THETA * n_ref; r.curv = tab + TB_CURV * n_ref;
    r.curv_d = tab + TB_CURV_D * n_ref; r.x = tab + TB_X * n_ref; r.y = tab + TB_Y * n_ref;
    r.tx = tab + TB_TX * n_ref; r.ty = tab + TB_TY * n_ref; r.inv = tab + TB_INVLEN * n_ref;
    r.bucket = reinterpret_cast<const int *>(tab + TB_ROWS * n_ref);
    r.n_ref = n_ref;
    r.pos_first = r.pos[0];
    r.pos_last = r.pos[n_ref - 1];
    return r;
}

// the same over a block whose end vertices need not be staged (single-launch variant with a table window)
__device__ __forceinline__ RefTab ref_tab(const double *tab, int n_ref, double pos_first, double pos_last) {
    RefTab r;
    r.pos = tab + TB_POS * n_ref; r.theta = tab + TB_THETA * n_ref; r.curv = tab + TB_CURV * n_ref;
    r.curv_d = tab + TB_CURV_D * n_ref; r.x = tab + TB_X * n_ref; r.y = tab + TB_Y * n_ref;
    r.tx = tab + TB_TX * n_ref; r.ty = tab + TB_TY * n_ref; r.inv = tab + TB_INVLEN * n_ref;
    r.bucket = reinterpret_cast<const int *>(tab + TB_ROWS * n_ref);
    r.n_ref = n_ref;
    r.pos_first = pos_first;
    r.pos_last = pos_last;
    return r;
}

struct LonPair {
    Poly lon;
    double lat_T;   // delta_tau of the pair's lateral polynomials (sampling.py:229-237); 0 with COEFFS_IN
    double T;       // the pair's time sample (grid mode)
    int L;          // traj_len clamped to [1, N+1]
};

// raw inputs of a pair's longitudinal polynomial (loads only, so that they can be requested early)
struct LonPairIn {
    double c[6];   // COEFFS_IN: the coefficients; grid mode: c[0] = T, c[1] = longitudinal sample
    int L;
};

// (T, longitudinal sample) pair of a candidate: its index / nD on a grid, itself for explicit polynomials -- or its group, where the
// caller said which candidates share their longitudinal polynomial (KArgs::pair_of)
template <bool COEFFS_IN, class KA>
__device__ __forceinline__ int64_t pair_index(const KA &a, int64_t gidx) {
    if (COEFFS_IN) return a.pair_of ? (int64_t)a.pair_of[gidx] : gidx;
    return (int64_t)((uint32_t)gidx / (uint32_t)a.nD);
}

template <bool COEFFS_IN>
__device__ __forceinline__ LonPairIn lon_pair_fetch(const KArgs &a, int64_t pair) {
    LonPairIn r;
    if (COEFFS_IN) {
        const int64_t row = a.group_first ? (int64_t)a.group_first[pair] : pair;   // (grouped: the group's first candidate)
        const double *pl = a.lon_coeffs + 6 * row;
#pragma unroll
        for (int k = 0; k < 6; ++k) r.c[k] = pl[k];
        r.L = a.traj_len_c[row];
    } else {
        const int iT = (int)((uint32_t)pair / (uint32_t)a.nL), iL = (int)((uint32_t)pair - (uint32_t)iT * (uint32_t)a.nL);
        const double *gb = grid_base(a);
        r.c[0] = gb[iT];
        r.c[1] = gb[a.nT + iL];
        r.L = reinterpret_cast<const int32_t *>(gb + a.nT + a.nL + a.nD)[iT];
    }
    return r;
}

// longitudinal polynomial of a (T, longitudinal sample) pair: sampling.py:253-266
template <bool COEFFS_IN>
__device__ __forceinline__ LonPair lon_pair_make(const KArgs &a, const LonPairIn &in, int n) {
    LonPair r;
    r.lat_T = 0.0;
    r.T = 0.0;
    if (COEFFS_IN) {
        r.lon = {in.c[0], in.c[1], in.c[2], in.c[3], in.c[4], in.c[5]};
    } else {
        const double T = in.c[0];
        if (a.lon_mode == RP_LON_STOPPING)
            r.lon = quintic_coeffs(a.x0_lon[0], a.x0_lon[1], a.x0_lon[2], in.c[1], 0.0, 0.0, T);  // sampling.py:259-263
        else
            r.lon = quartic_coeffs(a.x0_lon[0], a.x0_lon[1], a.x0_lon[2], T, in.c[1]);             // sampling.py:254-258
        r.lat_T = T;
        r.T = T;
        if (a.low_vel_mode) {  // sampling.py:229-234 (evaluate_state_at_tau clamps nothing here: tau == delta_tau)
            const double sg = r.lon.pos(T) - a.x0_lon[0];
            r.lat_T = sg <= 0.0 ? T : sg;
        }
    }
    r.L = in.L > n ? n : (in.L < 1 ? 1 : in.L);
    return r;
}

template <bool COEFFS_IN>
__device__ __forceinline__ LonPair lon_pair(const KArgs &a, int64_t pair, int n) {
    return lon_pair_make<COEFFS_IN>(a, lon_pair_fetch<COEFFS_IN>(a, pair), n);
}

// step i of a pair's profile -> f[PF_FIELDS]
__device__ __forceinline__ void lon_step(const KArgs &a, const RefTab &rt, const Poly &lon, int i, double *f) {
    const int n_ref = rt.n_ref;
    const double t = (double)i * a.dt;
    // -- polynomial evaluation, reactive_planner.py:751-753,776
    const double s = lon.pos(t), sdd = lon.acc(t);
    double sd = lon.vel(t);
    if (fabs(sd) < RP_EPS) sd = 0.0;
    const bool moving = sd > 0.001;              // :811,:822,:842
    const double inv_sd = moving ? rp_rcp(sd) : 0.0;
    // -- segment lookup + interpolation factors (:835-839); Python's negative index wraps
    const int ub = a.n_buckets > 0
                       ? upper_bound_bucket(rt.pos, rt.bucket, n_ref, a.n_buckets, a.bucket_inv_h, rt.pos_first, rt.pos_last, s)
                       : upper_bound(rt.pos, n_ref, a.search_iters, s);
    const int s_idx = (ub == n_ref) ? -1 : ub - 1;
    const int k0 = s_idx < 0 ? n_ref - 1 : s_idx, k1 = s_idx + 1;
    const double p0 = rt.pos[k0], inv_len = rt.inv[k0];
    const double ds = s - p0;
    const double lam = ds * inv_len;
    const double th0 = rt.theta[k0];
    const double th_ref = make_valid_orientation((rt.theta[k1] - th0) * ds * inv_len + th0);  // interpolate_angle
    const double c0 = rt.curv[k0], cd0 = rt.curv_d[k0];
    const double k_r = (rt.curv[k1] - c0) * lam + c0;              // :876-880
    const double k_r_d = (rt.curv_d[k1] - cd0) * lam + cd0;
    // -- foot point and unit normal of (s, d) -> (x, y) = foot + d * normal, :908-917
    int k = ub - 1;
    k = k < 0 ? 0 : (k > n_ref - 2 ? n_ref - 2 : k);
    const double lam2 = (s - rt.pos[k]) * rt.inv[k];
    const double bx = rt.x[k], by = rt.y[k], ux0 = rt.tx[k], uy0 = rt.ty[k];
    const double px = bx + lam2 * (rt.x[k + 1] - bx), py = by + lam2 * (rt.y[k + 1] - by);
    const double ax = ux0 + lam2 * (rt.tx[k + 1] - ux0), ay = uy0 + lam2 * (rt.ty[k + 1] - uy0);
    const double inv_tn = rp_rsqrt(ax * ax + ay * ay);
    f[PF_S] = s; f[PF_SD] = sd; f[PF_SDD] = sdd; f[PF_INV_SD] = inv_sd; f[PF_TH_REF] = th_ref;
    f[PF_KR] = k_r; f[PF_KRD] = k_r_d; f[PF_PX] = px; f[PF_PY] = py;
    f[PF_NX] = -(ay * inv_tn); f[PF_NY] = ax * inv_tn;
    f[PF_INDOM] = (s >= rt.pos_first && s <= rt.pos_last) ? 1.0 : 0.0;
    rp_sincos(th_ref, &f[PF_SIN_REF], &f[PF_COS_REF]);
}

// The same step, cut into three independent instruction chains (single-launch variant: a lone wavefront issues one
// instruction every ~10 cycles whatever its kind, so the profile rows of a workgroup's one or two pairs -- at most a
// wavefront's worth of items -- cost the LENGTH of lon_step's chain; three wavefronts of the workgroup each run one part for
// every item, the fourth clears the mask rows).  Same expressions as lon_step, field for field.
//   part 0: theta_ref and its cos / sin (+ s)      part 1: s_dot, s_ddot, 1 / s_dot, k_r, k_r'      part 2: foot point, unit normal, in-domain
template <int PART>
__device__ __forceinline__ void lon_step_part(const KArgs &a, const RefTab &rt, const Poly &lon, int i, double s, double *o, size_t n0) {
    const int n_ref = rt.n_ref;
    const int ub = a.n_buckets > 0
                       ? upper_bound_bucket(rt.pos, rt.bucket, n_ref, a.n_buckets, a.bucket_inv_h, rt.pos_first, rt.pos_last, s)
                       : upper_bound(rt.pos, n_ref, a.search_iters, s);
    if (PART == 2) {
        int k = ub - 1;
        k = k < 0 ? 0 : (k > n_ref - 2 ? n_ref - 2 : k);
        const double lam2 = (s - rt.pos[k]) * rt.inv[k];
        const double bx = rt.x[k], by = rt.y[k], ux0 = rt.tx[k], uy0 = rt.ty[k];
        const double px = bx + lam2 * (rt.x[k + 1] - bx), py = by + lam2 * (rt.y[k + 1] - by);
        const double ax = ux0 + lam2 * (rt.tx[k + 1] - ux0), ay = uy0 + lam2 * (rt.ty[k + 1] - uy0);
        const double inv_tn = rp_rsqrt(ax * ax + ay * ay);
        o[PF_PX * n0] = px; o[PF_PY * n0] = py;
        o[PF_NX * n0] = -(ay * inv_tn); o[PF_NY * n0] = ax * inv_tn;
        o[PF_INDOM * n0] = (s >= rt.pos_first && s <= rt.pos_last) ? 1.0 : 0.0;
        return;
    }
    const int s_idx = (ub == n_ref) ? -1 : ub - 1;
    const int k0 = s_idx < 0 ? n_ref - 1 : s_idx, k1 = s_idx + 1;
    const double p0 = rt.pos[k0], inv_len = rt.inv[k0];
    const double ds = s - p0;
    if (PART == 0) {
        const double th0 = rt.theta[k0];
        const double th_ref = make_valid_orientation((rt.theta[k1] - th0) * ds * inv_len + th0);  // interpolate_angle
        o[PF_S * n0] = s; o[PF_TH_REF * n0] = th_ref;
        double sn, cs;
        rp_sincos(th_ref, &sn, &cs);
        o[PF_SIN_REF * n0] = sn; o[PF_COS_REF * n0] = cs;
    } else {
        const double t = (double)i * a.dt;
        const double sdd = lon.acc(t);
        double sd = lon.vel(t);
        if (fabs(sd) < RP_EPS) sd = 0.0;
        const double inv_sd = sd > 0.001 ? rp_rcp(sd) : 0.0;
        const double lam = ds * inv_len;
        const double c0 = rt.curv[k0], cd0 = rt.curv_d[k0];
        o[PF_SD * n0] = sd; o[PF_SDD * n0] = sdd; o[PF_INV_SD * n0] = inv_sd;
        o[PF_KR * n0] = (rt.curv[k1] - c0) * lam + c0;              // :876-880
        o[PF_KRD * n0] = (rt.curv_d[k1] - cd0) * lam + cd0;
    }
}

// ------------------------------------------------------------------------------------------------
// Broad phase of the collision query, once per (pair, step) instead of once per (candidate, step, obstacle).
// All nD candidates of a pair are, at step i, inside a circle that follows from the pair's profile alone:
//   valid steps (i < L):  rear axle = foot_i + d n_i with |d| <= lat_abs_d + 0.2 |d0'| T + 0.0173 |d0''| T^2
//       (quintic Hermite basis: d = d1 + (d0 - d1) A + d0' T B + d0'' T^2 C, 0 <= A <= 1, |B| <= 0.196, |C| <= 0.01728);
//   extended steps (i >= L, trajectories.py:168-197): rear axle = last valid point + c_i (cos, sin)(theta_last) with
//       theta_last = theta_ref(s_last) (the lateral polynomial ends with d' = 0), c_i monotone in (v_last, a_last),
//       which are linear in the lateral sample d at the last valid step.
// The ego rectangle adds wb_rear_axle + ego_radius.  Obstacles whose bounding circle at this step misses that
// circle cannot touch any candidate of the pair; the others get their bit.  Where the assumptions do not hold
// (explicit polynomials, T not on the time grid, standstill at the last valid step, lateral variable outside
// [0, lat_T]) every bit is set, i.e. nothing is culled.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mask_as_double(uint64_t m) { return __longlong_as_double((long long)m); }
__device__ __forceinline__ uint64_t double_as_mask(double d) { return (uint64_t)__double_as_longlong(d); }

// what pair_step_bound reads of the pair, of step i and of the last valid step L - 1
struct BoundIn {
    int L;
    double T, lat_T, s0;                                   // pair
    double s_i, px_i, py_i;                                // step i (used when i < L)
    double sd1, sdd1, kr1, krd1, px1, py1, cs1, sn1, inv_sd1;   // step L - 1 (used when i >= L); cs1, sn1 = cos / sin theta_ref
};

// circle (cx, cy, R) that contains the ego rectangle of every candidate of the pair at step i; false: no bound
__device__ __forceinline__ bool pair_step_bound(const KArgs &a, int i, const BoundIn &in, double &cx, double &cy, double &R) {
    const int L = in.L;
    if (!(a.lat_abs_d < 1e300)) return false;                                   // NaN / inf lateral samples
    if (!(fabs((double)(L - 1) * a.dt - in.T) <= 1e-9 * in.T)) return false;   // T is not the last valid step
    const double wb = fabs(a.wb_rear_axle);
    if (i < L) {
        const double tau = a.low_vel_mode ? in.s_i - in.s0 : (double)i * a.dt;
        if (!(tau >= 0.0 && tau <= in.lat_T * (1.0 + 1e-12))) return false;
        const double dmax = a.lat_abs_d + 0.2 * fabs(a.x0_lat[1]) * in.lat_T + 0.0173 * fabs(a.x0_lat[2]) * in.lat_T * in.lat_T;
        cx = in.px_i; cy = in.py_i;
        R = dmax + wb + a.ego_radius;
    } else {
        const bool moving = in.inv_sd1 > 0.0 || a.low_vel_mode;
        if (!moving) return false;   // orientation of the last valid step is carried, not theta_ref
        const double sd = in.sd1, sdd = in.sdd1, kr = in.kr1, krd = in.krd1;
        const double d0 = a.lat_dmin, d1 = a.lat_dmax;
        const double v0 = sd * (1.0 - kr * d0), v1 = sd * (1.0 - kr * d1);                       // v = s' (1 - k_r d) / cos(0)
        const double a0 = sdd * (1.0 - kr * d0) - sd * sd * krd * d0, a1 = sdd * (1.0 - kr * d1) - sd * sd * krd * d1;
        const int m = i - L + 1;
        const double c_lo = ext_travel(fmin(v0, v1), fmin(a0, a1), m, a.dt) * (1.0 - 1e-9);
        const double c_hi = ext_travel(fmax(v0, v1), fmax(a0, a1), m, a.dt) * (1.0 + 1e-9);
        const double along = 0.5 * (c_lo + c_hi) + wb;
        cx = in.px1 + along * in.cs1; cy = in.py1 + along * in.sn1;   // (cos / sin of theta_ref(L - 1): profile fields)
        R = 0.5 * (c_hi - c_lo) + fmax(fabs(d0), fabs(d1)) + a.ego_radius;
    }
    R = R * (1.0 + 1e-9) + 1e-6;
    return cx == cx && cy == cy && R == R;   // (NaN foot point: no bound)
}

// the same from the profile fields of step i (and of step L - 1, recomputed, for extended steps)
template <bool COEFFS_IN>
__device__ __forceinline__ bool pair_step_bound_fields(const KArgs &a, const RefTab &rt, const LonPair &lp, int i, const double *f,
                                                       double &cx, double &cy, double &R) {
    if (COEFFS_IN) return false;
    BoundIn in;
    in.L = lp.L; in.T = lp.T; in.lat_T = lp.lat_T; in.s0 = lp.lon.c0;
    in.s_i = f[PF_S]; in.px_i = f[PF_PX]; in.py_i = f[PF_PY];
    in.sd1 = in.sdd1 = in.kr1 = in.krd1 = in.px1 = in.py1 = in.cs1 = in.sn1 = in.inv_sd1 = 0.0;
    if (i >= lp.L) {
        double g[PF_FIELDS];
        lon_step(a, rt, lp.lon, lp.L - 1, g);
        in.sd1 = g[PF_SD]; in.sdd1 = g[PF_SDD]; in.kr1 = g[PF_KR]; in.krd1 = g[PF_KRD];
        in.px1 = g[PF_PX]; in.py1 = g[PF_PY]; in.cs1 = g[PF_COS_REF]; in.sn1 = g[PF_SIN_REF]; in.inv_sd1 = g[PF_INV_SD];
    }
    return pair_step_bound(a, i, in, cx, cy, R);
}

// can static shape (kind, idx) reach the circle (cx, cy, R)?  Rectangles by their two slabs (tight for the long thin
// strips of a road boundary, where a bounding circle says nothing), triangles and circles by their bounding circle
__device__ __forceinline__ bool static_shape_near(const ObsTables &ob, int kind, int idx, double cx, double cy, double R) {
    if (kind == 0) {
        const gcdouble o = (gcdouble)ob.sobb + (size_t)idx * OB_ROW;
        const Obb b = {o[OB_CX], o[OB_CY], o[OB_UX], o[OB_UY], o[OB_HL], o[OB_HW]};
        return !obb_slab_far(b, cx, cy, R);
    }
    const gcdouble o = kind == 1 ? (gcdouble)ob.tri + (size_t)idx * 10 + 6 : (gcdouble)ob.circ + (size_t)idx * 4;
    const double ex = o[0] - cx, ey = o[1] - cy, er = R + o[2];
    return ex * ex + ey * ey <= er * er * 1.000001;
}

// f[PF_NEAR], f[PF_NEAR_S] of step i
// `last`: the profile fields of the pair's last valid step L - 1 when the caller has them (rp_lon_kernel: handed on from the
// lane that worked that step out); else they are worked out again for every extended step (lon_step is ~450 instructions).
#ifndef RP_LON_BROAD_BATCH
#define RP_LON_BROAD_BATCH 8
#endif
template <bool COEFFS_IN>
__device__ __forceinline__ void near_mask_step(const KArgs &a, const RefTab &rt, const LonPair &lp, int i, double *f,
                                               const double *last = nullptr) {
    f[PF_NEAR] = f[PF_NEAR_S] = mask_as_double(0);
    if (COEFFS_IN || !a.use_near_mask || !a.has_obstacles) return;   // masks unused: every shape is tested
    const ObsTables &ob = a.obs;
    double cx = 0.0, cy = 0.0, R = 0.0;
    bool bounded;
    if (last != nullptr && i >= lp.L) {
        BoundIn in;
        in.L = lp.L; in.T = lp.T; in.lat_T = lp.lat_T; in.s0 = lp.lon.c0;
        in.s_i = f[PF_S]; in.px_i = f[PF_PX]; in.py_i = f[PF_PY];
        in.sd1 = last[PF_SD]; in.sdd1 = last[PF_SDD]; in.kr1 = last[PF_KR]; in.krd1 = last[PF_KRD];
        in.px1 = last[PF_PX]; in.py1 = last[PF_PY]; in.cs1 = last[PF_COS_REF]; in.sn1 = last[PF_SIN_REF]; in.inv_sd1 = last[PF_INV_SD];
        bounded = pair_step_bound(a, i, in, cx, cy, R);
    } else {
        bounded = pair_step_bound_fields<COEFFS_IN>(a, rt, lp, i, f, cx, cy, R);
    }
    const int k = a.time_step0 + i * a.factor - ob.dyn_t0;
    if (ob.n_dyn > 0 && k >= 0 && k < ob.n_steps) {   // (outside: no dynamic obstacle exists at this scenario step)
        uint64_t m = ~0ull;
        if (bounded) {
            const gcdouble dyn = (gcdouble)ob.dyn;
            m = 0;
            // One 16-byte load per obstacle (its centre at this step; dyn_xy_offset) and its radius as a scalar (the obstacle's largest
            // over the steps, dyn_rmax_offset: a superset of the per-step test), eight obstacles requested before the first is tested.
            // Round 2 read cx, cy and r_bound from three planes under a `#pragma unroll 8`, for which the compiler emitted load,
            // load, load, wait per obstacle; with the loads batched the phase stayed at ~150 cycles per load instruction -- the
            // 153 vector loads per wavefront (8 wavefronts per CU) are what it costs, not their latency (in-kernel stamps on cfg3:
            // 37 k cycles of the kernel's 48 k, then 22 k batched)
            typedef double dbl2 __attribute__((ext_vector_type(2)));
            typedef const dbl2 __attribute__((address_space(4))) *gcdouble2;
            const gcdouble2 xy = (gcdouble2)(dyn + dyn_xy_offset(ob.n_dyn, ob.n_steps));
            const gcdouble rmax = dyn + dyn_rmax_offset(ob.n_dyn, ob.n_steps);
            constexpr int kB = RP_LON_BROAD_BATCH;
            for (int j0 = 0; j0 < ob.n_dyn; j0 += kB) {
                dbl2 oc[kB];
#pragma unroll
                for (int u = 0; u < kB; ++u) {   // (a batch's tail repeats the last obstacle: same circle, same bit)
                    const int j = j0 + u < ob.n_dyn ? j0 + u : ob.n_dyn - 1;
                    oc[u] = xy[(size_t)j * ob.n_steps + k];
                }
#pragma unroll
                for (int u = 0; u < kB; ++u) {
                    const int j = j0 + u < ob.n_dyn ? j0 + u : ob.n_dyn - 1;
                    const double dx = oc[u].x - cx, dy = oc[u].y - cy, rr = R + rmax[j];   // NaN centre: absent, no bit
                    m |= (uint64_t)(dx * dx + dy * dy <= rr * rr * 1.000001) << (j < 63 ? j : 63);
                }
            }
        }
        f[PF_NEAR] = mask_as_double(m);
    }
    // (static shapes: no (pair, step) mask -- every pose looks its own cell of the grid over them up, rp_device.h: static_grid_mask)
}

// `n16` 16-byte pieces from global memory into the workgroup's LDS, four per thread requested before the first is stored, the
// stores through an explicit LDS pointer.  Written as `dst[k] = src[k]` under `#pragma unroll 4` with dst a plain pointer, the
// compiler emitted load, wait, flat_store per piece: a generic store may alias the next load, so the four round trips of a
// thread came one after the other -- and flat stores count on the vector-memory counter besides.
// (Measured on cfg2 / cfg2 + road boundary: no change of the kernel's duration -- the other wavefronts of the workgroup cover
// these round trips; kept because it is what the source says it does.)
template <int NTHREADS = RP_BLOCK, int BATCH = 4>
__device__ __forceinline__ void copy16_to_lds(const double *src, double *dst_lds, int n16, int tid) {
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef dbl2 __attribute__((address_space(3))) *lds_dbl2;
    const dbl2 *s = reinterpret_cast<const dbl2 *>(src);
    const lds_dbl2 d = (lds_dbl2) reinterpret_cast<dbl2 *>(dst_lds);
    for (int k0 = tid; k0 < n16; k0 += BATCH * NTHREADS) {
        dbl2 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) { const int k = k0 + u * NTHREADS; v[u] = s[k < n16 ? k : n16 - 1]; }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) { const int k = k0 + u * NTHREADS; if (k < n16) d[k] = v[u]; }
    }
}

// One group of G lanes per (T, longitudinal sample) pair, lane = time step: the profile is written once
// and shared by the nD candidates of the pair (the reference recomputes it nD times).
template <int G, bool COEFFS_IN, bool LDS_TABLES>
#ifndef RP_LON_WAVES
#define RP_LON_WAVES 1
#endif
__global__ __launch_bounds__(RP_BLOCK, RP_LON_WAVES) void rp_lon_kernel(const KArgsGL ag) {
    const KArgs &a = ag.k;
    extern __shared__ double lds[];
    RP_LSTAMP(0);
    touch_kernargs<10>();
    RP_LSTAMP(1);
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;   // (uniform) an earlier level of the chain has delivered
    const int tid = threadIdx.x;
    const int n_ref = a.n_ref;
    if (a.publish_grids && blockIdx.x == 0) {   // the grids of this launch's kernarg segment -> device memory, for the kernels behind
        const int words = a.nT + a.nL + a.nD + ((a.nT + 1) >> 1);   // [T | L | D | traj_len (int32)]
        const double *src = grid_base(a);
        double *dst = const_cast<double *>(a.T);
        for (int k = tid; k < words; k += RP_BLOCK) dst[k] = src[k];
    }
    const double *tab;
    if (LDS_TABLES) {   // 16-byte loads, four in flight per lane
        copy16_to_lds(a.tables, lds, a.table_words >> 1, tid);   // table_words is even
        tab = lds;
        RP_LSTAMP(2);
        __syncthreads();
        RP_LSTAMP(3);
    } else {
        tab = a.tables;
    }
    const RefTab rt = ref_tab(tab, n_ref);

    const int lane = tid & 63;
    const int gl = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    constexpr int GPB = RP_BLOCK / G;
    const int N = a.N, n = N + 1;
    const int nchunks = (n + G - 1) / G;
    const bool draw = (a.flags & RP_FLAG_DRAW_ALL) != 0;

    for (int64_t slot = (int64_t)blockIdx.x * GPB + tid / G; slot < a.pair_count; slot += (int64_t)gridDim.x * GPB) {
        // all lanes of a group share the slot; groups of one wave may leave the loop at different trip counts,
        // which is fine here: the body uses only group-level ballots on lanes that are all still active.
        const LonPair lp = lon_pair<COEFFS_IN>(a, a.pair_begin + slot, n);
        const int L = lp.L;
        RP_LSTAMP(4);
        double *const prow = a.profile + ((size_t)slot * PF_FIELDS) * (size_t)n;
        bool bad_a = false, bad_v = false;
        // fields of the last valid step L - 1 that the bound of the EXTENDED steps needs (pair_step_bound): handed on by the lane
        // that works that step out, in the step block that holds it -- every later step of the pair lies in this or a later block
        double last[PF_FIELDS];
#pragma unroll
        for (int k = 0; k < PF_FIELDS; ++k) last[k] = 0.0;
        const bool masks = !COEFFS_IN && a.use_near_mask && a.has_obstacles;
#pragma nounroll
        for (int c = 0; c < nchunks; ++c) {
            const int i = c * G + gl;
            double f[PF_FIELDS];
            lon_step(a, rt, lp.lon, i, f);
            RP_LSTAMP(5);
            if (masks && L - 1 >= c * G && L - 1 < c * G + G) {   // (group-uniform: all lanes of a group share the pair)
                const int src = L - 1 - c * G;
                last[PF_SD] = group_bcast<G>(f[PF_SD], src); last[PF_SDD] = group_bcast<G>(f[PF_SDD], src);
                last[PF_KR] = group_bcast<G>(f[PF_KR], src); last[PF_KRD] = group_bcast<G>(f[PF_KRD], src);
                last[PF_PX] = group_bcast<G>(f[PF_PX], src); last[PF_PY] = group_bcast<G>(f[PF_PY], src);
                last[PF_COS_REF] = group_bcast<G>(f[PF_COS_REF], src); last[PF_SIN_REF] = group_bcast<G>(f[PF_SIN_REF], src);
                last[PF_INV_SD] = group_bcast<G>(f[PF_INV_SD], src);
            }
            if (i <= N) near_mask_step<COEFFS_IN>(a, rt, lp, i, f, (masks && L >= 1) ? last : nullptr);
            else f[PF_NEAR] = f[PF_NEAR_S] = 0.0;
            RP_LSTAMP(6);
            bad_a |= (i < L) && (fabs(f[PF_SDD]) > a.a_max);   // pre-filter, :798
            bad_v |= (i < L) && (f[PF_SD] < -RP_EPS);          // pre-filter, :802
            if (i <= N) {
                double *o = prow + i;
#pragma unroll
                for (int k = 0; k < PF_FIELDS; ++k) o[(size_t)k * n] = f[k];
            }
            RP_LSTAMP(7);
        }
        const bool any_a = group_ballot<G>(bad_a, gbase) != 0, any_v = group_ballot<G>(bad_v, gbase) != 0;
        if (gl == 0) {
            PairHdr h;
            h.lat_T = lp.lat_T;
            h.s0 = lp.lon.c0;   // s[0] == c0 exactly (:762)
            h.pre_reason = draw ? RP_REASON_NONE : (any_a ? RP_REASON_ACCELERATION : (any_v ? RP_REASON_VELOCITY : RP_REASON_NONE));
            h.L = L;
            a.pair_hdr[slot] = h;
        }
        RP_LSTAMP(8);
    }
}

// Store of one state-row element.  WT (single-launch variant): agent-scope write-through store.  The rows of a small
// batch then do not sit dirty in the L2s until the end-of-kernel write-back: measured 23.2 -> 21.8 us on cfg2
// (profiles/r01_dead_ends.txt also records what did NOT pay: running the selection epilogue in the last workgroup on
// top of such stores).
template <bool WT, bool NT = false>
__device__ __forceinline__ void st_row(double *p, double v) {
    if (WT) __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    else if (NT) __builtin_nontemporal_store(v, p);   // large batches: whole 128-byte lines, written once, never read by a kernel
                                                      // (cfg5 2.53 -> 2.33 ms; on partial-line runs they measured 1.3-2.6x slower)
    else *p = v;
}

// one step of a longitudinal profile in registers
struct ProfStep { double f[PF_FIELDS]; };

// rows of one profile step: wave-uniform base + 32-bit byte offset per lane (one scalar-operand add per row)
// MULV: k * n8 formed per lane (v_mad_u32_u24) instead of 16 hoisted scalar registers (see row_at) -- pays in the variants
// that run out of scalar registers (16 lanes per candidate), costs a few percent in those that do not (measured: cfg5 draw).
template <int FIELDS = PF_FIELDS, bool MULV = false>   // the first FIELDS rows (kernels without a collision query do not read cos / sin of theta_ref)
__device__ __forceinline__ ProfStep load_profile(const char *base, uint32_t off8, uint32_t n8) {
    ProfStep p;
    if (MULV) asm volatile("" : "+v"(n8));
#pragma unroll
    for (int k = 0; k < FIELDS; ++k)
        p.f[k] = *reinterpret_cast<const double *>(base + (size_t)(MULV ? __umul24((uint32_t)k, n8) + off8 : off8 + (uint32_t)k * n8));
#pragma unroll
    for (int k = FIELDS; k < PF_FIELDS; ++k) p.f[k] = 0.0;
    return p;
}

// Single-launch variant: the workgroup's own profile rows live in LDS item by item -- [pair][step][PF_STRIDE] doubles,
// fields of one (pair, step) next to each other -- so that every field sits at a compile-time offset from ONE address per
// lane (the field-major layout of the global rows costs a scalar row offset per field, 16 registers held across the step
// loop and spilled).  PF_STRIDE = 17 doubles: lanes of consecutive steps start 17 eight-byte banks apart, conflict-free.
template <int FIELDS = PF_FIELDS>
__device__ __forceinline__ ProfStep load_profile_item(const double *item) {
    ProfStep p;
#pragma unroll
    for (int k = 0; k < FIELDS; ++k) p.f[k] = item[k];
#pragma unroll
    for (int k = FIELDS; k < PF_FIELDS; ++k) p.f[k] = 0.0;
    return p;
}

// raw per-candidate inputs of the evaluation kernel, fetched one iteration ahead of their use
struct CandIn {
    double v[6];   // grid mode: v[0] = lateral sample d;  COEFFS_IN: lateral coefficients
    double lat_T, s0;
    int pre_reason, L;
    int64_t pair_slot;
};

// lateral input of a candidate (global loads only): the lateral sample d, or the explicit coefficients
template <bool COEFFS_IN>
__device__ __forceinline__ void fetch_lateral(const KArgs &a, int64_t gidx, CandIn &ci, int64_t hdr_pair0) {
    int64_t pair;
    if (COEFFS_IN) {
        const double *pt = a.lat_coeffs + 6 * gidx;
#pragma unroll
        for (int k = 0; k < 6; ++k) ci.v[k] = pt[k];
        pair = pair_index<true>(a, gidx);
    } else {
        const uint32_t g32 = (uint32_t)gidx, nd = (uint32_t)a.nD;   // candidate indices fit 32 bits (checked by the host)
        const uint32_t p32 = g32 / nd;
        pair = p32;
        ci.v[0] = grid_base(a)[a.nT + a.nL + (int)(g32 - p32 * nd)];
    }
    ci.pair_slot = pair - hdr_pair0;
}

__device__ __forceinline__ void fetch_header(const PairHdr *hdrs, CandIn &ci) {
    const PairHdr h = hdrs[ci.pair_slot];
    ci.lat_T = h.lat_T;
    ci.s0 = h.s0;
    ci.pre_reason = h.pre_reason;
    ci.L = h.L;
}

// hdrs[pair - hdr_pair0]: the pair headers written by rp_lon_kernel (global) or by the workgroup itself (LDS)
template <bool COEFFS_IN>
__device__ __forceinline__ CandIn fetch_candidate(const KArgs &a, int64_t gidx, const PairHdr *hdrs, int64_t hdr_pair0) {
    CandIn ci;
    fetch_lateral<COEFFS_IN>(a, gidx, ci, hdr_pair0);
    fetch_header(hdrs, ci);
    return ci;
}

// ------------------------------------------------------------------------------------------------
// Candidate evaluation: one group of G lanes per candidate, lane = time step.
// ONE_CHUNK: N + 1 <= G, the step loop disappears together with its carried state.
//
// Arithmetic notes (all within the 1e-6 state tolerance, typically ~1e-13):
//   * theta_cl = atan(d') on the moving branch, hence cos(theta_cl) = 1/sqrt(1 + d'^2) and
//     tan(theta_cl) = d' without a second and third transcendental; only standstill lanes
//     (reactive_planner.py:864-873) evaluate sin/cos, behind a wave-uniform branch.
//   * divisions by quantities reused several times (s_dot, 1 - k_r d, segment length) become one
//     Newton-refined reciprocal each; threshold tests are rearranged to multiplications.
// ------------------------------------------------------------------------------------------------
// STAGE_OUT (with MAT): the state rows of the candidates of a wavefront are collected in LDS and leave
// as ONE linear, 16-byte-per-lane stream -- the [14][N+1] blocks of consecutive candidates are
// contiguous in memory and 112 (N+1) bytes long, i.e. always 16-byte aligned.  Half the store
// instructions of row-wise 8-byte stores, every one of them a full 1 KiB.
//
// LON_FUSED (single-launch step for batches of at most one candidate per lane group, i.e.
// gridDim.x * GPB >= count): the workgroup first computes the longitudinal profiles of the (at most
// a.lds_pairs) pairs its own candidates belong to into LDS -- the work of rp_lon_kernel without the
// launch, the global round trip of the profile rows and the kernel boundary.
// BLOCK: threads per workgroup.  256 (four wavefronts) wherever the workgroup shares something -- the single-launch variant's
// tables and profile rows, the LDS tile of the staged copy-out; the plain two-kernel variants of LARGE batches run one wavefront
// per workgroup (BLOCK = 64): a workgroup is placed, and its slot freed, as a whole, and the four wavefronts of a 256-thread
// workgroup finish far apart (in-kernel stamps on cfg3: the first at 30 k cycles, the workgroup at 42 k).
// SWEEP: the bounded collision sweep (KArgs::sweep_bound) -- a variant of its own (16 lanes per candidate, no state rows, two-kernel path):
// as a run-time switch of the plain variants it cost them two more spilled registers and 3.5 % (cfg3 eager 93.6 -> 97.0 us).
template <int G, bool MAT, bool COEFFS_IN, int COLL, bool ONE_CHUNK, bool STAGE_OUT, bool LON_FUSED, int BLOCK = RP_BLOCK, bool SWEEP = false>
__global__ __launch_bounds__(BLOCK, RP_WAVES_PER_SIMD) void rp_eval_kernel(const KArgsG ag) {
    static_assert(!SWEEP || (G == 16 && !MAT && COLL != 0 && !STAGE_OUT && !LON_FUSED && BLOCK == RP_BLOCK), "the sweep's variant");
    const KArgs &a = ag.k;
    extern __shared__ double lds_out[];   // STAGE_OUT: [groups per block][14][N+1];  LON_FUSED: tables, profiles, headers, votes
    static_assert(!(LON_FUSED && STAGE_OUT), "the single-launch variant stores state rows directly");
    constexpr bool RP_WT = LON_FUSED && RP_WRITE_THROUGH;
    constexpr bool RP_NT = MAT && !LON_FUSED && !STAGE_OUT;   // rows of the two-kernel path stored straight to memory: streaming stores
    touch_kernargs<10>();

    const int tid = threadIdx.x;
    RP_STAMP(0);
    RP_TL(0);
    const int lane = tid & 63;
    const int gl = lane & (G - 1);        // lane inside the group = time step inside the chunk
    const int gbase = lane & ~(G - 1);    // first lane of the group inside the wave
    static_assert(BLOCK == RP_BLOCK || (!LON_FUSED && !STAGE_OUT), "smaller workgroups: plain two-kernel variants only");
    constexpr int GPB = BLOCK / G;        // groups per block
    constexpr int GPW = 64 / G;           // groups per wave
    constexpr int PFN = COLL ? PF_FIELDS : PF_COS_REF;   // profile rows this variant reads
    const int wave_in_block = tid >> 6;
    const int group_in_wave = gbase / G;
    const int grp = tid / G;              // group inside the block

    // a level of a chain (rp_plan_levels) behind the one that delivered: nothing to do.  (The winner re-evaluation of the level that
    // delivered belongs to that level: its number is not below the gate's.)
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;
    if (a.single_index && *a.single_index < 0) {   // no winner to re-evaluate (uniform): only the completion ticket
        if (a.host_seq && a.ticket_if_none && tid == 0) __hip_atomic_store(a.host_seq, a.seq_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int64_t wave_first = ((int64_t)blockIdx.x * GPB) + (int64_t)wave_in_block * GPW;  // first group of this wave
    int64_t count_ = a.single_index ? 1 : a.count;
    if (!LON_FUSED && a.index_list) {   // list mode: as many candidates as the list holds (workgroups beyond it leave an empty partial)
        const uint32_t lc = *a.list_count;
        count_ = (int64_t)(lc < (uint32_t)a.list_cap ? lc : (uint32_t)a.list_cap);
        if ((int64_t)blockIdx.x * GPB >= count_) {
            if (a.partials) {
                const Partials bp = partials_at(a.partials, a.partials_cap, a.partials_first + (int)blockIdx.x);
                if (tid == 0) { bp.cost[0] = 0.0; bp.idx[0] = -1; }
                if (tid >= 8 && tid < 8 + RP_PARTIAL_CNT) bp.cnt[tid - 8] = 0u;
            }
            return;
        }
    }
    const int64_t count = count_;
    if (!LON_FUSED && a.lazy_ctl && blockIdx.x == 0 && tid < (int)(sizeof(LazyCtl) / 8))   // (first pass of the lazy stage)
        reinterpret_cast<unsigned long long *>(a.lazy_ctl)[tid] = 0ull;
    if (!LON_FUSED && a.sweep_init && blockIdx.x == 0 && tid == 0)   // (pass 1 of the bounded sweep: no free candidate known yet)
        __hip_atomic_store(a.sweep_init, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The sweep itself: a lane group evaluates its candidate only if pass 1 left it FEASIBLE and no cheaper free candidate is known; a
    // workgroup none of whose candidates needs that leaves an empty partial and is gone after one round trip.
    constexpr bool sweep = SWEEP;
    bool sweep_need = true;
    if (sweep) {
        const int64_t sl = wave_first + group_in_wave;
        const bool v0 = sl < count;
        const int64_t slc = v0 ? sl : 0;
        const uint32_t st0 = a.status[slc];
        const double cs0 = a.cost[slc];
        const unsigned long long bound = __hip_atomic_load(a.sweep_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sweep_need = v0 && RP_STATUS_LABEL(st0) == RP_LABEL_FEASIBLE && cs0 == cs0 && cost_key(cs0) <= bound;
        __shared__ int sh_sweep_any;
        if (tid == 0) sh_sweep_any = 0;
        __syncthreads();
        if (__ballot(sweep_need) != 0 && lane == 0) sh_sweep_any = 1;
        __syncthreads();
        if (!sh_sweep_any) {   // (workgroup-uniform)
            if (a.partials) {
                const Partials bp = partials_at(a.partials, a.partials_cap, a.partials_first + (int)blockIdx.x);
                if (tid == 0) { bp.cost[0] = 0.0; bp.idx[0] = -1; }
                if (tid >= 8 && tid < 8 + RP_PARTIAL_CNT) bp.cnt[tid - 8] = 0u;
            }
            return;
        }
    }

    __shared__ GroupScratch sh_grp[GPB];
    __shared__ double sh_best_cost[GPB];
    __shared__ long long sh_best_idx[GPB];
    __shared__ int sh_cnt[10];

    // where the profile rows and pair headers live: global memory (rp_lon_kernel) or this workgroup's LDS
    const double *prof_base = a.profile;
    const PairHdr *hdr_base = a.pair_hdr;
    int64_t pair0 = a.pair_begin;
    CandIn cin;
    ProfStep pf0;        // first step block of this lane group's candidate
    const double *slot_lds = nullptr;   // single-launch variants with static shapes: the workgroup's copy of ObsTables::slot
    if (LON_FUSED) {
        // dynamic LDS: reference tables | profile rows [lds_pairs][PF_FIELDS][n] | pair headers | pre-filter votes |
        //              time sample of each pair [lds_pairs]
        const int n0 = a.N + 1;
        double *const lds_tab = lds_out;
        double *const lds_prof = lds_tab + a.table_words;   // table_words is even: 16-byte aligned
        PairHdr *const lds_hdr = reinterpret_cast<PairHdr *>(lds_prof + (size_t)a.lds_pairs * PF_STRIDE * (size_t)n0);   // (PF_STRIDE * 8 * pairs * n0: a multiple of 8)
        int *const lds_flags = reinterpret_cast<int *>(lds_hdr + a.lds_pairs);
        double *const lds_T = reinterpret_cast<double *>(lds_flags + ((a.lds_pairs + 1) & ~1));   // [lds_pairs] time sample of the pair
        // member slots of the static clusters (ObsTables::slot), 16-byte aligned behind the time samples
        double *const lds_slot = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(lds_T + a.lds_pairs) + 15) & ~(uintptr_t)15);
        const int n_slots = (COLL == 2 && a.has_obstacles) ? a.obs.n_clus * a.obs.clus_per : 0;
        slot_lds = lds_slot;
        // candidates of this workgroup: slots [blockIdx.x * GPB, ...) -- exactly one group of lanes each
        const int64_t s_first = (int64_t)blockIdx.x * GPB;
        const int64_t s_last = (s_first + GPB <= count ? s_first + GPB : count) - 1;
        const int64_t g_first = a.single_index ? *a.single_index : a.cand_begin + (s_first < count ? s_first : 0);
        const int64_t g_last = a.single_index ? g_first : a.cand_begin + (s_last >= s_first ? s_last : s_first);
        const int64_t p_first = COEFFS_IN ? g_first : (int64_t)((uint32_t)g_first / (uint32_t)a.nD);
        const int64_t p_last = COEFFS_IN ? g_last : (int64_t)((uint32_t)g_last / (uint32_t)a.nD);
        int P = (int)(p_last - p_first) + 1;
        P = P > a.lds_pairs ? a.lds_pairs : P;   // (cannot happen: the host sizes lds_pairs for the worst alignment)
        const int items = P * n0;                // item = (pair, step)
        // -- every global load of the prologue is requested here, one round trip for all of them: the grid values of the
        //    workgroup's pairs and candidates (grid plans: ONE request per value and workgroup, handed on through LDS -- the
        //    grids may sit in the kernarg segment, which the L2 does not keep: per-lane requests from every wavefront of
        //    465 workgroups for the same nine lines cost the kernel 1.1 us), the table block
        __shared__ double sh_gT[GPB], sh_gL[GPB], sh_gD[GPB];
        __shared__ int sh_gtl[GPB];
        const int64_t slot0 = wave_first + group_in_wave;
        const int64_t g0 = a.single_index ? g_first : a.cand_begin + (slot0 < count ? slot0 : (wave_first < count ? wave_first : s_first));
        int j = lane;   // items are spread over the lanes of every wavefront alike: wavefront w runs part w of lon_step for all of them
        LonPairIn lin;
        double g_T = 0.0, g_L = 0.0, g_D = 0.0;
        int g_tl = 0;
        if (COEFFS_IN) {
            fetch_lateral<COEFFS_IN>(a, g0, cin, p_first);
            if (j < items) lin = lon_pair_fetch<COEFFS_IN>(a, p_first + (int)((uint32_t)j / (uint32_t)n0));
        } else {
            cin.pair_slot = (int64_t)((uint32_t)g0 / (uint32_t)a.nD) - p_first;
            if (tid < P) {
                const LonPairIn q = lon_pair_fetch<COEFFS_IN>(a, p_first + tid);
                g_T = q.c[0]; g_L = q.c[1]; g_tl = q.L;
            } else if (tid >= 64 && tid < 64 + GPB) {
                const int64_t gq = g_first + (tid - 64);
                const uint32_t g32 = (uint32_t)(gq <= g_last ? gq : g_last), nd = (uint32_t)a.nD;
                g_D = grid_base(a)[a.nT + a.nL + (int)(g32 - (g32 / nd) * nd)];
            }
        }
        RP_STAMP(15);   // single-launch prologue: candidate / grid loads requested
        __shared__ int sh_miss;
        const bool windowed = a.win_n > 0;
        if (windowed) {
            // the vertices [win_k0, win_k0 + win_n) of every row and the bucket entries that point into them, at the places
            // they have in the whole block (8-byte pieces: rows start at multiples of n_ref doubles)
            const int wn = a.win_n, total = TB_ROWS * wn;
            typedef double __attribute__((address_space(3))) *lds_double;
            const lds_double ltab = (lds_double)lds_tab;
            for (int i0 = tid; i0 < total; i0 += 4 * RP_BLOCK) {   // (four pieces per thread requested before the first is stored: copy16_to_lds)
                double v[4];
                int at[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * RP_BLOCK < total ? i0 + u * RP_BLOCK : total - 1;
                    at[u] = (idx >> a.win_shift) * a.n_ref + a.win_k0 + (idx & (wn - 1));
                    v[u] = a.tables[at[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * RP_BLOCK < total) ltab[at[u]] = v[u];
            }
            const int *bsrc = reinterpret_cast<const int *>(a.tables + TB_ROWS * a.n_ref) + a.win_b0;
            int *bdst = reinterpret_cast<int *>(lds_tab + TB_ROWS * a.n_ref) + a.win_b0;
            for (int idx = tid; idx < a.win_nb; idx += RP_BLOCK) bdst[idx] = bsrc[idx];
        } else {
            copy16_to_lds(a.tables, lds_tab, a.table_words >> 1, tid);
        }
        if (!COEFFS_IN) {
            if (tid < P) { sh_gT[tid] = g_T; sh_gL[tid] = g_L; sh_gtl[tid] = g_tl; }
            else if (tid >= 64 && tid < 64 + GPB) sh_gD[tid - 64] = g_D;
        }
        RP_STAMP(16);   // table copy issued (this wave's share)
        if (tid < P) lds_flags[tid] = 0;
        if (tid < 10) sh_cnt[tid] = 0;
        if (tid < GPB) { sh_best_cost[tid] = 0.0; sh_best_idx[tid] = -1; }
        if (tid == 0) sh_miss = 0;
        __syncthreads();
        RP_STAMP(4);    // single-launch prologue: inputs and tables have arrived
        const RefTab rt = ref_tab(lds_tab, a.n_ref, a.pos_first, a.pos_last);
        auto item_input = [&](int jj) {   // grid plans: from the workgroup's LDS copy
            if (COEFFS_IN) return lon_pair_fetch<COEFFS_IN>(a, p_first + (int)((uint32_t)jj / (uint32_t)n0));
            const int p = (int)((uint32_t)jj / (uint32_t)n0);
            LonPairIn q;
            q.c[0] = sh_gT[p]; q.c[1] = sh_gL[p]; q.L = sh_gtl[p];
            return q;
        };
        if (!COEFFS_IN) {
            cin.v[0] = sh_gD[(int)(g0 - g_first)];
            if (j < items) lin = item_input(j);
        }
        LonPair lp;
        double s_it = 0.0;
        if (j < items) {
            lp = lon_pair_make<COEFFS_IN>(a, lin, n0);
            s_it = lp.lon.pos((double)(j - (int)((uint32_t)j / (uint32_t)n0) * n0) * a.dt);
        }
#pragma nounroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {   // (workgroup-uniform) some item left the window: the whole block, then everything again
                if (!windowed || !sh_miss) break;
                copy16_to_lds(a.tables, lds_tab, a.table_words >> 1, tid);
                if (tid < P) lds_flags[tid] = 0;
                __syncthreads();
                j = lane;
                if (j < items) {
                    lin = item_input(j);
                    lp = lon_pair_make<COEFFS_IN>(a, lin, n0);
                    s_it = lp.lon.pos((double)(j - (int)((uint32_t)j / (uint32_t)n0) * n0) * a.dt);
                }
            }
            const bool whole = !windowed || pass == 1;
            if (COLL == 2 && pass == 0 && wave_in_block == RP_BLOCK / 64 - 1) {
                // the wavefront that only clears mask words below brings the slot table of the static clusters into LDS
                // while the others work out the profile rows (one round trip, requests back to back)
                copy16_to_lds<64, 8>(a.obs.slot, lds_slot, n_slots * (RP_SLOT_ROW / 2), lane);
            }
            while (j < items) {
                const int p = (int)((uint32_t)j / (uint32_t)n0), i = j - p * n0;
                RP_STAMP(18);
                const double s = s_it;
                const bool inside = whole || (s >= a.win_s_lo && s < a.win_s_hi);
                double *o = lds_prof + (size_t)j * PF_STRIDE;   // item j = (p, i): [p][i][field]
                RP_STAMP(19);
                if (wave_in_block == 0) {
                    if (inside) lon_step_part<0>(a, rt, lp.lon, i, s, o, (size_t)1);
                    else sh_miss = 1;
                    RP_STAMP(20);
                } else if (wave_in_block == 1) {
                    if (inside) lon_step_part<1>(a, rt, lp.lon, i, s, o, (size_t)1);
                    int fl = 0;   // pre-filter votes, reactive_planner.py:798,802 (the profile's s_ddot, s_dot: polynomial values)
                    if (i < lp.L) {
                        const double t = (double)i * a.dt;
                        double sd = lp.lon.vel(t);
                        if (fabs(sd) < RP_EPS) sd = 0.0;
                        fl = (fabs(lp.lon.acc(t)) > a.a_max ? 1 : 0) | (sd < -RP_EPS ? 2 : 0);
                    }
                    if (fl) atomicOr(&lds_flags[p], fl);
                    if (i == 0) { lds_T[p] = lp.T; lds_hdr[p].lat_T = lp.lat_T; lds_hdr[p].s0 = lp.lon.c0; lds_hdr[p].L = lp.L; }
                } else if (wave_in_block == 2) {
                    if (inside) lon_step_part<2>(a, rt, lp.lon, i, s, o, (size_t)1);
                } else {
                    o[PF_NEAR] = o[PF_NEAR_S] = mask_as_double(0);   // filled by the whole workgroup after the barrier
                }
                j += 64;
                if (j < items) {
                    lin = item_input(j);
                    lp = lon_pair_make<COEFFS_IN>(a, lin, n0);
                    s_it = lp.lon.pos((double)(j - (int)((uint32_t)j / (uint32_t)n0) * n0) * a.dt);
                }
            }
            RP_STAMP(21);
            __syncthreads();
        }
        RP_STAMP(17);   // single-launch prologue: profile rows computed (before the broad-phase masks)
        // header of this lane's candidate (the pre-filter verdict comes from the votes), its first profile step and its lateral
        // polynomial (sampling.py:226-238, 268-270) -- here, ahead of the mask phase, whose lanes mostly wait for obstacle rows
        {
            const PairHdr h = lds_hdr[cin.pair_slot];
            const int fl = lds_flags[cin.pair_slot];
            cin.lat_T = h.lat_T;
            cin.s0 = h.s0;
            cin.L = h.L;
            cin.pre_reason = (a.flags & RP_FLAG_DRAW_ALL) ? RP_REASON_NONE
                             : ((fl & 1) ? RP_REASON_ACCELERATION : ((fl & 2) ? RP_REASON_VELOCITY : RP_REASON_NONE));
            Poly lat;
            if (COEFFS_IN) lat = {cin.v[0], cin.v[1], cin.v[2], cin.v[3], cin.v[4], cin.v[5]};
            else lat = quintic_coeffs(a.x0_lat[0], a.x0_lat[1], a.x0_lat[2], cin.v[0], 0.0, 0.0, cin.lat_T);
            if (gl == 0) park_poly(sh_grp[grp].poly, lat);   // same-wave LDS ordering makes it visible to the group
        }
        if (!COEFFS_IN && a.use_near_mask && a.has_obstacles && a.obs.n_dyn > 0) {   // (only dynamic obstacles have (pair, step) masks)
            // Broad phase of the collision query (near_mask_step), spread over the whole workgroup: one lane per
            // (item, shape) test; every lane derives its item's bound from the profile rows in LDS (one lane per item
            // walking every shape would sit on the critical path of the workgroup).
            const ObsTables &ob = a.obs;
            auto item_bound = [&](int it, double &bx, double &by, double &bR) -> bool {
                const int p = (int)((uint32_t)it / (uint32_t)n0), i = it - p * n0;
                const double *row = lds_prof + (size_t)p * PF_STRIDE * (size_t)n0;   // [step][field] of pair p
                BoundIn in;
                in.L = lds_hdr[p].L; in.T = lds_T[p]; in.lat_T = lds_hdr[p].lat_T; in.s0 = lds_hdr[p].s0;
                const double *it_i = row + (size_t)i * PF_STRIDE, *it_1 = row + (size_t)(in.L - 1) * PF_STRIDE;
                in.s_i = it_i[PF_S]; in.px_i = it_i[PF_PX]; in.py_i = it_i[PF_PY];
                in.sd1 = it_1[PF_SD]; in.sdd1 = it_1[PF_SDD]; in.kr1 = it_1[PF_KR]; in.krd1 = it_1[PF_KRD];
                in.px1 = it_1[PF_PX]; in.py1 = it_1[PF_PY]; in.cs1 = it_1[PF_COS_REF]; in.sn1 = it_1[PF_SIN_REF];
                in.inv_sd1 = it_1[PF_INV_SD];
                return pair_step_bound(a, i, in, bx, by, bR);
            };
            auto mask_word = [&](int it, int field) -> unsigned long long * {
                return reinterpret_cast<unsigned long long *>(lds_prof + (size_t)it * PF_STRIDE + field);
            };
            const gcdouble dyn = (gcdouble)ob.dyn;
            const size_t plane = (size_t)ob.n_dyn * (size_t)ob.n_steps;
            // lanes_per_item lanes share an item: each works the bound out once and takes every lanes_per_item-th shape
            const int lanes_per_item = items >= RP_BLOCK ? 1 : RP_BLOCK / items;
            const int sub = tid % lanes_per_item;
            for (int it = tid / lanes_per_item; it < items; it += RP_BLOCK / lanes_per_item) {
                const int i = it - (int)((uint32_t)it / (uint32_t)n0) * n0;
                const int kk = a.time_step0 + i * a.factor - ob.dyn_t0;
                const bool dyn_here = ob.n_dyn > 0 && kk >= 0 && kk < ob.n_steps;   // else: no dynamic obstacle at this scenario step
                // this lane's first obstacle circle is requested before the bound is worked out (the bound is ~100 dependent
                // instructions on LDS values: the round trip hides behind them)
                const bool first = dyn_here && sub < ob.n_dyn;
                double f_cx = 0.0, f_cy = 0.0, f_r = 0.0;
                if (first) {
                    const gcdouble o = dyn + (size_t)sub * ob.n_steps + kk;
                    f_cx = o[0]; f_cy = o[plane]; f_r = o[6 * plane];
                }
                double bx, by, bR;
                RP_STAMP(22);
                const bool bounded = item_bound(it, bx, by, bR);
                RP_STAMP(23);
                if (!bounded) {   // no bound: every bit
                    if (sub == 0 && dyn_here) atomicOr(mask_word(it, PF_NEAR), ~0ull);
                    continue;
                }
                if (dyn_here) {
                    uint64_t m = 0;
                    if (first) {
                        const double dx = f_cx - bx, dy = f_cy - by, rr = bR + f_r;   // NaN centre: absent, no bit
                        if (dx * dx + dy * dy <= rr * rr * 1.000001) m |= 1ull << (sub < 63 ? sub : 63);
                    }
                    for (int jo = sub + lanes_per_item; jo < ob.n_dyn; jo += lanes_per_item) {
                        const gcdouble o = dyn + (size_t)jo * ob.n_steps + kk;
                        const double dx = o[0] - bx, dy = o[plane] - by, rr = bR + o[6 * plane];
                        if (dx * dx + dy * dy <= rr * rr * 1.000001) m |= 1ull << (jo < 63 ? jo : 63);
                    }
                    if (m) atomicOr(mask_word(it, PF_NEAR), m);
                }
                // (static shapes have no (pair, step) mask: static_grid_mask per pose)
            }
            __syncthreads();
        }
        RP_STAMP(10);   // single-launch prologue: profile rows in LDS
        prof_base = lds_prof;
        hdr_base = lds_hdr;
        pair0 = p_first;
        pf0 = load_profile_item<PFN>(lds_prof + ((size_t)cin.pair_slot * (size_t)n0 + (size_t)(gl <= a.N ? gl : a.N)) * PF_STRIDE);
    } else {
        // (lanes without a candidate shadow the first candidate of their wavefront: its profile rows exist and are close)
        const int64_t slot0 = wave_first + group_in_wave;
        const int64_t ls0 = slot0 < count ? slot0 : (wave_first < count ? wave_first : (a.index_list ? (int64_t)blockIdx.x * GPB : 0));
        const int64_t g0 = a.single_index ? *a.single_index : (a.index_list ? (int64_t)a.index_list[ls0] : a.cand_begin + ls0);
        cin = fetch_candidate<COEFFS_IN>(a, g0, hdr_base, pair0);
        const int32_t ps0 = (int32_t)(pair_index<COEFFS_IN>(a, g0) - pair0);
        const int32_t pw0 = __builtin_amdgcn_readfirstlane(ps0);   // pair of the wavefront's first candidate (wave-uniform)
        const uint32_t n80 = (uint32_t)(a.N + 1) * 8u;
        pf0 = load_profile<PFN, G == 16>(reinterpret_cast<const char *>(prof_base + ((size_t)pw0 * PF_FIELDS) * (size_t)(a.N + 1)),
                           (uint32_t)(ps0 - pw0) * PF_FIELDS * n80 + (uint32_t)(gl <= a.N ? gl : a.N) * 8u, n80);
    }

    if (!LON_FUSED) {
        if (tid < 10) sh_cnt[tid] = 0;
        if (tid < GPB) { sh_best_cost[tid] = 0.0; sh_best_idx[tid] = -1; }
        __syncthreads();
    }
    RP_STAMP(1);

    // The kernel arguments of everything below are read through a pointer the compiler cannot see through: the loads
    // then happen HERE, after the prologue, instead of at kernel entry -- where the values would be live across the
    // prologue of the single-launch variant, lose their scalar registers to it and be reloaded (v_readlane) at every use
    // in the step loop.
    typedef const KArgs __attribute__((address_space(4))) *kargs_cptr;
    // (the kernarg segment itself -- KArgs is the kernel's only explicit argument, at offset 0; `&a` would make the compiler
    //  materialise a private copy of the by-value struct)
    // Variants that write no state rows and have no long prologue keep the plain argument accesses (entry-block loads):
    // they spilled little to begin with and lose 2 % to the later loads (cfg5 production mode 1.126 -> 1.155 ms).
    kargs_cptr ap_late = (LON_FUSED || MAT) ? (kargs_cptr)__builtin_amdgcn_kernarg_segment_ptr() : (kargs_cptr)&a;
    if (RP_LATE_KERNARGS && LON_FUSED) asm volatile("" : "+s"(ap_late));
    const KArgs __attribute__((address_space(4))) &al = *ap_late;
    const int N = al.N, n = N + 1;
    const int nchunks = ONE_CHUNK ? 1 : (n + G - 1) / G;
    const double dt = al.dt;
    const bool draw = (al.flags & RP_FLAG_DRAW_ALL) != 0;
    const bool low = al.low_vel_mode != 0;
    const uint32_t cm = al.constraint_mask;
    const int mid = n / 2;  // int(len(v) / 2), cost_function.py:59
    double *const gs_poly = sh_grp[grp].poly;
    double *const gs_last = sh_grp[grp].last;

    // this lane's share of the cost of one (possibly extended) state, cost_function.py:51-71 / 82-92:
    // every lane adds its own step, the lanes holding steps N and N/2 also add the terminal terms
    auto cost_terms = [&](int i, double acc, double v, double s, double d, double th_cl) -> double {
        double e, cst;
        e = al.w_a * acc; cst = e * e;
        e = 0.25 * (al.desired_d - d); cst = __builtin_fma(e, e, cst);
        e = 0.25 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
        if (al.has_speed) { e = 5.0 * (v - al.desired_speed); cst = __builtin_fma(e, e, cst); }
        if (al.has_s) { e = 0.25 * (al.desired_s - s); cst = __builtin_fma(e, e, cst); }
        if (i == N) {
            e = 20.0 * (al.desired_d - d); cst = __builtin_fma(e, e, cst);
            e = 5.0 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
            if (al.has_speed) { e = v - al.desired_speed; cst += 50.0 * (e * e); }
            if (al.has_s) { e = 20.0 * (al.desired_s - s); cst = __builtin_fma(e, e, cst); }
        }
        if (i == mid && al.has_speed) { e = v - al.desired_speed; cst += 100.0 * (e * e); }
        return cst;
    };

    // The grid covers the batch: every lane group evaluates exactly one candidate (no persistent loop -- a loop over
    // candidates costs ~40 VGPRs of hoisted invariants and look-ahead state, i.e. the third wavefront per SIMD).
    if (wave_first < count) {   // wave-uniform
        const int64_t w0 = wave_first;
        const int64_t slot = w0 + group_in_wave;                      // local candidate slot of this group
        const bool valid = slot < count && sweep_need;
        // (lanes without a candidate shadow the first candidate of their wavefront, whose profile rows exist)
        const int64_t gidx = al.single_index ? *al.single_index
                             : ((!LON_FUSED && al.index_list) ? (int64_t)al.index_list[valid ? slot : w0] : al.cand_begin + (valid ? slot : w0));

        // ---- lateral polynomial: sampling.py:226-238, 268-270
        const int L = cin.L;
        const double s0 = cin.s0;
        const uint32_t pre_reason = (uint32_t)cin.pre_reason;   // pre-filter verdict of the pair (label stays None)
        // profile rows are addressed arithmetically (no dependence on the header load just issued)
        const int32_t pair_slot_ = (int32_t)(pair_index<COEFFS_IN>(al, gidx) - pair0);
        const int32_t pair_w = __builtin_amdgcn_readfirstlane(pair_slot_);   // wavefront's first candidate: wave-uniform base
        const uint32_t n8p = (uint32_t)n * 8u;
        const char *const pbase = reinterpret_cast<const char *>(prof_base + ((size_t)pair_w * PF_FIELDS) * (size_t)n);
        const uint32_t poff8 = (uint32_t)(pair_slot_ - pair_w) * PF_FIELDS * n8p;
        ProfStep pf = pf0;
        if (!LON_FUSED) {   // (single-launch variant: parked in the prologue)
            Poly lat;
            if (COEFFS_IN) lat = {cin.v[0], cin.v[1], cin.v[2], cin.v[3], cin.v[4], cin.v[5]};
            else lat = quintic_coeffs(al.x0_lat[0], al.x0_lat[1], al.x0_lat[2], cin.v[0], 0.0, 0.0, cin.lat_T);
            if (gl == 0) park_poly(gs_poly, lat);   // same-wave LDS ordering makes it visible to the group
        }
        RP_STAMP(2);   // polynomial ready

        // per-candidate carried state
        int fail_step = -1, ood_step = -1;
        uint32_t fail_reason = RP_REASON_NONE;
        bool collide = false;
        double cost_acc = 0.0;

        bool alive = valid && pre_reason == RP_REASON_NONE;
        if (__ballot(alive) != 0) {   // wave-uniform
            double theta_carry = al.x0_orientation;  // theta_gl[base-1]; at base 0 the i == 0 rule of :866
            double kappa_carry = 0.0;
            double cumx = 0.0, cumy = 0.0;
            const bool store_ok = MAT && valid && pre_reason == RP_REASON_NONE;
            // Row addresses: a wave-uniform base (the first candidate of this wavefront) plus a 32-bit byte offset per
            // lane; each row then costs one 32-bit add with a scalar operand (row * n * 8) instead of a 64-bit
            // address per (lane, row) formed from 14 hoisted -- and spilled -- row pointers.
            // Rows stored directly lie row_stride doubles apart (N + 1 rounded up to whole 64-byte lines): a lane group's run of
            // G consecutive steps then starts on a line boundary and covers whole lines, which is what a write-through store
            // needs to reach memory as one full-line write (248-byte rows cost 1.28 x the bytes: profiles/r02_pmc_traffic.json).
            const int ns = (MAT && !STAGE_OUT) ? al.row_stride : n;
            // split tail (state_offset): only where 16 lanes store rows straight to memory on the two-kernel path
            constexpr bool SPLIT_OK = MAT && !STAGE_OUT && !LON_FUSED && G == 16;
            const int tail_m = SPLIT_OK ? al.tail_split : 0;
            const uint32_t cand8 = (uint32_t)state_block_doubles(ns, tail_m) * 8u;   // bytes of one candidate's block
            char *const obase = !MAT ? nullptr
                                : (STAGE_OUT ? reinterpret_cast<char *>(lds_out + (size_t)(wave_in_block * GPW) * RP_N_ARRAYS * (size_t)n)
                                             : reinterpret_cast<char *>(al.states) + (size_t)w0 * (size_t)cand8);
            const uint32_t n8 = (uint32_t)ns * 8u;
            const uint32_t lane_off8 = (uint32_t)group_in_wave * (STAGE_OUT ? RP_N_ARRAYS * n8 : cand8);   // this group's candidate inside the wavefront
            // row * n8 is formed where it is used, from a per-lane copy of n8 the compiler cannot see through (one
            // v_mad_u32_u24 per row): as a loop invariant it would be hoisted into 14 scalar registers that live across
            // the whole step loop -- the kernel runs out of those first (spilled ones cost a v_readlane per use)
            // (only where scalar registers are short: the single-launch variants; elsewhere the extra multiply costs ~4 %)
            constexpr bool ROW_MULV = LON_FUSED;
            constexpr bool LATE_STORE = MAT && !STAGE_OUT;
            uint32_t n8v = n8;
            auto row_at = [&](uint32_t off8, int row) -> double * {
                return reinterpret_cast<double *>(obase + (size_t)(ROW_MULV ? __umul24((uint32_t)row, n8v) + off8 : off8 + (uint32_t)row * n8));
            };

#pragma nounroll
            for (int c = 0; c < nchunks; ++c) {   // wave-uniform
                const int base = c * G;
                const int i = base + gl;
                const bool live = i <= N;
                const bool act = i < L;
                if (ROW_MULV) asm volatile("" : "+v"(n8v));   // (see row_at: keeps row * n8 out of the loop-invariant scalar registers)

                // -- this step of the pair's longitudinal profile (coalesced: lanes = consecutive steps)
                if (!ONE_CHUNK && c > 0) {   // dead lanes clamp into the rows
                    if (LON_FUSED) pf = load_profile_item<PFN>(prof_base + ((size_t)pair_slot_ * (size_t)n + (size_t)(live ? i : N)) * PF_STRIDE);
                    else pf = load_profile<PFN, G == 16>(pbase, poff8 + (uint32_t)(live ? i : N) * 8u, n8p);
                }
                double s = pf.f[PF_S], sd = pf.f[PF_SD], sdd = pf.f[PF_SDD];
                const double inv_sd = pf.f[PF_INV_SD], th_ref = pf.f[PF_TH_REF];
                const double k_r = pf.f[PF_KR], k_r_d = pf.f[PF_KRD];
                const double px = pf.f[PF_PX], py = pf.f[PF_PY], nx = pf.f[PF_NX], ny = pf.f[PF_NY];
                const bool s_in_dom = pf.f[PF_INDOM] != 0.0;

                // A step block that holds extended states only (i >= L in every lane: horizons longer than the trajectories, cfg3 one
                // block in six, cfg5 every second) has nothing to evaluate -- its lanes take every value from the last valid state
                // below.  (Wave-uniform: the lanes of a wavefront belong to one pair, rarely two.)  Same-box A/B, kernel time
                // (profiles/r04_extended_guard_ab.txt): cfg3 draw 101.5 -> 98.8 us, cfg3 + road boundary 110.0 -> 107.0, cfg4 draw
                // unchanged; cfg5 draw -- state rows, no collision query: all stores -- 2.137 -> 2.20 ms (the wavefronts of a
                // workgroup then fall out of step with their row stores), so that variant evaluates every block as before.  Reading only
                // the near mask of the profile in such a block was tried on top: slower everywhere (cfg3 draw 103.8 us).
                constexpr bool SKIP_EXTENDED = !(MAT && COLL == 0);
                double d = 0.0, dd = 0.0, ddd = 0.0, th_cl = 0.0, th_gl = 0.0, cos_gl = 1.0, sin_gl = 0.0;
                double kappa = 0.0, v = 0.0, acc = 0.0, kdot = 0.0, x = 0.0, y = 0.0;
                uint64_t near_sta_cell = 0;
                if (!SKIP_EXTENDED || __any(act)) {
                    // -- lateral polynomial, reactive_planner.py:756-777
                    const double t = (double)i * dt;
                    const double tau = low ? s - s0 : t;
                    d = poly_pos(gs_poly, tau); dd = poly_vel(gs_poly, tau); ddd = poly_acc(gs_poly, tau);
                    if (fabs(dd) < RP_EPS) dd = 0.0;
                    // Rows stored straight to memory leave once per step block, behind the horizon extension (LATE_STORE): valid and
                    // extended steps of a row then share ONE store instruction, i.e. whole 64-byte lines -- stored separately, the
                    // 32-byte sector that holds the last valid step was written twice (+ 24 B per row on cfg2, PMC WRITE_SIZE).
                    if (!LATE_STORE && store_ok && act) {   // curvilinear rows of valid steps are final here
                        const uint32_t off8 = lane_off8 + (uint32_t)i * 8u;
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_S), s);
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DOT), sd);
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DDOT), sdd);
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_D), d);
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DOT), dd);
                        st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DDOT), ddd);
                    }
                    RP_STAMP(3);   // profile loads + polynomial evaluation

                    // -- d', d'' (:810-832)
                    const bool moving = inv_sd > 0.0;
                    double dp, dpp;
                    if (!low) {
                        dp = dd * inv_sd;
                        const double ddot = ddd - dp * sdd;
                        dpp = ddot * inv_sd * inv_sd;
                    } else {
                        dp = dd;
                        dpp = ddd;
                    }

                    // -- orientations (:842-873) incl. the standstill carry of :866
                    const bool use_atan = moving || low;
                    th_cl = rp_atan(dp);          // np.arctan2(dp, 1.0)
                    th_gl = th_cl + th_ref;
                    // cos / sec / tan of theta_cl: algebraic on the atan branch
                    const double w2 = __builtin_fma(dp, dp, 1.0);
                    double cosT = rp_rsqrt(w2);
                    double secT = w2 * cosT;
                    double tanT = dp;
                    if (__any(act && !use_atan)) {   // standstill lanes: keep the orientation of the last moving step
                        const uint64_t mv = group_ballot<G>(use_atan && act, gbase);
                        const uint64_t below = mv & ((1ull << gl) - 1ull);
                        const int src = below ? 63 - __clzll(below) : 0;
                        const double th_from = group_bcast<G>(th_gl, src);
                        if (!use_atan) {
                            th_gl = below ? th_from : theta_carry;
                            th_cl = th_gl - th_ref;
                        }
                        double sn, cs;
                        rp_sincos(th_cl, &sn, &cs);
                        const double sc = rp_rcp(cs);
                        cosT = use_atan ? cosT : cs;
                        secT = use_atan ? secT : sc;
                        tanT = use_atan ? tanT : sn * sc;
                    }
                    // cos / sin of the heading theta = theta_ref + theta_cl from the profile's cos / sin of theta_ref (variants with a
                    // collision query; the others need them for one lane per candidate only, where the extension starts)
                    const double sinT = tanT * cosT;
                    cos_gl = heading_cos(pf.f[PF_COS_REF], pf.f[PF_SIN_REF], cosT, sinT);
                    sin_gl = heading_sin(pf.f[PF_COS_REF], pf.f[PF_SIN_REF], cosT, sinT);
                    RP_STAMP(5);   // atan + carry

                    // -- curvature, velocity, acceleration (:883-896)
                    const double oneKrD = 1.0 - k_r * d;
                    const double q = cosT * rp_rcp(oneKrD);
                    const double kterm = frenet_kterm(k_r_d, d, k_r, dp);
                    kappa = frenet_kappa(dpp, kterm, tanT, cosT, q, k_r);
                    const double f = oneKrD * secT;
                    v = sd * f;
                    acc = frenet_acc(sdd, f, sd, secT, oneKrD, tanT, kappa, k_r, kterm);

                    // -- previous-step values for the finite differences (DPP lane shift)
                    double th_prev = lane_prev(th_gl), ka_prev = lane_prev(kappa);
                    if (gl == 0) { th_prev = theta_carry; ka_prev = kappa_carry; }
                    const double dth = i > 0 ? th_gl - th_prev : 0.0;
                    kdot = i > 0 ? kappa - ka_prev : 0.0;   // np.append([0], np.diff(kappa_gl)), :923
                    RP_STAMP(6);   // kappa, v, a + prev-step shifts

                    // -- _check_constraints, reactive_planner.py:971-1017 (order: velocity, kappa, yaw, kappa_dot, acc)
    #if RP_BRANCHFREE_CONSTRAINTS
                    // (straight-line code: every lane evaluates the five tests and the first failing one in the reference's order is
                    //  picked by selects -- as an if / else-if chain the compiler emitted five nested exec-mask regions, each with its
                    //  own scalar loads of the limits and a wait on them)
                    uint32_t reason = RP_REASON_NONE;
                    {
                        const double a_max = al.a_max, v_switch = al.v_switch, kappa_max = al.kappa_max;
                        const double wk = al.wheelbase * kappa;
                        // |round(yaw, 5)| > kappa_max v           with yaw = dth / dt           (:993-995)
                        const bool bad_yaw = fabs(rint(dth * al.c_yaw)) > kappa_max * v * 1e5;
                        // |dka / dt| > v_delta_max / (wb cos^2(atan(wb kappa)))                  (:1001-1005)
                        const bool bad_kd = fabs(kdot) > al.c_kdot * __builtin_fma(wk, wk, 1.0);
                        // a_min <= a <= a_max (v_switch / v above the switching velocity)        (:1011-1014)
                        const bool fast = v > v_switch;
                        const double acc_l = fast ? acc * v : acc, acc_r = fast ? a_max * v_switch : a_max;
                        const bool bad_acc = !((-a_max <= acc) & (acc_l <= acc_r));
                        const bool bad_v = v < -RP_EPS, bad_k = fabs(kappa) > kappa_max;
                        reason = (((cm & RP_CHECK_ACCELERATION) != 0) & bad_acc) ? RP_REASON_ACCELERATION : reason;
                        reason = (((cm & RP_CHECK_KAPPA_DOT) != 0) & bad_kd) ? RP_REASON_KAPPA_DOT : reason;
                        reason = (((cm & RP_CHECK_YAW_RATE) != 0) & bad_yaw) ? RP_REASON_YAW_RATE : reason;
                        reason = (((cm & RP_CHECK_KAPPA) != 0) & bad_k) ? RP_REASON_KAPPA : reason;
                        reason = (((cm & RP_CHECK_VELOCITY) != 0) & bad_v) ? RP_REASON_VELOCITY : reason;
                        reason = act ? reason : RP_REASON_NONE;
                    }
    #else
                    uint32_t reason = RP_REASON_NONE;
                    if (act) {
                        const double wk = al.wheelbase * kappa;
                        // |round(yaw, 5)| > kappa_max v           with yaw = dth / dt           (:993-995)
                        const bool bad_yaw = fabs(rint(dth * al.c_yaw)) > al.kappa_max * v * 1e5;
                        // |dka / dt| > v_delta_max / (wb cos^2(atan(wb kappa)))                  (:1001-1005)
                        const bool bad_kd = fabs(kdot) > al.c_kdot * __builtin_fma(wk, wk, 1.0);
                        // a_min <= a <= a_max (v_switch / v above the switching velocity)        (:1011-1014)
                        const bool ok_acc = (-al.a_max <= acc) && (v > al.v_switch ? acc * v <= al.a_max * al.v_switch : acc <= al.a_max);
                        if ((cm & RP_CHECK_VELOCITY) && v < -RP_EPS) reason = RP_REASON_VELOCITY;
                        else if ((cm & RP_CHECK_KAPPA) && fabs(kappa) > al.kappa_max) reason = RP_REASON_KAPPA;
                        else if ((cm & RP_CHECK_YAW_RATE) && bad_yaw) reason = RP_REASON_YAW_RATE;
                        else if ((cm & RP_CHECK_KAPPA_DOT) && bad_kd) reason = RP_REASON_KAPPA_DOT;
                        else if ((cm & RP_CHECK_ACCELERATION) && !ok_acc) reason = RP_REASON_ACCELERATION;
                    }
    #endif
                    if (__any(reason != RP_REASON_NONE)) {   // wave-uniform
                        const uint64_t fm = group_ballot<G>(reason != RP_REASON_NONE, gbase);
                        const int fl = fm ? __ffsll((unsigned long long)fm) - 1 : 0;
                        const uint32_t r_first = (uint32_t)__shfl((int)reason, fl, G);
                        if (fm && fail_step < 0) { fail_step = base + fl; fail_reason = r_first; }
                        if (!draw && fail_step >= 0) alive = false;
                        if (__ballot(alive) == 0) break;   // wave-uniform: every candidate of this wave is decided
                    }
                    RP_STAMP(7);   // constraints + first-failure vote

                    // -- (s, d) -> (x, y) = foot point + d * unit normal, reactive_planner.py:908-917
                    const bool in_dom = s_in_dom && fabs(d) <= al.proj_d_limit;
                    x = px + d * nx; y = py + d * ny;
                    if (__any(act && !in_dom) || ood_step >= 0) {   // wave-uniform ("ood_step" alone is group-uniform: harmless)
                        const uint64_t om = group_ballot<G>(act && !in_dom, gbase);
                        if (om && ood_step < 0) ood_step = base + __ffsll((unsigned long long)om) - 1;
                        if (ood_step >= 0 && i >= ood_step) { x = 0.0; y = 0.0; }   // x, y stay np.zeros past the break
                    }
                    if (act) {   // Cartesian rows of valid steps are final here
                        if (!LATE_STORE && store_ok) {
                            const uint32_t off8 = lane_off8 + (uint32_t)i * 8u;
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_X), x);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_Y), y);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA), th_gl);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_V), v);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_A), acc);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA), kappa);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA_DOT), kdot);
                            st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA_CL), th_cl);
                        }
                        cost_acc += cost_terms(i, acc, v, s, d, th_cl);
                    }
                    // static shapes: the pose's cell of the grid over them (static_grid_mask) is requested here for the valid steps,
                    // whose pose is final, and behind the extension for the extended ones -- and waited for IN FRONT of the row stores.
                    // Loads and stores share one in-order counter (vmcnt): a load that is waited for behind the write-through stores
                    // of the rows is waited for together with their acknowledgements from memory (6 000 cycles per step block, measured).
                    if (COLL == 2 && live && alive && fail_step < 0 && ood_step < 0 && !collide && act)   // (= cell_wanted below)
                        near_sta_cell = static_grid_mask(al.obs.grid, al.obs.gx0, al.obs.gy0, al.obs.ginv, al.obs.gnx, al.obs.gny,
                                                         x + al.wb_rear_axle * cos_gl, y + al.wb_rear_axle * sin_gl);
                }
                const bool cell_wanted = COLL == 2 && live && alive && fail_step < 0 && ood_step < 0 && !collide;
                RP_STAMP(8);   // x, y + stores of valid steps

                // -- horizon extension (trajectories.py:168-197, 302-332); only chunks that hold states >= L
                const int ll = L - 1 - base;
                const bool take = (ll >= 0 && ll < G);     // this chunk holds the last valid state
                if (__any(take && L < n)) {                // wave-uniform: park the last valid state in LDS
                    if (!COLL) rp_sincos(th_gl, &sin_gl, &cos_gl);
                    if (take && gl == ll) {
                        double *o = gs_last;
                        o[0] = x; o[1] = y; o[2] = th_gl; o[3] = v; o[4] = acc; o[5] = kappa; o[6] = kdot;
                        o[7] = s; o[8] = d; o[9] = th_cl; o[10] = sd; o[11] = sdd; o[12] = dd; o[13] = ddd;
                        o[14] = cos_gl; o[15] = sin_gl;
                    }
                }
                if (!ONE_CHUNK && c + 1 < nchunks) {
                    theta_carry = group_bcast<G>(th_gl, G - 1);
                    kappa_carry = group_bcast<G>(kappa, G - 1);
                }
                if (__any(live && !act)) {                 // wave-uniform: this chunk holds extended states
                    double termx = 0.0, termy = 0.0;
                    const double *o = gs_last;             // same-wave LDS write above is ordered before these reads
                    const double tk = (double)(i - L + 1) * dt;   // np.arange(1, steps + 1) * dt
                    double vt = 0.0;
                    if (!act) {   // i >= L: extended state
                        vt = o[3] + tk * o[4];                         // :182 (a[-1] already holds a[last])
                        vt = vt * (vt >= 0.0 ? 1.0 : 0.0);             // :184
                        termx = dt * vt * o[14];                       // :195-196
                        termy = dt * vt * o[15];
                    }
                    const double scx = group_scan<G>(termx) + cumx, scy = group_scan<G>(termy) + cumy;   // np.cumsum
                    if (!act) {
                        x = o[0] + scx;
                        y = o[1] + scy;
                        th_gl = o[2];                                  // :188
                        cos_gl = o[14];
                        sin_gl = o[15];
                        const double sv = o[10] + tk * 0.0;            // :313, s_ddot[-1] is the zero padding
                        const double e_sd = sv * (sv >= 0.0 ? 1.0 : 0.0);   // :315
                        const double e_dd = o[12] + tk * 0.0;          // :319
                        const double e_s = o[7] + tk * o[10];          // :330
                        const double e_d = o[8] + tk * o[12];          // :331
                        if (live) cost_acc += cost_terms(i, o[4], vt, e_s, e_d, o[9]);
                        if (LATE_STORE) {   // what the step block's store below writes for this lane
                            v = vt; acc = o[4]; kappa = o[5]; kdot = o[6];
                            s = e_s; d = e_d; th_cl = o[9];
                            sd = e_sd; sdd = o[11]; dd = e_dd; ddd = o[13];
                        } else if (store_ok && live) {
                            {
                                const uint32_t off8 = lane_off8 + (uint32_t)i * 8u;
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_X), x);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_Y), y);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA), th_gl);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_V), vt);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_A), o[4]);            // :179
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA), o[5]);        // :190
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA_DOT), o[6]);    // :192
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_S), e_s);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_D), e_d);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA_CL), o[9]);     // :327
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DOT), e_sd);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DDOT), o[11]);      // :323
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DOT), e_dd);
                                st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DDOT), o[13]);      // :324
                            }
                        }
                    }
                    if (!ONE_CHUNK && c + 1 < nchunks) {
                        cumx = group_bcast<G>(scx, G - 1);
                        cumy = group_bcast<G>(scy, G - 1);
                    }
                }
                if (COLL == 2) {
                    if (cell_wanted && !act)
                        near_sta_cell = static_grid_mask(al.obs.grid, al.obs.gx0, al.obs.gy0, al.obs.ginv, al.obs.gnx, al.obs.gny,
                                                         x + al.wb_rear_axle * cos_gl, y + al.wb_rear_axle * sin_gl);
                    uint32_t c_lo = (uint32_t)near_sta_cell, c_hi = (uint32_t)(near_sta_cell >> 32);
                    asm volatile("" : "+v"(c_lo), "+v"(c_hi));   // the cell has arrived before the first row store is issued
                    near_sta_cell = ((uint64_t)c_hi << 32) | c_lo;
                }
                // (lanes in the padding behind step N store too: their values are never read, and the run of the lane group
                //  then ends on a line boundary)
                if (SPLIT_OK && tail_m > 0 && base >= tail_m) {   // (wave-uniform) the partial step block: two rows to a 128-byte line
                    if (store_ok) {
                        char *const tbase = obase + (size_t)(lane_off8 + (uint32_t)(RP_N_ARRAYS * tail_m) * 8u + (uint32_t)gl * 8u);
                        const bool lo = gl < 8;
                        auto pair_store = [&](int q, double r0, double r1) {   // lanes 0 .. 7: row 2q, lanes 8 .. 15: row 2q + 1 of lanes 0 .. 7
                            const double up = dpp_f64<DPP_ROW_SHR8, 0xf>(r1);
                            st_row<RP_WT, RP_NT>(reinterpret_cast<double *>(tbase + q * 128), lo ? r0 : up);
                        };
                        pair_store(0, x, y); pair_store(1, th_gl, v); pair_store(2, acc, kappa); pair_store(3, kdot, s);
                        pair_store(4, d, th_cl); pair_store(5, sd, sdd); pair_store(6, dd, ddd);
                    }
                } else
                if (LATE_STORE && store_ok && i < ns) {   // (whole 128-byte lines: completing only the 64-byte half that holds step N
                                                          //  costs 15-20 % on the large batches -- cfg5 2.96 vs 2.51 ms)
                    const uint32_t off8 = lane_off8 + (uint32_t)i * 8u;
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_X), x);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_Y), y);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA), th_gl);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_V), v);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_A), acc);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA), kappa);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_KAPPA_DOT), kdot);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_S), s);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_D), d);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_THETA_CL), th_cl);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DOT), sd);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_S_DDOT), sdd);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DOT), dd);
                    st_row<RP_WT, RP_NT>(row_at(off8, RP_D_DDOT), ddd);
                }
                RP_STAMP(9);   // extension + scans + stores of the step block

                // -- eager collision query for every pose, reactive_planner.py:1033-1046
                if (COLL) {
                    // (a candidate that has collided in an earlier step block is not asked again: the label is all the reference keeps
                    //  of a collision, and the kinematic checks -- which can still turn it into INFEASIBLE_KINEMATIC -- go on)
                    const bool want = live && alive && fail_step < 0 && ood_step < 0 && !collide;
                    bool hit = false;
                    // With the broad-phase masks of the (pair, step) profile a pose whose masks are empty has nothing to be
                    // tested against: the whole query (sin / cos of the heading, ego rectangle, walks) is skipped unless some
                    // lane of the wavefront has a bit set.  (Few obstacles: most wavefronts skip it at most steps.)
                    constexpr bool masked = !COEFFS_IN;   // (the host keeps use_near_mask set for grid plans with obstacles; a pair
                                                          //  without a bound gets all-ones masks)
                    // dynamic obstacles: the (pair, step) mask of the profile; static shapes: the pose's own cell of the grid over them
                    const uint64_t near_dyn = double_as_mask(pf.f[PF_NEAR]);
                    const double ego_cx = x + al.wb_rear_axle * cos_gl, ego_cy = y + al.wb_rear_axle * sin_gl;
                    const uint64_t near_sta = want ? near_sta_cell : 0;
                    // (explicit polynomials have no (pair, step) masks of dynamic obstacles: those are all tested; static shapes go
                    //  through the grid for every kind of plan)
                    const bool ask = want && ((!masked && al.obs.n_dyn > 0) || (near_dyn | near_sta) != 0);
                    RP_STAMP(44);
#ifdef RP_WALK_COUNT
                    if (a.debug && !a.single_index && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) rp_walk_dbg = a.debug + 40;
#endif
                    if (__any(ask)) {   // wave-uniform; every lane runs the query code (wave-level culling inside)
                        ObsTables ob;   // (copied out of the constant address space, member by member)
                        ob.sobb = al.obs.sobb; ob.tri = al.obs.tri; ob.circ = al.obs.circ; ob.dyn = al.obs.dyn;
                        ob.clus = al.obs.clus; ob.clus_info = al.obs.clus_info;
                        ob.n_sobb = al.obs.n_sobb; ob.n_tri = al.obs.n_tri; ob.n_circ = al.obs.n_circ; ob.n_dyn = al.obs.n_dyn;
                        ob.n_steps = al.obs.n_steps; ob.dyn_t0 = al.obs.dyn_t0; ob.n_clus = al.obs.n_clus; ob.clus_per = al.obs.clus_per;
                        ob.dyn_rmax_all = al.obs.dyn_rmax_all;
                        const Obb ego = {ego_cx, ego_cy, cos_gl, sin_gl, al.half_length, al.half_width};
                        RP_STAMP(45);
                        hit = pose_collides<masked, COLL == 2, LON_FUSED && COLL == 2, true, !MAT>(ob, ego, al.ego_radius, al.time_step0 + i * al.factor, ask,
                                                                                             near_dyn, near_sta, slot_lds) && ask;
                    }
                    RP_STAMP(46);
                    collide |= group_ballot<G>(hit, gbase) != 0;
                }
                RP_STAMP(11);  // collision
            }
        }
        RP_STAMP(12);  // state stores issued, chunk loop done
        if (MAT && STAGE_OUT) {   // linear copy-out of this wavefront's candidates (wave-uniform)
            // blocks are written for every candidate that passed the pre-filter; failed candidates keep
            // whatever steps were computed (same contract as direct stores)
            const uint64_t okmask = __ballot(valid && pre_reason == RP_REASON_NONE);
            if (okmask) {
                const int blk2 = (RP_N_ARRAYS * n) >> 1;             // double2 pieces per candidate (14 n is even)
                const double2 *src = reinterpret_cast<const double2 *>(lds_out + (size_t)(wave_in_block * GPW) * RP_N_ARRAYS * (size_t)n);
                double2 *dst = reinterpret_cast<double2 *>(al.states + ((size_t)w0 * RP_N_ARRAYS) * (size_t)n);
                for (int k = lane; k < GPW * blk2; k += 64) {
                    const int cand = k / blk2;
                    if ((okmask >> (cand * G)) & 1ull) RP_COPY_OUT_STORE(dst + k, src[k]);
                }
            }
        }

        // ---- label, reason, cost (the group's LAST lane holds the reduced cost and does the bookkeeping) ----
        uint32_t status;
        const bool decided_bad = pre_reason != RP_REASON_NONE || fail_step >= 0 || ood_step >= 0;
        if (pre_reason != RP_REASON_NONE) {
            status = RP_LABEL_NONE | (pre_reason << 4);
        } else if (fail_step >= 0) {   // :902 INFEASIBLE_KINEMATIC
            status = RP_LABEL_INFEASIBLE_KINEMATIC | (fail_reason << 4) | ((uint32_t)fail_step << 8);
        } else if (ood_step >= 0) {    // :915; draw mode stores it as INFEASIBLE_KINEMATIC (:942-943)
            status = (draw ? RP_LABEL_INFEASIBLE_KINEMATIC : RP_LABEL_NONE) | (RP_REASON_OUT_OF_DOMAIN << 4) |
                     ((uint32_t)ood_step << 8);
        } else {
            status = collide ? RP_LABEL_INFEASIBLE_COLLISION : RP_LABEL_FEASIBLE;
        }
        double cost = group_sum_last<G>(cost_acc);
        if (decided_bad) cost = __builtin_nan("");
        if (valid && gl == G - 1) {
            if (!LON_FUSED && al.index_list) {   // a round of the lazy stage: the label of the candidate itself; its cost stays the first pass's
                const int64_t fs = gidx - al.cand_begin;
                al.status[fs] = status;
                cost = decided_bad ? cost : al.cost[fs];
            } else if (sweep) {   // the candidate's label; its cost stays pass 1's; a free candidate lowers the bound for everyone behind it
                al.status[slot] = status;
                cost = decided_bad ? cost : al.cost[slot];
                if (status == RP_LABEL_FEASIBLE && cost == cost)
                    __hip_atomic_fetch_min(al.sweep_bound, cost_key(cost), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                al.status[slot] = status;
                al.cost[slot] = cost;
            }
            const uint32_t lab = RP_STATUS_LABEL(status), rs = RP_STATUS_REASON(status);
            if (rs) atomicAdd(&sh_cnt[2 + rs], 1);
            if (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION) atomicAdd(&sh_cnt[0], 1);
            if (lab == RP_LABEL_INFEASIBLE_COLLISION) atomicAdd(&sh_cnt[1], 1);
            if (lab == RP_LABEL_FEASIBLE && cost == cost &&
                better(cost, gidx, sh_best_cost[grp], (int64_t)sh_best_idx[grp])) {   // slot owned by this group
                sh_best_cost[grp] = cost;
                sh_best_idx[grp] = gidx;
            }
        }
    }
    RP_STAMP(13);  // cost reduction + status written

    // ---- block partial: lexicographic (cost, index) min + counters ----
    if (al.partials) {
        __syncthreads();
        if (tid < 64) {   // first wavefront: the slots of the lane groups, the counters lane by lane
            double bc = tid < GPB ? sh_best_cost[tid < GPB ? tid : 0] : 0.0;
            long long bi = tid < GPB ? sh_best_idx[tid < GPB ? tid : 0] : -1;
            if (GPB >= 16) {
                wave_min_pair(bc, bi);   // one DPP reduction instead of a 16-trip walk by one lane (cfg2: 2.1 k -> 1.4 k cycles)
            } else {                     // 4 or 8 slots: the walk is shorter than the reduction (cfg5 draw: 4 candidates per workgroup)
                bc = 0.0; bi = -1;
#pragma unroll
                for (int k = 0; k < GPB; ++k)
                    if (sh_best_idx[k] >= 0 && better(sh_best_cost[k], (int64_t)sh_best_idx[k], bc, (int64_t)bi)) { bc = sh_best_cost[k]; bi = sh_best_idx[k]; }
            }
            const Partials bp = partials_at(al.partials, al.partials_cap, al.partials_first + (int)blockIdx.x);
            if (tid == 0) { bp.cost[0] = bi >= 0 ? bc : 0.0; bp.idx[0] = bi; }
            if (tid >= 8 && tid < 8 + RP_PARTIAL_CNT) bp.cnt[tid - 8] = (uint32_t)sh_cnt[tid - 8];
        }
    }
    RP_STAMP(14);
    RP_TL(1);
    if (al.single_index && al.host_seq) {   // winner re-evaluation (one workgroup): ticket after the state rows
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(al.host_seq, al.seq_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// rp_cost_kernel -- costs-only evaluation of LARGE grid batches, ONE LANE PER CANDIDATE, the lane walks its candidate's steps.
//
// What rp_eval_kernel's layout (a group of 16 lanes per candidate, lane = time step) pays for the time axis lying across lanes --
// previous-step values through DPP shifts, first-failure votes through ballots and ds_bpermute, the horizon extension's parked
// state and scans, group-uniform values parked in LDS and re-read, lanes behind step N idle, per-candidate bookkeeping four
// times per wavefront -- buys coalesced state ROWS.  A plan that keeps no rows (production mode: 12 B per candidate leave the
// kernel) does not need that: with the time axis inside the lane the previous step is a register, the first failure is the
// step the loop is at, the extension reads registers, and what is uniform over a (T, longitudinal sample) pair -- the whole
// profile step -- is the same for (nearly) all lanes of the wavefront.  ~165 vector instructions per (wavefront, step) for 64
// candidates against ~290 per step block of 16 steps for 4.
// Arithmetic: expression for expression that of rp_eval_kernel (same labels, reasons, first failing steps, costs).  The cost is
// summed in rp_eval_kernel's order too -- sixteen partial sums by step mod 16 (one lane of the group each, there), combined in
// the order of group_sum_last<16> -- so the two kernels agree bit for bit.
// Grid plans without the collision query (no obstacles, RP_FLAG_SKIP_COLLISION, or the first pass of the cost-ordered stage),
// no state rows, two-kernel path (profiles from rp_lon_kernel).  One wavefront per workgroup; slot = blockIdx.x * 64 + lane.
// ------------------------------------------------------------------------------------------------
#define RP_COST_BLOCK 64
#ifndef RP_COST_WAVES
#define RP_COST_WAVES 4   // 128 registers (ten of them spilled): cfg5 442 -> 411 us, cfg4 316 -> 295 us against three wavefronts per SIMD
#endif
// LOW: low-velocity mode (the lateral polynomial runs over the arc length, reactive_planner.py:756-772) -- launch-uniform, a template
// parameter so that the step loop carries no selects for it
// COLL: 0 no collision query, 1 dynamic obstacles only, 2 static shapes as well (as rp_eval_kernel's): the eager query for every
// pose of every candidate (reactive_planner.py:1033-1046) -- the lanes of a wavefront stand at the SAME step, so the (pair, step)
// mask, the obstacle rows and the walk over them are (nearly) uniform over the wavefront.
template <bool LOW, int COLL>
__global__ __launch_bounds__(RP_COST_BLOCK, COLL ? RP_WAVES_PER_SIMD : RP_COST_WAVES) void rp_cost_kernel(const KArgsG ag) {
    const KArgs &a = ag.k;
    touch_kernargs<10>();
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;
    const int lane = threadIdx.x;
    const int64_t count = a.count;
    const int64_t w0 = (int64_t)blockIdx.x * RP_COST_BLOCK;
    if (a.lazy_ctl && blockIdx.x == 0 && lane < (int)(sizeof(LazyCtl) / 8))   // (first pass of the lazy stage)
        reinterpret_cast<unsigned long long *>(a.lazy_ctl)[lane] = 0ull;
    if (a.sweep_init && blockIdx.x == 0 && lane == 0) __hip_atomic_store(a.sweep_init, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (pass 1 of the bounded sweep)
    const int64_t slot = w0 + lane;
    const bool valid = slot < count;
    const int64_t gidx = a.cand_begin + (valid ? slot : w0);   // (lanes behind the batch shadow the wavefront's first candidate)
    // ---- candidate: lateral sample, pair header, lateral polynomial (sampling.py:226-238, 268-270)
    const uint32_t g32 = (uint32_t)gidx, nd = (uint32_t)a.nD;
    const uint32_t p32 = g32 / nd;
    const double d_target = grid_base(a)[a.nT + a.nL + (int)(g32 - p32 * nd)];
    const int64_t pair_slot = (int64_t)p32 - a.pair_begin;
    const PairHdr h = a.pair_hdr[pair_slot];
    const int L = h.L;
    const double s0 = h.s0;
    const uint32_t pre_reason = (uint32_t)h.pre_reason;
    const Poly lat = quintic_coeffs(a.x0_lat[0], a.x0_lat[1], a.x0_lat[2], d_target, 0.0, 0.0, h.lat_T);
    double P[16];
    park_poly(P, lat);
    const int N = a.N, n = N + 1;
    const double dt = a.dt;
    constexpr bool low = LOW;
    const uint32_t cm = a.constraint_mask;
    const int mid = n / 2;
    // Horner steps as three-address v_fma_f64 (rp_math.h: rp_fma3): the coefficients stay where they are -- the compiler's two-address
    // v_fmac_f64 wants a copy of the coefficient per step (13 v_mov_b64 per time step).  Same operations as poly_pos / poly_vel / poly_acc.
    auto p_pos = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(rp_fma3(rp_fma3(P[5], t, P[4]), t, P[3]), t, P[2]), t, P[1]), t, P[0]); };
    auto p_vel = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(rp_fma3(P[10], t, P[9]), t, P[8]), t, P[7]), t, P[6]); };
    auto p_acc = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(P[14], t, P[13]), t, P[12]), t, P[11]); };

    auto cost_terms = [&](int i, double acc, double v, double s, double d, double th_cl) -> double {   // (rp_eval_kernel's, word for word)
        double e, cst;
        e = a.w_a * acc; cst = e * e;
        e = 0.25 * (a.desired_d - d); cst = __builtin_fma(e, e, cst);
        e = 0.25 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
        if (a.has_speed) { e = 5.0 * (v - a.desired_speed); cst = __builtin_fma(e, e, cst); }
        if (a.has_s) { e = 0.25 * (a.desired_s - s); cst = __builtin_fma(e, e, cst); }
        if (i == N) {
            e = 20.0 * (a.desired_d - d); cst = __builtin_fma(e, e, cst);
            e = 5.0 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
            if (a.has_speed) { e = v - a.desired_speed; cst += 50.0 * (e * e); }
            if (a.has_s) { e = 20.0 * (a.desired_s - s); cst = __builtin_fma(e, e, cst); }
        }
        if (i == mid && a.has_speed) { e = v - a.desired_speed; cst += 100.0 * (e * e); }
        return cst;
    };

    int fail_step = -1, ood_step = -1;
    uint32_t fail_reason = RP_REASON_NONE;
    bool collide = false;
    // the sixteen partial sums of the cost (step mod 16: what the sixteen lanes of a group hold in rp_eval_kernel) live in LDS,
    // [k][lane]: the step loop touches one of them per step, by a wave-uniform k (in registers a run-time k costs a chain of 64
    // selects per step, unrolled sixteen-fold the loop does not fit the instruction cache)
    __shared__ double sh_csum[16][RP_COST_BLOCK];
    typedef double __attribute__((address_space(3))) *lds_double;
    const lds_double csum = (lds_double)&sh_csum[0][lane];
#pragma unroll
    for (int k = 0; k < 16; ++k) csum[k * RP_COST_BLOCK] = 0.0;
    bool alive = valid && pre_reason == RP_REASON_NONE;
    if (__ballot(alive) != 0) {   // wave-uniform
        double th_prev = a.x0_orientation, ka_prev = 0.0;
        // last valid state (step L - 1) for the horizon extension (trajectories.py:168-197, 302-332)
        double l_v = 0.0, l_acc = 0.0, l_s = 0.0, l_d = 0.0, l_thcl = 0.0, l_sd = 0.0, l_dd = 0.0;
        double l_x = 0.0, l_y = 0.0, l_cos = 1.0, l_sin = 0.0, cumx = 0.0, cumy = 0.0;   // (COLL: poses of the extended steps)
        // profile rows: wave-uniform row bases (scalar registers) + one 32-bit byte offset per lane (its pair's block).  (The offset is from
        // the batch's first pair: the host takes this kernel only while the profile rows of a launch stay below 4 GiB -- cost_rows_ok.
        // Relative to the wavefront's first pair, as rp_chunk_kernel has it, measured 5 % slower here: cfg5 368 -> 389 us.)
        const uint32_t voff = (uint32_t)pair_slot * (uint32_t)PF_FIELDS * (uint32_t)n * 8u;
        const char *const pb = reinterpret_cast<const char *>(a.profile);
        const size_t n8 = (size_t)n * 8;
#pragma nounroll
        for (int i = 0; i <= N; ++i) {   // wave-uniform
            const bool act = i < L;
            const lds_double cs_k = csum + (i & 15) * RP_COST_BLOCK;
            const char *const ob = pb + (size_t)i * 8;
            auto fld = [&](int k) -> double { return *reinterpret_cast<const double *>(ob + (size_t)k * n8 + voff); };
            double pose_x = 0.0, pose_y = 0.0, pose_cos = 1.0, pose_sin = 0.0;   // (COLL) rear-axle pose of this lane at step i
            if (__any(act)) {   // (wave-uniform) some lane's step i is a valid one
                // -- the pair's profile at this step (the lanes of a pair read the same words)
                double s = fld(PF_S), sd = fld(PF_SD), sdd = fld(PF_SDD);
                const double inv_sd = fld(PF_INV_SD), th_ref = fld(PF_TH_REF);
                const double k_r = fld(PF_KR), k_r_d = fld(PF_KRD);
                const bool s_in_dom = fld(PF_INDOM) != 0.0;
                // -- lateral polynomial, reactive_planner.py:756-777
                const double t = (double)i * dt;
                const double tau = low ? s - s0 : t;
                double d = p_pos(tau), dd = p_vel(tau), ddd = p_acc(tau);
                if (fabs(dd) < RP_EPS) dd = 0.0;
                // -- d', d'' (:810-832)
                const bool moving = inv_sd > 0.0;
                double dp, dpp;
                if (!low) {
                    dp = dd * inv_sd;
                    const double ddot = ddd - dp * sdd;
                    dpp = ddot * inv_sd * inv_sd;
                } else {
                    dp = dd;
                    dpp = ddd;
                }
                // -- orientations (:842-873) incl. the standstill carry of :866
                const bool use_atan = moving || low;
                double th_cl = rp_atan(dp);
                double th_gl = th_cl + th_ref;
                const double w2 = __builtin_fma(dp, dp, 1.0);
                double cosT = rp_rsqrt(w2);
                double secT = w2 * cosT;
                double tanT = dp;
                if (!low && __any(act && !use_atan)) {   // standstill lanes keep the orientation of the step before (:866)
                    if (!use_atan) {
                        th_gl = th_prev;
                        th_cl = th_gl - th_ref;
                    }
                    double sn, cs;
                    rp_sincos(th_cl, &sn, &cs);
                    const double sc = rp_rcp(cs);
                    cosT = use_atan ? cosT : cs;
                    secT = use_atan ? secT : sc;
                    tanT = use_atan ? tanT : sn * sc;
                }
                // -- curvature, velocity, acceleration (:883-896)
                const double oneKrD = 1.0 - k_r * d;
                const double q = cosT * rp_rcp(oneKrD);
                const double kterm = frenet_kterm(k_r_d, d, k_r, dp);
                double kappa = frenet_kappa(dpp, kterm, tanT, cosT, q, k_r);
                const double f = oneKrD * secT;
                double v = sd * f;
                double acc = frenet_acc(sdd, f, sd, secT, oneKrD, tanT, kappa, k_r, kterm);
                const double dth = i > 0 ? th_gl - th_prev : 0.0;
                const double kdot = i > 0 ? kappa - ka_prev : 0.0;
                // -- _check_constraints, reactive_planner.py:971-1017 (order: velocity, kappa, yaw, kappa_dot, acc)
                uint32_t reason = RP_REASON_NONE;
                {
                    const double a_max = a.a_max, v_switch = a.v_switch, kappa_max = a.kappa_max;
                    const double wk = a.wheelbase * kappa;
                    const bool bad_yaw = fabs(rint(dth * a.c_yaw)) > kappa_max * v * 1e5;
                    const bool bad_kd = fabs(kdot) > a.c_kdot * __builtin_fma(wk, wk, 1.0);
                    const bool fast = v > v_switch;
                    const double acc_l = fast ? acc * v : acc, acc_r = fast ? a_max * v_switch : a_max;
                    const bool bad_acc = !((-a_max <= acc) & (acc_l <= acc_r));
                    const bool bad_v = v < -RP_EPS, bad_k = fabs(kappa) > kappa_max;
                    reason = (((cm & RP_CHECK_ACCELERATION) != 0) & bad_acc) ? RP_REASON_ACCELERATION : reason;
                    reason = (((cm & RP_CHECK_KAPPA_DOT) != 0) & bad_kd) ? RP_REASON_KAPPA_DOT : reason;
                    reason = (((cm & RP_CHECK_YAW_RATE) != 0) & bad_yaw) ? RP_REASON_YAW_RATE : reason;
                    reason = (((cm & RP_CHECK_KAPPA) != 0) & bad_k) ? RP_REASON_KAPPA : reason;
                    reason = (((cm & RP_CHECK_VELOCITY) != 0) & bad_v) ? RP_REASON_VELOCITY : reason;
                    reason = act ? reason : RP_REASON_NONE;
                }
                if (reason != RP_REASON_NONE && fail_step < 0) { fail_step = i; fail_reason = reason; alive = false; }
                // -- out of the projection domain (:908-917): no reason counter, the kinematic verdict of a later step still counts
                const bool in_dom = s_in_dom && fabs(d) <= a.proj_d_limit;
                if (act && !in_dom && ood_step < 0) ood_step = i;
                if (COLL) {   // (s, d) -> (x, y) = foot point + d * unit normal (:908-917); heading from cos / sin of theta_ref and theta_cl
                    const double sinT = tanT * cosT;
                    pose_x = fld(PF_PX) + d * fld(PF_NX);
                    pose_y = fld(PF_PY) + d * fld(PF_NY);
                    pose_cos = heading_cos(fld(PF_COS_REF), fld(PF_SIN_REF), cosT, sinT);
                    pose_sin = heading_sin(fld(PF_COS_REF), fld(PF_SIN_REF), cosT, sinT);
                }
                if (act) {
                    *cs_k += cost_terms(i, acc, v, s, d, th_cl);
                    th_prev = th_gl;
                    ka_prev = kappa;
                }
                if (__any(i == L - 1)) {   // (wave-uniform; the lanes of a wavefront belong to one or two pairs: one or two such steps)
                    const bool lastv = i == L - 1;
                    l_v = lastv ? v : l_v; l_acc = lastv ? acc : l_acc; l_s = lastv ? s : l_s; l_d = lastv ? d : l_d;
                    l_thcl = lastv ? th_cl : l_thcl; l_sd = lastv ? sd : l_sd; l_dd = lastv ? dd : l_dd;
                    if (COLL) { l_x = lastv ? pose_x : l_x; l_y = lastv ? pose_y : l_y; l_cos = lastv ? pose_cos : l_cos; l_sin = lastv ? pose_sin : l_sin; }
                }
                if (__ballot(alive) == 0) break;   // wave-uniform: every candidate of this wavefront is decided
            }
            if (!act) {   // i >= L: extended state (cost_function.py sums over the extended arrays)
                const double tk = (double)(i - L + 1) * dt;   // np.arange(1, steps + 1) * dt
                double vt = l_v + tk * l_acc;                 // :182
                vt = vt * (vt >= 0.0 ? 1.0 : 0.0);            // :184
                const double e_s = l_s + tk * l_sd;           // :330
                const double e_d = l_d + tk * l_dd;           // :331
                *cs_k += cost_terms(i, l_acc, vt, e_s, e_d, l_thcl);
                if (COLL) {   // x[L:] = x[last] + cumsum(dt * v_tmp * cos(theta[last])), same for y (:195-196); heading held (:188)
                    cumx += dt * vt * l_cos;
                    cumy += dt * vt * l_sin;
                    pose_x = l_x + cumx; pose_y = l_y + cumy; pose_cos = l_cos; pose_sin = l_sin;
                }
            }
            // -- eager collision query for this pose, reactive_planner.py:1033-1046 (a candidate that has collided is not asked again:
            //    the label is all the reference keeps of a collision; the kinematic checks -- which can still turn it into
            //    INFEASIBLE_KINEMATIC -- go on)
            if (COLL) {
                const bool want = alive && fail_step < 0 && ood_step < 0 && !collide;
                const double ego_cx = pose_x + a.wb_rear_axle * pose_cos, ego_cy = pose_y + a.wb_rear_axle * pose_sin;
                const uint64_t near_dyn = want ? double_as_mask(fld(PF_NEAR)) : 0;
                uint64_t near_sta = 0;
                if (COLL == 2 && want)
                    near_sta = static_grid_mask(a.obs.grid, a.obs.gx0, a.obs.gy0, a.obs.ginv, a.obs.gnx, a.obs.gny, ego_cx, ego_cy);
                const bool ask = want && (near_dyn | near_sta) != 0;
                if (__any(ask)) {   // wave-uniform; every lane runs the query code (wave-level culling inside)
                    const Obb ego = {ego_cx, ego_cy, pose_cos, pose_sin, a.half_length, a.half_width};
                    collide |= pose_collides<true, COLL == 2, false, true>(a.obs, ego, a.ego_radius, a.time_step0 + i * a.factor, ask, near_dyn, near_sta) && ask;
                }
            }
        }
    }
    // ---- label, reason, cost
    uint32_t status;
    const bool decided_bad = pre_reason != RP_REASON_NONE || fail_step >= 0 || ood_step >= 0;
    if (pre_reason != RP_REASON_NONE) status = RP_LABEL_NONE | (pre_reason << 4);
    else if (fail_step >= 0) status = RP_LABEL_INFEASIBLE_KINEMATIC | (fail_reason << 4) | ((uint32_t)fail_step << 8);
    else if (ood_step >= 0) status = RP_LABEL_NONE | (RP_REASON_OUT_OF_DOMAIN << 4) | ((uint32_t)ood_step << 8);
    else status = collide ? RP_LABEL_INFEASIBLE_COLLISION : RP_LABEL_FEASIBLE;
    // the sixteen partial sums in the order of group_sum_last<16> (lane 15's tree: ror 8, 4, 2, 1)
    double cost;
    {
        double s8[8], s4[4], s2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] = csum[(8 + k) * RP_COST_BLOCK] + csum[k * RP_COST_BLOCK];
#pragma unroll
        for (int k = 0; k < 4; ++k) s4[k] = s8[4 + k] + s8[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) s2[k] = s4[2 + k] + s4[k];
        cost = s2[1] + s2[0];
    }
    if (decided_bad) cost = __builtin_nan("");
    if (valid) {
        a.status[slot] = status;
        a.cost[slot] = cost;
    }
    // ---- block partial: lexicographic (cost, index) min + counters of this wavefront
    if (a.partials) {
        const uint32_t lab = RP_STATUS_LABEL(status), rs = RP_STATUS_REASON(status);
        double bc = (valid && lab == RP_LABEL_FEASIBLE && cost == cost) ? cost : 0.0;
        long long bi = (valid && lab == RP_LABEL_FEASIBLE && cost == cost) ? (long long)gidx : -1;
        wave_min_pair(bc, bi);
        const Partials bp = partials_at(a.partials, a.partials_cap, a.partials_first + (int)blockIdx.x);
        const unsigned int n_feas = (unsigned int)__popcll(__ballot(valid && (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION)));
        unsigned int mine = 0;
#pragma unroll
        for (uint32_t r = 1; r < 8; ++r) {
            const unsigned int cnt = (unsigned int)__popcll(__ballot(valid && rs == r));
            mine = lane == (int)(2 + r) ? cnt : mine;
        }
        mine = lane == 0 ? n_feas : mine;   // [0] n_feasible, [1] n_collision, [2 + r] reasons
        const unsigned int n_coll = (unsigned int)__popcll(__ballot(valid && lab == RP_LABEL_INFEASIBLE_COLLISION));
        mine = lane == 1 ? n_coll : mine;
        if (lane < RP_PARTIAL_CNT) bp.cnt[lane] = mine;
        if (lane == 0) { bp.cost[0] = bi >= 0 ? bc : 0.0; bp.idx[0] = bi; }
    }
}

// ------------------------------------------------------------------------------------------------
// rp_chunk_kernel -- costs-only evaluation of MID-SIZED grid batches: ONE LANE PER (CANDIDATE, STEP BLOCK OF 16 STEPS), one WAVEFRONT
// per (64 candidates, step block).
//
// rp_cost_kernel (one lane per candidate) needs a wavefront of 64 candidates per SIMD -- four to hide its own instruction latency:
// 262 144 candidates -- before it beats the lanes-over-steps kernel; a batch the size of cfg3 (62 496 candidates, N = 60) leaves it
// one wavefront per SIMD, issuing an instruction every ten cycles.  This kernel keeps its instruction stream -- the time axis inside
// the lane: the previous step is a register, the first failure is the step the loop is at, the extension reads registers; what is
// uniform over a (T, longitudinal sample) pair is uniform over the wavefront -- and cuts the time axis of a candidate into its step
// blocks of sixteen (the blocks rp_eval_kernel's sixteen lanes walk one after the other): a workgroup of NB = ceil((N + 1) / 16)
// wavefronts evaluates 64 candidates, wavefront q their step block q, and cfg3 fills the chip with 3 906 wavefronts of seventeen
// iterations.
// (All lanes of a wavefront at the SAME step: a block of extended states runs the few instructions of the extension only, the (pair,
// step) masks of the collision query and the obstacle rows are wave-uniform.  Measured first, and dropped: the G blocks of a candidate
// in the lanes of ONE wavefront -- every iteration then runs the valid-step code AND the extension, the query walks the obstacles
// near four different time steps: cfg3 46 us without / 156 us with the eager query, against 52 / 95 us of the 16-lane kernel.)
// What crosses a block boundary:
//   * theta and kappa of the step in front of the block (yaw rate and kappa_dot are first differences, reactive_planner.py:923,993)
//     and the last valid state (step L - 1) for a block of extended states (trajectories.py:168-197, 302-332): every wavefront but the
//     first runs ONE iteration ahead of its block -- step min(16 q - 1, L - 1) without constraints, cost or collision query;
//     the standstill carry of the orientation (:866: a chain over the steps) is resolved there by looking the last moving step
//     up in the profile (whether a step moves depends on the pair alone);
//   * the first failing step, the first step out of the projection domain, "some pose collides": through LDS behind the loop, the
//     first block that has one / any block;
//   * the cost: rp_eval_kernel sums it as sixteen partial sums by step mod 16, each over the step blocks in ascending order, then the
//     tree of group_sum_last<16> -- here every lane leaves ITS block's sixteen terms in LDS and the workgroup's first wavefront forms
//     the partial sums over the blocks in ascending order: the same additions in the same order, the same bits (rp_cost_kernel does
//     the same inside a lane).
// Arithmetic: expression for expression that of rp_eval_kernel / rp_cost_kernel (same labels, reasons, first failing steps, costs).
// Grid plans, no state rows, two-kernel path (profiles from rp_lon_kernel), 17 <= N + 1 <= 112 (two to seven step blocks).
// ------------------------------------------------------------------------------------------------
#define RP_CHUNK_BLOCK 64    // candidates per workgroup (one wavefront per step block of theirs: G * 64 threads)
#define RP_CHUNK_STEPS 16
#define RP_CHUNK_MAX_BLOCKS 7   // step blocks (= wavefronts of a workgroup): 7 * (8 KB of cost terms + 768 B of keys) < 64 KB of LDS; N + 1 <= 112
__host__ __device__ constexpr size_t rp_chunk_lds_bytes(int nb) { return (size_t)nb * (RP_CHUNK_STEPS * RP_CHUNK_BLOCK * sizeof(double) + 3 * RP_CHUNK_BLOCK * sizeof(int)); }
template <bool LOW, int COLL>
#ifndef RP_CHUNK_WAVES
#define RP_CHUNK_WAVES RP_COST_WAVES
#endif
__global__ __launch_bounds__(RP_CHUNK_BLOCK * RP_CHUNK_MAX_BLOCKS, COLL ? RP_WAVES_PER_SIMD : RP_CHUNK_WAVES) void rp_chunk_kernel(const KArgsG ag) {
    const KArgs &a = ag.k;
    extern __shared__ double lds_chunk[];   // [NB][16][64] cost terms | [3][NB][64] int keys (rp_chunk_lds_bytes)
    const int NB = (int)(blockDim.x >> 6);  // step blocks per candidate = wavefronts of this workgroup (2 .. RP_CHUNK_MAX_BLOCKS)
    touch_kernargs<10>();
    if (level_gate_closed(a.gate, a.gate_seq, a.gate_level)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;                // this lane's candidate inside the workgroup
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wavefront's step block (wave-uniform)
    const int64_t count = a.count;
    const int64_t w0 = (int64_t)blockIdx.x * RP_CHUNK_BLOCK;
    if (a.lazy_ctl && blockIdx.x == 0 && tid < (int)(sizeof(LazyCtl) / 8))   // (first pass of the lazy stage)
        reinterpret_cast<unsigned long long *>(a.lazy_ctl)[tid] = 0ull;
    if (a.sweep_init && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.sweep_init, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (pass 1 of the bounded sweep)
    const int64_t slot = w0 + lane;
    const bool valid = slot < count;
    const int64_t gidx = a.cand_begin + (valid ? slot : w0);   // (lanes behind the batch shadow the wavefront's first candidate)
    // ---- candidate: lateral sample, pair header, lateral polynomial (sampling.py:226-238, 268-270)
    const uint32_t g32 = (uint32_t)gidx, nd = (uint32_t)a.nD;
    const uint32_t p32 = g32 / nd;
    const double d_target = grid_base(a)[a.nT + a.nL + (int)(g32 - p32 * nd)];
    const int32_t pair_slot = (int32_t)((int64_t)p32 - a.pair_begin);
    const int32_t pair_w = __builtin_amdgcn_readfirstlane(pair_slot);   // pair of the wavefront's first candidate: wave-uniform base
    const PairHdr h = a.pair_hdr[pair_slot];
    const int L = h.L;
    const double s0 = h.s0;
    const uint32_t pre_reason = (uint32_t)h.pre_reason;
    const Poly lat = quintic_coeffs(a.x0_lat[0], a.x0_lat[1], a.x0_lat[2], d_target, 0.0, 0.0, h.lat_T);
    double P[16];
    park_poly(P, lat);
    const int N = a.N, n = N + 1;
    const double dt = a.dt;
    constexpr bool low = LOW;
    const uint32_t cm = a.constraint_mask;
    const int mid = n / 2;
    auto p_pos = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(rp_fma3(rp_fma3(P[5], t, P[4]), t, P[3]), t, P[2]), t, P[1]), t, P[0]); };
    auto p_vel = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(rp_fma3(P[10], t, P[9]), t, P[8]), t, P[7]), t, P[6]); };
    auto p_acc = [&](double t) { return rp_fma3(rp_fma3(rp_fma3(P[14], t, P[13]), t, P[12]), t, P[11]); };

    auto cost_terms = [&](int i, double acc, double v, double s, double d, double th_cl) -> double {   // (rp_eval_kernel's, word for word)
        double e, cst;
        e = a.w_a * acc; cst = e * e;
        e = 0.25 * (a.desired_d - d); cst = __builtin_fma(e, e, cst);
        e = 0.25 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
        if (a.has_speed) { e = 5.0 * (v - a.desired_speed); cst = __builtin_fma(e, e, cst); }
        if (a.has_s) { e = 0.25 * (a.desired_s - s); cst = __builtin_fma(e, e, cst); }
        if (i == N) {
            e = 20.0 * (a.desired_d - d); cst = __builtin_fma(e, e, cst);
            e = 5.0 * fabs(th_cl); cst = __builtin_fma(e, e, cst);
            if (a.has_speed) { e = v - a.desired_speed; cst += 50.0 * (e * e); }
            if (a.has_s) { e = 20.0 * (a.desired_s - s); cst = __builtin_fma(e, e, cst); }
        }
        if (i == mid && a.has_speed) { e = v - a.desired_speed; cst += 100.0 * (e * e); }
        return cst;
    };

    int fail_step = -1, ood_step = -1;
    uint32_t fail_reason = RP_REASON_NONE;
    bool collide = false;
    // this lane's sixteen cost terms (one per step of its block), [block][j][candidate]: summed over the blocks behind the loop
    typedef double __attribute__((address_space(3))) *lds_double;
    typedef int __attribute__((address_space(3))) *lds_int;
    const lds_double csum = (lds_double)(lds_chunk + ((size_t)q * RP_CHUNK_STEPS) * RP_CHUNK_BLOCK + lane);
    const lds_int keys = (lds_int) reinterpret_cast<int *>(lds_chunk + (size_t)NB * RP_CHUNK_STEPS * RP_CHUNK_BLOCK);   // [3][NB][64]: fail, ood, collide
#pragma unroll
    for (int k = 0; k < RP_CHUNK_STEPS; ++k) csum[k * RP_CHUNK_BLOCK] = 0.0;
    const int i_first = q * RP_CHUNK_STEPS;          // first step of this wavefront's block
    const bool has_block = i_first <= N;             // (the host launches ceil((N + 1) / 16) wavefronts: every one has a step)
    bool alive = valid && pre_reason == RP_REASON_NONE && has_block;
    if (__ballot(alive) != 0) {   // wave-uniform
        double th_prev = a.x0_orientation, ka_prev = 0.0;
        // last valid state (step L - 1) for the horizon extension (trajectories.py:168-197, 302-332)
        double l_v = 0.0, l_acc = 0.0, l_s = 0.0, l_d = 0.0, l_thcl = 0.0, l_sd = 0.0, l_dd = 0.0;
        double l_x = 0.0, l_y = 0.0, l_cos = 1.0, l_sin = 0.0, cumx = 0.0, cumy = 0.0;   // (COLL: poses of the extended steps)
        // profile rows: wave-uniform row bases (the block of the wavefront's first pair) + one 32-bit byte offset per lane
        const size_t n8 = (size_t)n * 8;
        const char *const pb = reinterpret_cast<const char *>(a.profile) + (size_t)pair_w * PF_FIELDS * n8;
        const uint32_t voff = (uint32_t)(pair_slot - pair_w) * (uint32_t)PF_FIELDS * (uint32_t)n8;
        // iteration -1: the step in front of the block (or the last valid step, if that comes first) -- see the header
        const int i_pre = i_first - 1 < L - 1 ? i_first - 1 : L - 1;
#pragma nounroll
        for (int j = -1; j < RP_CHUNK_STEPS; ++j) {   // wave-uniform
            const bool pre = j < 0;
            const int i = pre ? i_pre : i_first + j;
            const bool live = pre ? (q > 0 && has_block) : i <= N;   // (q > 0: i_pre >= 0, since L >= 1)
            const bool act = live && i < L;
            const uint32_t ioff = voff + (uint32_t)(live ? i : 0) * 8u;
            auto fld = [&](int k) -> double { return *reinterpret_cast<const double *>(pb + (size_t)k * n8 + ioff); };
            double pose_x = 0.0, pose_y = 0.0, pose_cos = 1.0, pose_sin = 0.0;   // (COLL) rear-axle pose of this lane at step i
            if (__any(act)) {   // (wave-uniform) some lane's step is a valid one
                // -- the pair's profile at this step
                double s = fld(PF_S), sd = fld(PF_SD), sdd = fld(PF_SDD);
                const double inv_sd = fld(PF_INV_SD), th_ref = fld(PF_TH_REF);
                const double k_r = fld(PF_KR), k_r_d = fld(PF_KRD);
                const bool s_in_dom = fld(PF_INDOM) != 0.0;
                // -- lateral polynomial, reactive_planner.py:756-777
                const double t = (double)i * dt;
                const double tau = low ? s - s0 : t;
                double d = p_pos(tau), dd = p_vel(tau), ddd = p_acc(tau);
                if (fabs(dd) < RP_EPS) dd = 0.0;
                // -- d', d'' (:810-832)
                const bool moving = inv_sd > 0.0;
                double dp, dpp;
                if (!low) {
                    dp = dd * inv_sd;
                    const double ddot = ddd - dp * sdd;
                    dpp = ddot * inv_sd * inv_sd;
                } else {
                    dp = dd;
                    dpp = ddd;
                }
                // -- orientations (:842-873) incl. the standstill carry of :866
                const bool use_atan = moving || low;
                double th_cl = rp_atan(dp);
                double th_gl = th_cl + th_ref;
                const double w2 = __builtin_fma(dp, dp, 1.0);
                double cosT = rp_rsqrt(w2);
                double secT = w2 * cosT;
                double tanT = dp;
                if (!low && __any(act && !use_atan)) {   // standstill lanes keep the orientation of the step before (:866)
                    if (pre && act && !use_atan) {
                        // the step before is not this lane's: the orientation it carries is that of the last MOVING step in front of
                        // step i (x_0's when there is none) -- theta_cl + theta_ref there, with theta_cl = atan(d' / s')
                        int m = i - 1;
                        double inv_m = 0.0;
                        while (m >= 0) {
                            inv_m = *reinterpret_cast<const double *>(pb + (size_t)PF_INV_SD * n8 + voff + (uint32_t)m * 8u);
                            if (inv_m > 0.0) break;
                            --m;
                        }
                        th_prev = a.x0_orientation;
                        if (m >= 0) {
                            double ddm = p_vel((double)m * dt);
                            if (fabs(ddm) < RP_EPS) ddm = 0.0;
                            th_prev = rp_atan(ddm * inv_m) + *reinterpret_cast<const double *>(pb + (size_t)PF_TH_REF * n8 + voff + (uint32_t)m * 8u);
                        }
                    }
                    if (!use_atan) {
                        th_gl = th_prev;
                        th_cl = th_gl - th_ref;
                    }
                    double sn, cs;
                    rp_sincos(th_cl, &sn, &cs);
                    const double sc = rp_rcp(cs);
                    cosT = use_atan ? cosT : cs;
                    secT = use_atan ? secT : sc;
                    tanT = use_atan ? tanT : sn * sc;
                }
                // -- curvature, velocity, acceleration (:883-896)
                const double oneKrD = 1.0 - k_r * d;
                const double qk = cosT * rp_rcp(oneKrD);
                const double kterm = frenet_kterm(k_r_d, d, k_r, dp);
                double kappa = frenet_kappa(dpp, kterm, tanT, cosT, qk, k_r);
                const double f = oneKrD * secT;
                double v = sd * f;
                double acc = frenet_acc(sdd, f, sd, secT, oneKrD, tanT, kappa, k_r, kterm);
                const double dth = i > 0 ? th_gl - th_prev : 0.0;
                const double kdot = i > 0 ? kappa - ka_prev : 0.0;
                // -- _check_constraints, reactive_planner.py:971-1017 (order: velocity, kappa, yaw, kappa_dot, acc)
                uint32_t reason = RP_REASON_NONE;
                {
                    const double a_max = a.a_max, v_switch = a.v_switch, kappa_max = a.kappa_max;
                    const double wk = a.wheelbase * kappa;
                    const bool bad_yaw = fabs(rint(dth * a.c_yaw)) > kappa_max * v * 1e5;
                    const bool bad_kd = fabs(kdot) > a.c_kdot * __builtin_fma(wk, wk, 1.0);
                    const bool fast = v > v_switch;
                    const double acc_l = fast ? acc * v : acc, acc_r = fast ? a_max * v_switch : a_max;
                    const bool bad_acc = !((-a_max <= acc) & (acc_l <= acc_r));
                    const bool bad_v = v < -RP_EPS, bad_k = fabs(kappa) > kappa_max;
                    reason = (((cm & RP_CHECK_ACCELERATION) != 0) & bad_acc) ? RP_REASON_ACCELERATION : reason;
                    reason = (((cm & RP_CHECK_KAPPA_DOT) != 0) & bad_kd) ? RP_REASON_KAPPA_DOT : reason;
                    reason = (((cm & RP_CHECK_YAW_RATE) != 0) & bad_yaw) ? RP_REASON_YAW_RATE : reason;
                    reason = (((cm & RP_CHECK_KAPPA) != 0) & bad_k) ? RP_REASON_KAPPA : reason;
                    reason = (((cm & RP_CHECK_VELOCITY) != 0) & bad_v) ? RP_REASON_VELOCITY : reason;
                    reason = (act && !pre) ? reason : RP_REASON_NONE;   // (the step in front of the block is judged by the lane that owns it)
                }
                if (reason != RP_REASON_NONE && fail_step < 0) { fail_step = i; fail_reason = reason; alive = false; }
                // -- out of the projection domain (:908-917): no reason counter, the kinematic verdict of a later step still counts
                const bool in_dom = s_in_dom && fabs(d) <= a.proj_d_limit;
                if (act && !pre && !in_dom && ood_step < 0) ood_step = i;
                if (COLL) {   // (s, d) -> (x, y) = foot point + d * unit normal (:908-917); heading from cos / sin of theta_ref and theta_cl
                    const double sinT = tanT * cosT;
                    pose_x = fld(PF_PX) + d * fld(PF_NX);
                    pose_y = fld(PF_PY) + d * fld(PF_NY);
                    pose_cos = heading_cos(fld(PF_COS_REF), fld(PF_SIN_REF), cosT, sinT);
                    pose_sin = heading_sin(fld(PF_COS_REF), fld(PF_SIN_REF), cosT, sinT);
                }
                if (act) {
                    if (!pre) csum[j * RP_CHUNK_BLOCK] = cost_terms(i, acc, v, s, d, th_cl);
                    th_prev = th_gl;
                    ka_prev = kappa;
                }
                if (__any(act && i == L - 1)) {   // (wave-uniform) the last valid state: in its own block, and ahead of every block of extended states
                    const bool lastv = act && i == L - 1;
                    l_v = lastv ? v : l_v; l_acc = lastv ? acc : l_acc; l_s = lastv ? s : l_s; l_d = lastv ? d : l_d;
                    l_thcl = lastv ? th_cl : l_thcl; l_sd = lastv ? sd : l_sd; l_dd = lastv ? dd : l_dd;
                    if (COLL) { l_x = lastv ? pose_x : l_x; l_y = lastv ? pose_y : l_y; l_cos = lastv ? pose_cos : l_cos; l_sin = lastv ? pose_sin : l_sin; }
                }
                if (__ballot(alive) == 0) break;   // wave-uniform: every candidate has failed in this block
            }
            if (pre) {
                // (COLL) the extended steps in front of this lane's block, L .. i_first - 1: x[L:] = x[last] + cumsum(dt * v_tmp * cos(theta[last])) (:195-196)
                if (COLL && __any(live && i_first > L)) {
                    for (int k = 0; __any(live && L + k < i_first); ++k) {
                        if (live && L + k < i_first) {
                            const double tk = (double)(k + 1) * dt;
                            double vt = l_v + tk * l_acc;
                            vt = vt * (vt >= 0.0 ? 1.0 : 0.0);
                            cumx += dt * vt * l_cos;
                            cumy += dt * vt * l_sin;
                        }
                    }
                }
                continue;
            }
            if (live && !act) {   // i >= L: extended state (cost_function.py sums over the extended arrays)
                const double tk = (double)(i - L + 1) * dt;   // np.arange(1, steps + 1) * dt
                double vt = l_v + tk * l_acc;                 // :182
                vt = vt * (vt >= 0.0 ? 1.0 : 0.0);            // :184
                const double e_s = l_s + tk * l_sd;           // :330
                const double e_d = l_d + tk * l_dd;           // :331
                csum[j * RP_CHUNK_BLOCK] = cost_terms(i, l_acc, vt, e_s, e_d, l_thcl);
                if (COLL) {   // heading held (:188)
                    cumx += dt * vt * l_cos;
                    cumy += dt * vt * l_sin;
                    pose_x = l_x + cumx; pose_y = l_y + cumy; pose_cos = l_cos; pose_sin = l_sin;
                }
            }
            // -- eager collision query for this pose, reactive_planner.py:1033-1046
            if (COLL) {
                const bool want = live && alive && fail_step < 0 && ood_step < 0 && !collide;
                const double ego_cx = pose_x + a.wb_rear_axle * pose_cos, ego_cy = pose_y + a.wb_rear_axle * pose_sin;
                const uint64_t near_dyn = want ? double_as_mask(fld(PF_NEAR)) : 0;
                uint64_t near_sta = 0;
                if (COLL == 2 && want)
                    near_sta = static_grid_mask(a.obs.grid, a.obs.gx0, a.obs.gy0, a.obs.ginv, a.obs.gnx, a.obs.gny, ego_cx, ego_cy);
                const bool ask = want && (near_dyn | near_sta) != 0;
                if (__any(ask)) {   // wave-uniform; every lane runs the query code (wave-level culling inside)
                    const Obb ego = {ego_cx, ego_cy, pose_cos, pose_sin, a.half_length, a.half_width};
                    collide |= pose_collides<true, COLL == 2, false, true>(a.obs, ego, a.ego_radius, a.time_step0 + i * a.factor, ask, near_dyn, near_sta) && ask;
                }
            }
        }
    }
    // ---- the candidate's verdict over its G blocks: the first block that failed / left the domain, any pose that collides.  Blocks are
    //      in step order, so the smallest step is the first block's: minima of (step, reason) keys, through LDS; the workgroup's first
    //      wavefront does the bookkeeping of the 64 candidates.
    keys[(0 * NB + q) * RP_CHUNK_BLOCK + lane] = fail_step >= 0 ? (fail_step << 3) | (int)fail_reason : 0x7fffffff;
    keys[(1 * NB + q) * RP_CHUNK_BLOCK + lane] = ood_step >= 0 ? ood_step : 0x7fffffff;
    keys[(2 * NB + q) * RP_CHUNK_BLOCK + lane] = collide ? 1 : 0;
    __syncthreads();
    if (q != 0) return;   // (wave-uniform)
    {
        int fkey = 0x7fffffff, okey = 0x7fffffff, ckey = 0;
        for (int b = 0; b < NB; ++b) {
            const int f2 = keys[(0 * NB + b) * RP_CHUNK_BLOCK + lane], o2 = keys[(1 * NB + b) * RP_CHUNK_BLOCK + lane];
            fkey = f2 < fkey ? f2 : fkey;
            okey = o2 < okey ? o2 : okey;
            ckey |= keys[(2 * NB + b) * RP_CHUNK_BLOCK + lane];
        }
        fail_step = fkey == 0x7fffffff ? -1 : fkey >> 3;
        fail_reason = fkey == 0x7fffffff ? RP_REASON_NONE : (uint32_t)(fkey & 7);
        ood_step = okey == 0x7fffffff ? -1 : okey;
        collide = ckey != 0;
    }
    // ---- label, reason, cost
    uint32_t status;
    const bool decided_bad = pre_reason != RP_REASON_NONE || fail_step >= 0 || ood_step >= 0;
    if (pre_reason != RP_REASON_NONE) status = RP_LABEL_NONE | (pre_reason << 4);
    else if (fail_step >= 0) status = RP_LABEL_INFEASIBLE_KINEMATIC | (fail_reason << 4) | ((uint32_t)fail_step << 8);
    else if (ood_step >= 0) status = RP_LABEL_NONE | (RP_REASON_OUT_OF_DOMAIN << 4) | ((uint32_t)ood_step << 8);
    else status = collide ? RP_LABEL_INFEASIBLE_COLLISION : RP_LABEL_FEASIBLE;
    // sixteen partial sums by step mod 16, each over the step blocks (= the workgroup's wavefronts) in ascending order, then the tree of
    // group_sum_last<16> (lane 15's: ror 8, 4, 2, 1) -- rp_eval_kernel's order of additions
    double cost;
    {
        const lds_double gsum = (lds_double)(lds_chunk + lane);
        const int nblocks = NB;
        double pj[RP_CHUNK_STEPS];
#pragma unroll
        for (int k = 0; k < RP_CHUNK_STEPS; ++k) pj[k] = gsum[k * RP_CHUNK_BLOCK];   // (0.0 + the first block's term: the term itself)
        for (int b = 1; b < nblocks; ++b) {
#pragma unroll
            for (int k = 0; k < RP_CHUNK_STEPS; ++k) pj[k] += gsum[(b * RP_CHUNK_STEPS + k) * RP_CHUNK_BLOCK];
        }
        double s8[8], s4[4], s2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] = pj[8 + k] + pj[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) s4[k] = s8[4 + k] + s8[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) s2[k] = s4[2 + k] + s4[k];
        cost = s2[1] + s2[0];
    }
    if (decided_bad) cost = __builtin_nan("");
    const bool mine = valid;   // (the workgroup's first wavefront: one lane per candidate)
    if (mine) {
        a.status[slot] = status;
        a.cost[slot] = cost;
    }
    // ---- block partial: lexicographic (cost, index) min + counters of this wavefront
    if (a.partials) {
        const uint32_t lab = RP_STATUS_LABEL(status), rs = RP_STATUS_REASON(status);
        double bc = (mine && lab == RP_LABEL_FEASIBLE && cost == cost) ? cost : 0.0;
        long long bi = (mine && lab == RP_LABEL_FEASIBLE && cost == cost) ? (long long)gidx : -1;
        wave_min_pair(bc, bi);
        const Partials bp = partials_at(a.partials, a.partials_cap, a.partials_first + (int)blockIdx.x);
        const unsigned int n_feas = (unsigned int)__popcll(__ballot(mine && (lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION)));
        unsigned int cnt_mine = 0;
#pragma unroll
        for (uint32_t r = 1; r < 8; ++r) {
            const unsigned int cnt = (unsigned int)__popcll(__ballot(mine && rs == r));
            cnt_mine = lane == (int)(2 + r) ? cnt : cnt_mine;
        }
        cnt_mine = lane == 0 ? n_feas : cnt_mine;   // [0] n_feasible, [1] n_collision, [2 + r] reasons
        const unsigned int n_coll = (unsigned int)__popcll(__ballot(mine && lab == RP_LABEL_INFEASIBLE_COLLISION));
        cnt_mine = lane == 1 ? n_coll : cnt_mine;
        if (lane < RP_PARTIAL_CNT) bp.cnt[lane] = cnt_mine;
        if (lane == 0) { bp.cost[0] = bi >= 0 ? bc : 0.0; bp.idx[0] = bi; }
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU winner exchange on the device (replaces the multiprocessing.Queue fan-in of
// ReactivePlanner._get_optimal_trajectory, reactive_planner.py:1084-1111): every rank's result block -- FinalizeOut header
// + winner state rows, exactly what rp_finalize_kernel left in device memory -- has been all-gathered (RCCL) into `msgs`
// [world][msg_bytes]; one workgroup picks the global winner (lexicographic (cost, index)), sums the counters, and writes
// the combined block to the pinned host mirror, followed by the completion ticket.  pad_ of the output = owner rank.
// The count of colliding candidates before the GLOBAL winner is a second, rarely needed message (distributed.py).
// ------------------------------------------------------------------------------------------------
#define RP_COMBINE_THREADS 256
#define RP_COMBINE_MAX_WORLD 64
__global__ __launch_bounds__(RP_COMBINE_THREADS) void rp_combine_kernel(const char *msgs, int world, size_t msg_bytes, int n,
                                                                        FinalizeOut *host_out, unsigned long long seq) {
    __shared__ double sh_cost[RP_COMBINE_MAX_WORLD];
    __shared__ long long sh_idx[RP_COMBINE_MAX_WORLD];
    __shared__ long long sh_cnt[RP_COMBINE_MAX_WORLD][11];   // n_candidates, n_feasible, n_collision, reasons[8]
    __shared__ int sh_owner;
    __shared__ long long sh_sum[11];
    __shared__ unsigned int sh_flag[RP_COMBINE_MAX_WORLD];
    const int tid = threadIdx.x;
    if (tid < world) {
        const FinalizeOut *m = reinterpret_cast<const FinalizeOut *>(msgs + (size_t)tid * msg_bytes);
        sh_flag[tid] = m->w_status & RP_WSTATUS_ROWS_ON_HOST;
        sh_cost[tid] = m->r.best_cost;
        sh_idx[tid] = m->r.best_index;
        sh_cnt[tid][0] = m->r.n_candidates; sh_cnt[tid][1] = m->r.n_feasible; sh_cnt[tid][2] = m->r.n_collision;
        for (int k = 0; k < 8; ++k) sh_cnt[tid][3 + k] = m->r.reason_counts[k];
    }
    __syncthreads();
    if (tid < 11) {
        long long t = 0;
        for (int r = 0; r < world; ++r) t += sh_cnt[r][tid];
        sh_sum[tid] = t;
    }
    if (tid == 32) {   // (another wavefront than the sums)
        int owner = -1;
        for (int r = 0; r < world; ++r)
            if (sh_idx[r] >= 0 && (owner < 0 || better(sh_cost[r], (int64_t)sh_idx[r], sh_cost[owner], (int64_t)sh_idx[owner]))) owner = r;
        sh_owner = owner;
    }
    __syncthreads();
    const int owner = sh_owner;
    auto host_store = [](void *p, unsigned long long v) {
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    constexpr int words = (int)(offsetof(FinalizeOut, seq) / 8);
    if (tid < words) {
        const int k = tid;
        const unsigned long long nanb = 0x7ff8000000000000ull;
        unsigned long long v = nanb;
        if (k == 0) v = (unsigned long long)(owner >= 0 ? sh_idx[owner] : -1ll);
        else if (k == 1 || k == 30) v = owner >= 0 ? (unsigned long long)__double_as_longlong(sh_cost[owner]) : nanb;
        else if (k == 2) v = (unsigned long long)sh_sum[0];
        else if (k == 3) v = (unsigned long long)sh_sum[1];
        else if (k == 4 || k == 28) v = 0ull;                       // collisions before the global winner: second message
        else if (k == 5) v = (unsigned long long)sh_sum[2];
        else if (k >= 6 && k < 14) v = (unsigned long long)sh_sum[3 + k - 6];
        else if (k == 27) v = 0ull;
        else if (k == 29) v = (owner >= 0 ? (unsigned long long)(RP_LABEL_FEASIBLE | sh_flag[owner]) : 0ull) | ((unsigned long long)(unsigned int)owner << 32);
        host_store(reinterpret_cast<unsigned long long *>(host_out) + k, v);
    }
    if (owner >= 0) {
        const double *src = reinterpret_cast<const double *>(msgs + (size_t)owner * msg_bytes + sizeof(FinalizeOut));
        double *dst = reinterpret_cast<double *>(host_out + 1);
        for (int k = tid; k < RP_N_ARRAYS * n; k += RP_COMBINE_THREADS) host_store(dst + k, (unsigned long long)__double_as_longlong(src[k]));
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&host_out->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// rp_fetch_states of padded state rows: [rows][ns] -> compact [rows][n] in device memory (grid-stride, one element per
// thread and turn; the transfer to the host that follows is the slow part).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rp_compact_rows_kernel(const double *__restrict__ src, double *__restrict__ dst, int n, int ns, int tail,
                                                             uint32_t inv_n, unsigned long long total) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long blk_dev = (unsigned long long)state_block_doubles(ns, tail);
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        // one state block (14 n elements) at a time keeps the row index inside the exact range of the reciprocal
        const unsigned long long blk = k / (unsigned long long)(RP_N_ARRAYS * n);
        const uint32_t r = (uint32_t)(k - blk * (unsigned long long)(RP_N_ARRAYS * n));
        const uint32_t row = __umulhi(r, inv_n);
        dst[k] = src[blk * blk_dev + (unsigned long long)state_offset((int)row, (int)(r - row * (uint32_t)n), ns, tail)];
    }
}

// ------------------------------------------------------------------------------------------------
// Large batches leave one partial per workgroup (tens of thousands): fold them to gridDim.x partials
// before the one-workgroup selection epilogue.  One wavefront per output partial.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void rp_fold_partials_kernel(void *base, int cap, int n_in, int out_first) {
    const int lane = threadIdx.x;
    const Partials in = partials_at(base, cap), out = partials_at(base, cap, out_first);
    double bc = 0.0;
    long long bi = -1;
    double cnt[RP_PARTIAL_CNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // exact in double: counts are far below 2^53
    for (int k = blockIdx.x * 64 + lane; k < n_in; k += gridDim.x * 64) {
        const double pc = in.cost[k];
        const long long pi = in.idx[k];
        if (pi >= 0 && better(pc, (int64_t)pi, bc, (int64_t)bi)) { bc = pc; bi = pi; }
#pragma unroll
        for (int r = 0; r < RP_PARTIAL_CNT; ++r) cnt[r] += (double)in.cnt[(size_t)k * RP_PARTIAL_CNT + r];
    }
    wave_min_pair(bc, bi);
#pragma unroll
    for (int r = 0; r < RP_PARTIAL_CNT; ++r) cnt[r] = group_sum_last<64>(cnt[r]);
    if (lane == 63) {
        out.cost[blockIdx.x] = bc;
        out.idx[blockIdx.x] = bi;
        for (int r = 0; r < RP_PARTIAL_CNT; ++r) out.cnt[(size_t)blockIdx.x * RP_PARTIAL_CNT + r] = (uint32_t)cnt[r];
    }
}

// ------------------------------------------------------------------------------------------------
// Continuous collision check of one trajectory (reactive_planner.py:1049-1058): segment i = the tight rectangle
// around the ego rectangles of poses i and i + 1 (merge_swept), tested against the obstacles of scenario time index
// t0 + i -- the reference's ego object starts at x_0.time_step and advances one index per pose, planning.factor does
// not enter (:1050).  One lane per segment; first_hit receives the smallest colliding segment (atomicMin).
// ------------------------------------------------------------------------------------------------
#define RP_SWEPT_THREADS 256
// One workgroup; poses and results live in the pinned host mirror of the result block (device-visible): no copies, no memset --
// the kernel reads 3 n doubles over the bus, writes the first colliding segment (and the rectangles, if asked) back and hands
// the completion ticket to the spinning host thread.  Layout behind the FinalizeOut header: poses [3][n] | boxes [n-1][6].
__global__ __launch_bounds__(RP_SWEPT_THREADS) void rp_swept_kernel(ObsTables ob, int n, double wb_rear_axle, double hl, double hw, int t0,
                                                                    FinalizeOut *host_out, int want_boxes, unsigned long long seq) {
    __shared__ int sh_first;
    const int tid = threadIdx.x;
    const double *poses = reinterpret_cast<const double *>(host_out + 1);
    double *boxes = const_cast<double *>(poses) + 3 * (size_t)n;
    if (tid == 0) sh_first = 0x7fffffff;
    __syncthreads();
    for (int base = 0; base < n - 1; base += RP_SWEPT_THREADS) {   // (uniform trip count: pose_collides works wavefront-wide)
        const int i = base + tid;
        const bool want = i < n - 1;
        const int ic = want ? i : 0;
        double s0, c0, s1, c1;
        sincos(poses[2 * n + ic], &s0, &c0);
        sincos(poses[2 * n + ic + 1], &s1, &c1);
        const Obb a = {poses[ic] + wb_rear_axle * c0, poses[n + ic] + wb_rear_axle * s0, c0, s0, hl, hw};
        const Obb b = {poses[ic + 1] + wb_rear_axle * c1, poses[n + ic + 1] + wb_rear_axle * s1, c1, s1, hl, hw};
        const Obb m = merge_swept(a, b);
        if (want && want_boxes) {
            double *o = boxes + (size_t)i * 6;
            o[0] = m.cx; o[1] = m.cy; o[2] = m.ux; o[3] = m.uy; o[4] = m.hl; o[5] = m.hw;
        }
        // one radius for the wavefront (the rejection tests of pose_collides take a wave-uniform bound)
        const double r = (double)wave_max_f32(want ? (float)sqrt(m.hl * m.hl + m.hw * m.hw) : 0.f) * 1.000001;
        const bool hit = pose_collides<false, true>(ob, m, r, t0 + i, want, 0, 0);
        if (want && hit) atomicMin(&sh_first, i);
    }
    __threadfence_system();   // the rectangles, before the ticket
    __syncthreads();
    if (tid == 0) {
        host_out->n_before = (unsigned long long)(unsigned int)sh_first;
        __threadfence_system();
        __hip_atomic_store(&host_out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// Range of the costs of the kinematically feasible candidates (TrajectoryBundle.min_costs / max_costs after
// _get_optimal_trajectory, reactive_planner.py:650-651).  out[0] = min key, out[1] = max key, out[2] = count; keys are the
// order-preserving integer images of the doubles (sign flip), so that integer atomics order them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RP_BLOCK) void rp_cost_range_kernel(const uint32_t *status, const double *cost, int64_t count, unsigned long long *out) {
    __shared__ unsigned long long sh[3];
    if (threadIdx.x == 0) { sh[0] = ~0ull; sh[1] = 0ull; sh[2] = 0ull; }
    __syncthreads();
    unsigned long long lo = ~0ull, hi = 0ull, n = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * RP_BLOCK + threadIdx.x; i < count; i += (int64_t)gridDim.x * RP_BLOCK) {
        const uint32_t lab = RP_STATUS_LABEL(status[i]);
        const double c = cost[i];
        if ((lab == RP_LABEL_FEASIBLE || lab == RP_LABEL_INFEASIBLE_COLLISION) && c == c) {
            const unsigned long long k = cost_key(c);
            lo = k < lo ? k : lo;
            hi = k > hi ? k : hi;
            ++n;
        }
    }
    if (n) { atomicMin(&sh[0], lo); atomicMax(&sh[1], hi); atomicAdd(&sh[2], n); }
    __syncthreads();
    if (threadIdx.x == 0 && sh[2]) { atomicMin(&out[0], sh[0]); atomicMax(&out[1], sh[1]); atomicAdd(&out[2], sh[2]); }
}

// ------------------------------------------------------------------------------------------------
// Block partials from the status / cost arrays (plug-in cost path, rp_select).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RP_BLOCK) void rp_partials_kernel(const uint32_t *status, const double *cost, int64_t count,
                                                               int64_t cand_begin, void *partials, int partials_cap) {
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
        double best_cost = 0.0;
        int64_t best_index = -1;
        for (int k = 0; k < RP_BLOCK; ++k)
            if (sh_idx[k] >= 0 && better(sh_cost[k], sh_idx[k], best_cost, best_index)) {
                best_cost = sh_cost[k];
                best_index = sh_idx[k];
            }
        const Partials bp = partials_at(partials, partials_cap, (int)blockIdx.x);
        bp.cost[0] = best_cost;
        bp.idx[0] = best_index;
        for (int r = 0; r < RP_PARTIAL_CNT; ++r) bp.cnt[r] = (uint32_t)sh_cnt[r];
    }
}

// infeasible_count_collision: colliding feasible samples that precede the winner in cost order
// (the lazy loop of reactive_planner.py:1031-1046 touches exactly those).  out += count.
__global__ __launch_bounds__(RP_BLOCK) void rp_count_before_kernel(const uint32_t *status, const double *cost, int64_t count,
                                                                   int64_t cand_begin, const ResultCore *res,
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

