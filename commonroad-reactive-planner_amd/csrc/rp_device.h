// rp_device.h -- device-side building blocks of the fused trajectory kernel (gfx950 / CDNA4).
//
// Layout idea: one *group* of G lanes (G = 32 or 64, i.e. half or a whole
// wavefront) evaluates one candidate trajectory; lane l of the group owns time steps
// l, l+G, l+2G, ...  Neighbour-step dependencies of the reference loop
// (theta_gl[i-1], kappa_gl[i-1], the standstill orientation carry, "first failing step",
// the horizon extension and the cost sums) become cross-lane operations inside the group.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rp_amd.h"
#include "rp_math.h"

#define RP_EPS 1e-5  // _EPS, commonroad_rp/reactive_planner.py:49
#define RP_PI 3.14159265358979323846
#define RP_TWO_PI 6.28318530717958647692

// ------------------------------------------------------------------------------------------------
// group-level cross-lane helpers (all 64 lanes of the wave must execute them).
// A group is G = 16, 32 or 64 consecutive lanes = 1, 2 or 4 DPP rows of 16.  Neighbour moves, sums and
// prefix sums use DPP (plain VALU, no LDS round trip); only broadcasts from a run-time lane use
// ds_bpermute (__shfl).
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ uint64_t group_ballot(bool p, int gbase) {
    uint64_t b = __ballot(p);
    if (G == 64) return b;
    return (b >> gbase) & ((1ull << G) - 1ull);
}

template <int G>
__device__ __forceinline__ double group_bcast(double v, int src) { return __shfl(v, src, G); }

enum : int {
    DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118,
    DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128,
    DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143
};

// lanes whose source is outside the row / masked off receive 0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// value of the previous lane of the wave (lane 0: 0.0); callers patch the first lane of a group
__device__ __forceinline__ double lane_prev(double v) { return dpp_f64<DPP_WAVE_SHR1, 0xf>(v); }

// sum over the group, valid in the LAST lane of the group (gl == G - 1)
template <int G>
__device__ __forceinline__ double group_sum_last(double v) {
    v += dpp_f64<DPP_ROW_ROR8, 0xf>(v);
    v += dpp_f64<DPP_ROW_ROR4, 0xf>(v);
    v += dpp_f64<DPP_ROW_ROR2, 0xf>(v);
    v += dpp_f64<DPP_ROW_ROR1, 0xf>(v);          // every lane: sum of its row
    if (G >= 32) v += dpp_f64<DPP_ROW_BCAST15, 0xa>(v);   // rows 1, 3 += row 0, 2
    if (G == 64) v += dpp_f64<DPP_ROW_BCAST31, 0xc>(v);   // rows 2, 3 += lane 31 (rows 0 + 1)
    return v;
}

// inclusive prefix sum over the lanes of a group
template <int G>
__device__ __forceinline__ double group_scan(double v) {
    v += dpp_f64<DPP_ROW_SHR1, 0xf>(v);
    v += dpp_f64<DPP_ROW_SHR2, 0xf>(v);
    v += dpp_f64<DPP_ROW_SHR4, 0xf>(v);
    v += dpp_f64<DPP_ROW_SHR8, 0xf>(v);          // inclusive scan inside each row of 16
    if (G >= 32) v += dpp_f64<DPP_ROW_BCAST15, 0xa>(v);   // rows 1, 3 += total of row 0, 2
    if (G == 64) v += dpp_f64<DPP_ROW_BCAST31, 0xc>(v);   // rows 2, 3 += total of rows 0 + 1
    return v;
}

// ------------------------------------------------------------------------------------------------
// polynomials (PolynomialTrajectory.calc_*, commonroad_rp/polynomial_trajectory.py:240-271),
// evaluated in Horner form; ps = {c0..c5}, pv = {c1,2c2,3c3,4c4,5c5}, pa = {2c2,6c3,12c4,20c5}
// ------------------------------------------------------------------------------------------------
struct Poly {
    double c0, c1, c2, c3, c4, c5;
    __device__ __forceinline__ double pos(double t) const { return ((((c5 * t + c4) * t + c3) * t + c2) * t + c1) * t + c0; }
    __device__ __forceinline__ double vel(double t) const {
        return (((5.0 * c5 * t + 4.0 * c4) * t + 3.0 * c3) * t + 2.0 * c2) * t + c1;
    }
    __device__ __forceinline__ double acc(double t) const { return ((20.0 * c5 * t + 12.0 * c4) * t + 6.0 * c3) * t + 2.0 * c2; }
};

// QuinticTrajectory._calc_coeffs_static (polynomial_trajectory.py:292-320): closed-form solution of
// the 3x3 system instead of LAPACK gesv.
__device__ __forceinline__ Poly quintic_coeffs(double p0, double v0, double a0, double pf, double vf, double af, double T) {
    // (the fused multiply-adds are spelled out wherever a sum of two products could be contracted either way round: every kernel
    //  that inlines this function then computes the same bits -- see frenet_kappa in rp_kernels.h)
    const double T2 = T * T;
    const double bp = pf - __builtin_fma(0.5 * a0, T2, __builtin_fma(v0, T, p0));
    const double bv = vf - __builtin_fma(a0, T, v0);
    const double ba = af - a0;
    // c3 = (20 bp - 8 T bv + T^2 ba) / (2 T^3), c4 = (-30 bp + 14 T bv - 2 T^2 ba) / (2 T^4),
    // c5 = (12 bp - 6 T bv + T^2 ba) / (2 T^5): one Newton-refined reciprocal of T, then products
    const double iT = rp_rcp(T);
    const double iT2 = iT * iT, iT3 = iT2 * iT;
    const double Tbv = T * bv, T2ba = T2 * ba;
    Poly c;
    c.c0 = p0; c.c1 = v0; c.c2 = 0.5 * a0;
    c.c3 = (__builtin_fma(20.0, bp, -(8.0 * Tbv)) + T2ba) * (0.5 * iT3);
    c.c4 = __builtin_fma(-2.0, T2ba, __builtin_fma(-30.0, bp, 14.0 * Tbv)) * (0.5 * iT3 * iT);
    c.c5 = (__builtin_fma(12.0, bp, -(6.0 * Tbv)) + T2ba) * (0.5 * iT3 * iT2);
    return c;
}

// QuarticTrajectory._calc_coeffs_static_ (polynomial_trajectory.py:341-360), closed form of the 2x2.
__device__ __forceinline__ Poly quartic_coeffs(double p0, double v0, double a0, double T, double vd) {
    const double bv = __builtin_fma(-a0, T, vd - v0);
    const double ba = -a0;
    const double iT = rp_rcp(T);
    const double iT2 = iT * iT;
    Poly c;
    c.c0 = p0; c.c1 = v0; c.c2 = 0.5 * a0;
    c.c3 = __builtin_fma(3.0, bv, -(T * ba)) * (iT2 * (1.0 / 3.0));
    c.c4 = __builtin_fma(T, ba, -(2.0 * bv)) * (0.25 * iT2 * iT);
    c.c5 = 0.0;
    return c;
}

// ------------------------------------------------------------------------------------------------
// reference-path tables.  Nine rows of n doubles, contiguous: pos, theta, curv, curv_d, x, y, tx, ty
// (utils_coordinate_system.py:114-118 + the polyline and its vertex tangents) and invlen[k] =
// 1 / (pos[k+1] - pos[k]) (last entry: the wrap-around pair 1 / (pos[0] - pos[n-1])), followed by
// an int32 bucket table for the O(1) segment lookup.
// ------------------------------------------------------------------------------------------------
enum { TB_POS = 0, TB_THETA, TB_CURV, TB_CURV_D, TB_X, TB_Y, TB_TX, TB_TY, TB_INVLEN, TB_ROWS };

// make_valid_orientation (commonroad-io, used by interpolate_angle, utils_coordinate_system.py:43)
__device__ __forceinline__ double make_valid_orientation(double a) {
    double m = a;
    if (!(fabs(a) < RP_TWO_PI)) m = fmod(a, RP_TWO_PI);  // exact either way; fmod only off the fast path
    if (m < 0.0) m += RP_TWO_PI;
    if (RP_PI <= m && m <= RP_TWO_PI) m -= RP_TWO_PI;
    return m;
}

// first index with pos[idx] > s (n if none): np.argmax(ref_pos > s) (reactive_planner.py:835) is
// this index, or 0 when it is n.  Fixed trip count, no divergence.
__device__ __forceinline__ int upper_bound(const double *pos, int n, int iters, double s) {
    int lo = 0, hi = n;
    for (int it = 0; it < iters; ++it) {
        int mid = (lo + hi) >> 1;
        mid = mid < n ? mid : n - 1;
        bool right = (lo < hi) && !(pos[mid] > s);
        bool left = (lo < hi) && (pos[mid] > s);
        lo = right ? mid + 1 : lo;
        hi = left ? mid : hi;
    }
    return lo;
}

// Same result through a uniform bucket table: bucket[b] = first index with pos > pos[0] + b*h,
// h <= the shortest segment, so at most one vertex lies inside a bucket; the two forward steps and
// one backward step absorb the rounding of the bucket index.
__device__ __forceinline__ int upper_bound_bucket(const double *pos, const int *bucket, int n, int nb, double inv_h,
                                                  double pos_first, double pos_last, double s) {
    int ub = (s < pos_first) ? 0 : n;          // NaN -> n, like np.argmax of an all-False mask minus the -1 later
    if (s >= pos_first && s < pos_last) {
        int b = (int)((s - pos_first) * inv_h);
        b = b < nb ? b : nb - 1;
        ub = bucket[b];
        ub = (ub > 0 && pos[ub - 1] > s) ? ub - 1 : ub;
        ub = (ub < n && !(pos[ub] > s)) ? ub + 1 : ub;
        ub = (ub < n && !(pos[ub] > s)) ? ub + 1 : ub;
    }
    return ub;
}

// ------------------------------------------------------------------------------------------------
// collision primitives: ego OBB (RectOBB of reactive_planner.py:1041) against the obstacle tables.
// Closed sets, separating-axis tests; same formulas as oracle/rp_oracle.c.
// ------------------------------------------------------------------------------------------------
struct Obb { double cx, cy, ux, uy, hl, hw; };

// The four separating axes of two rectangles, in two halves so that a caller can test the axes of b first (for
// the thin strips of a road boundary nearly every pose that is not on the strip is separated along the strip's
// normal) and run the other half only for the lanes that are left.  obb_obb is the conjunction, same formulas.
__device__ __forceinline__ bool obb_sep_on_b_axes(const Obb &a, const Obb &b) {
    const double avx = -a.uy, avy = a.ux, bvx = -b.uy, bvy = b.ux;
    const double tx = b.cx - a.cx, ty = b.cy - a.cy;
    const double uu = a.ux * b.ux + a.uy * b.uy, uv = a.ux * bvx + a.uy * bvy;
    const double vu = avx * b.ux + avy * b.uy, vv = avx * bvx + avy * bvy;
    bool sep = fabs(tx * b.ux + ty * b.uy) > b.hl + (a.hl * fabs(uu) + a.hw * fabs(vu));
    sep |= fabs(tx * bvx + ty * bvy) > b.hw + (a.hl * fabs(uv) + a.hw * fabs(vv));
    return sep;
}
__device__ __forceinline__ bool obb_sep_on_a_axes(const Obb &a, const Obb &b) {
    const double avx = -a.uy, avy = a.ux, bvx = -b.uy, bvy = b.ux;
    const double tx = b.cx - a.cx, ty = b.cy - a.cy;
    const double uu = a.ux * b.ux + a.uy * b.uy, uv = a.ux * bvx + a.uy * bvy;
    const double vu = avx * b.ux + avy * b.uy, vv = avx * bvx + avy * bvy;
    bool sep = fabs(tx * a.ux + ty * a.uy) > a.hl + (b.hl * fabs(uu) + b.hw * fabs(uv));
    sep |= fabs(tx * avx + ty * avy) > a.hw + (b.hl * fabs(vu) + b.hw * fabs(vv));
    return sep;
}
__device__ __forceinline__ bool obb_obb(const Obb &a, const Obb &b) { return !(obb_sep_on_a_axes(a, b) || obb_sep_on_b_axes(a, b)); }

// cheap conservative rejection for long thin rectangles (road-boundary strips), where a bounding circle says
// nothing: the ego rectangle lies inside the circle (ego centre, ego_r), so it cannot reach b if that centre is
// farther than ego_r from b's slab along either of b's axes (these are two of the four separating axes with the
// ego's extent over-estimated)
__device__ __forceinline__ bool obb_slab_far(const Obb &b, double ex, double ey, double ego_r) {
    const double tx = ex - b.cx, ty = ey - b.cy;
    return fabs(tx * b.ux + ty * b.uy) > (b.hl + ego_r) * 1.000001 || fabs(ty * b.ux - tx * b.uy) > (b.hw + ego_r) * 1.000001;
}

// bits of the static clusters within reach of an ego rectangle centred at (ex, ey)
__device__ __forceinline__ uint64_t static_grid_mask(const unsigned long long *grid, double gx0, double gy0, double ginv, int gnx, int gny,
                                                     double ex, double ey) {
    const double fx = (ex - gx0) * ginv, fy = (ey - gy0) * ginv;
    if (gnx < 0 || !(fx == fx && fy == fy)) return ~0ull;   // no grid (a shape without finite extent) / NaN pose: nothing can be
                                                             // ruled out, the exact tests decide
    // outside the grid nothing is within reach: it covers every shape inflated by the reach
    if (!(fx >= 0.0 && fy >= 0.0 && fx < (double)gnx && fy < (double)gny)) return 0ull;
    return grid[(size_t)(int)fy * (size_t)gnx + (size_t)(int)fx];
}

// the slab along b's normal alone (the one that says something for a long thin strip)
__device__ __forceinline__ bool obb_normal_far(const Obb &b, double ex, double ey, double ego_r) {
    const double tx = ex - b.cx, ty = ey - b.cy;
    return fabs(ty * b.ux - tx * b.uy) > (b.hw + ego_r) * 1.000001;
}

__device__ __forceinline__ bool obb_tri(const Obb &a, const double *t) {
    double vx = -a.uy, vy = a.ux;
    double lx[3], ly[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double px = t[2 * k] - a.cx, py = t[2 * k + 1] - a.cy;
        lx[k] = px * a.ux + py * a.uy;
        ly[k] = px * vx + py * vy;
    }
    bool sep = fmin(lx[0], fmin(lx[1], lx[2])) > a.hl || fmax(lx[0], fmax(lx[1], lx[2])) < -a.hl;
    sep |= fmin(ly[0], fmin(ly[1], ly[2])) > a.hw || fmax(ly[0], fmax(ly[1], ly[2])) < -a.hw;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int k2 = (k + 1) % 3;
        double ex = lx[k2] - lx[k], ey = ly[k2] - ly[k];
        double nx = -ey, ny = ex;
        double p0 = lx[0] * nx + ly[0] * ny, p1 = lx[1] * nx + ly[1] * ny, p2 = lx[2] * nx + ly[2] * ny;
        double r = a.hl * fabs(nx) + a.hw * fabs(ny);
        sep |= fmin(p0, fmin(p1, p2)) > r || fmax(p0, fmax(p1, p2)) < -r;
    }
    return !sep;
}

__device__ __forceinline__ bool obb_circ(const Obb &a, double cx, double cy, double r) {
    double vx = -a.uy, vy = a.ux;
    double px = cx - a.cx, py = cy - a.cy;
    double lx = px * a.ux + py * a.uy, ly = px * vx + py * vy;
    double dx = fmax(fabs(lx) - a.hl, 0.0), dy = fmax(fabs(ly) - a.hw, 0.0);
    return dx * dx + dy * dy <= r * r;
}

// Tight rectangle around two rectangles: the occupancy between two consecutive poses for the continuous collision
// check (reactive_planner.py:1049-1052, commonroad-dc's trajectory_preprocess_obb_sum -- source absent; this is the
// "OBB sum" of the FCL library commonroad-dc builds on, OBB::operator+, restated in the plane).  Orientation: the
// bisector of the two headings (a rectangle is the same set with its axis flipped, so the second heading is flipped
// when the two point apart), or the line through the centres when they are farther apart than twice the sum of the
// larger half extents; extents: both rectangles projected on the two axes.  Same formulas as oracle/rp_oracle.c.
__device__ __forceinline__ Obb merge_swept(const Obb &a, const Obb &b) {
    const double dx = b.cx - a.cx, dy = b.cy - a.cy;
    const double lim = 2.0 * (fmax(a.hl, a.hw) + fmax(b.hl, b.hw));
    double nx, ny;
    if (dx * dx + dy * dy > lim * lim) { nx = dx; ny = dy; }
    else {
        const double s = (a.ux * b.ux + a.uy * b.uy) < 0.0 ? -1.0 : 1.0;
        nx = a.ux + s * b.ux; ny = a.uy + s * b.uy;
    }
    const double nrm = sqrt(nx * nx + ny * ny);
    nx = nx / nrm; ny = ny / nrm;
    const double mx = -ny, my = nx;
    // projections relative to a's centre: centre +- (hl |u.n| + hw |v.n|)
    const double ea0 = a.hl * fabs(a.ux * nx + a.uy * ny) + a.hw * fabs(a.uy * nx - a.ux * ny);
    const double ea1 = a.hl * fabs(a.ux * mx + a.uy * my) + a.hw * fabs(a.uy * mx - a.ux * my);
    const double eb0 = b.hl * fabs(b.ux * nx + b.uy * ny) + b.hw * fabs(b.uy * nx - b.ux * ny);
    const double eb1 = b.hl * fabs(b.ux * mx + b.uy * my) + b.hw * fabs(b.uy * mx - b.ux * my);
    const double pb0 = dx * nx + dy * ny, pb1 = dx * mx + dy * my;
    const double lo0 = fmin(-ea0, pb0 - eb0), hi0 = fmax(ea0, pb0 + eb0);
    const double lo1 = fmin(-ea1, pb1 - eb1), hi1 = fmax(ea1, pb1 + eb1);
    const double c0 = 0.5 * (lo0 + hi0), c1 = 0.5 * (lo1 + hi1);
    Obb m;
    m.cx = a.cx + (c0 * nx + c1 * mx);
    m.cy = a.cy + (c0 * ny + c1 * my);
    m.ux = nx; m.uy = ny;
    m.hl = 0.5 * (hi0 - lo0);
    m.hw = 0.5 * (hi1 - lo1);
    return m;
}

// Obstacle tables on the device.  Static shapes: array-of-structs rows of 8 doubles, every lane
// reads the same row (broadcast).  Dynamic OBBs: struct-of-arrays [7][n_dyn][n_steps] so that the
// lanes of a group (consecutive time steps) read consecutive addresses.
enum { OB_CX = 0, OB_CY, OB_UX, OB_UY, OB_HL, OB_HW, OB_R, OB_PAD, OB_ROW };  // static obb row
#define RP_SLOT_ROW 8
struct ObsTables {
    const double *sobb;     // [n_sobb][8]  cx, cy, ux, uy, hl, hw, r_bound, -
    const double *tri;      // [n_tri][10]  x1,y1,x2,y2,x3,y3, bx, by, r_bound, -
    const double *circ;     // [n_circ][4]  cx, cy, r, -
    const double *dyn;      // [7][n_dyn][n_steps]  cx, cy, ux, uy, hl, hw, r_bound (cx = NaN: absent); + dyn_xy_offset, dyn_rmax_offset
    // static shapes in table order, grouped into at most 63 clusters of consecutive shapes of one kind (boundary
    // rectangles come as polylines, so consecutive shapes are neighbours): bounding circle + member range
    const double *clus;        // [n_clus][4]  cx, cy, r, -
    const int32_t *clus_info;  // [n_clus][4]  kind (0 obb, 1 triangle, 2 circle), first, count, -
    // one row per member slot of the clusters, in cluster order (slot = cluster * clus_per + member): what the broad phase of
    // the single-launch prologue tests, as one flat table it can stage in LDS instead of chasing cluster descriptor -> shape row
    // through dependent loads.  Rectangles as they are; triangles and circles as the square around their bounding circle;
    // slots without a member: half extents -inf (never near).
    const double *slot;        // [n_clus * clus_per][RP_SLOT_ROW]  cx, cy, ux, uy, hl, hw, bit of the cluster (as a double), kind + 4 index
    // uniform grid over the static shapes (rp_host.hip: ensure_static_grid): cell -> bits of the clusters with a member within
    // reach of an ego rectangle whose centre lies in the cell.  Every pose looks its own mask up (static_grid_mask), so the walk
    // of the narrow phase visits the few clusters next to the pose instead of every cluster near any candidate of the pair.
    const unsigned long long *grid;   // [gny][gnx]
    double gx0, gy0, ginv;            // origin and 1 / cell size
    int32_t gnx, gny;
    int32_t n_sobb, n_tri, n_circ, n_dyn, n_steps, dyn_t0, n_clus, clus_per;   // clus_per: members per cluster (upper bound)
    double dyn_rmax_all;              // largest bounding radius of any dynamic obstacle at any step (pose_collides: circle pre-test of a lane's mask)
};
// behind the seven planes of `dyn`: [n_dyn][n_steps][2] cx, cy interleaved (16-byte aligned), then [n_dyn] the largest r_bound of
// each obstacle over the steps where it exists
__host__ __device__ inline size_t dyn_xy_offset(int n_dyn, int n_steps) { return ((size_t)7 * (size_t)n_dyn * (size_t)n_steps + 1) & ~(size_t)1; }
__host__ __device__ inline size_t dyn_rmax_offset(int n_dyn, int n_steps) { return dyn_xy_offset(n_dyn, n_steps) + (size_t)2 * (size_t)n_dyn * (size_t)n_steps; }
__host__ __device__ inline size_t dyn_table_doubles(int n_dyn, int n_steps) { return dyn_rmax_offset(n_dyn, n_steps) + (size_t)n_dyn; }
typedef const int32_t __attribute__((address_space(4))) *gcint;

typedef const double __attribute__((address_space(4))) *gcdouble;   // the tables are read-only for the kernels: constant address
                                                                    // space, so that wave-uniform rows come through scalar loads

// ---- wave-level culling --------------------------------------------------------------------------
// All lanes of a wavefront hold poses that are close in space (a few candidates x consecutive time steps).
// One circle around the ego rectangles of the lanes that want a query is tested against each obstacle's
// bounding circle with wave-uniform arithmetic on scalar-loaded table rows; only obstacles that survive
// reach the per-lane work (row loads, per-lane circle test, separating-axis test).  Conservative by
// construction, so the exact results are unchanged.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {   // lanes without a valid source keep their own value
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_f32<DPP_ROW_SHR1, 0xf>(v));
    v = fminf(v, dpp_f32<DPP_ROW_SHR2, 0xf>(v));
    v = fminf(v, dpp_f32<DPP_ROW_SHR4, 0xf>(v));
    v = fminf(v, dpp_f32<DPP_ROW_SHR8, 0xf>(v));       // lane 15 of every row: minimum of the row
    v = fminf(v, dpp_f32<DPP_ROW_BCAST15, 0xa>(v));    // rows 1, 3 also see row 0, 2
    v = fminf(v, dpp_f32<DPP_ROW_BCAST31, 0xc>(v));    // rows 2, 3 also see rows 0 + 1
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) { return -wave_min_f32(-v); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {   // lanes without a valid source keep their own value
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= dpp_u32<DPP_ROW_SHR1, 0xf>(v);
    v |= dpp_u32<DPP_ROW_SHR2, 0xf>(v);
    v |= dpp_u32<DPP_ROW_SHR4, 0xf>(v);
    v |= dpp_u32<DPP_ROW_SHR8, 0xf>(v);
    v |= dpp_u32<DPP_ROW_BCAST15, 0xa>(v);
    v |= dpp_u32<DPP_ROW_BCAST31, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {   // OR over the 64 lanes, wave-uniform result
    return ((uint64_t)wave_or_u32((uint32_t)(v >> 32)) << 32) | wave_or_u32((uint32_t)v);
}

struct WaveBound { double mx, my, R; };   // every ego rectangle of a querying lane lies within R of (mx, my)

__device__ __forceinline__ WaveBound ego_wave_bound(bool want, double cx, double cy, double ego_r) {
    const float inf = __builtin_huge_valf();
    const float fx = (float)cx, fy = (float)cy;
    const double x0 = wave_min_f32(want ? fx : inf), x1 = wave_max_f32(want ? fx : -inf);
    const double y0 = wave_min_f32(want ? fy : inf), y1 = wave_max_f32(want ? fy : -inf);
    WaveBound b;
    b.mx = 0.5 * (x0 + x1);
    b.my = 0.5 * (y0 + y1);
    const double hx = 0.5 * (x1 - x0), hy = 0.5 * (y1 - y0);
    // |c - (float)c| <= 2^-24 |c| for the lane values; L1 norm of the half extents >= their L2 norm (no square root)
    b.R = hx + hy + ego_r + (fabs(b.mx) + fabs(b.my) + hx + hy) * 2.4e-7;
    return b;
}

__device__ __forceinline__ bool coarse_near(const WaveBound &wb, double cx, double cy, double r) {
    const double dx = cx - wb.mx, dy = cy - wb.my, rr = r + wb.R;
    return dx * dx + dy * dy <= rr * rr * 1.000001;   // false for NaN (absent)
}

// cc.collide(ego pose at scenario time index t)  (reactive_planner.py:1040-1042).
// STATIC: the obstacle tables hold static shapes (else that half of the query is not compiled in).
// `want`: this lane asks; MASKED (a property of the kernel variant): `near` / `near_static` hold this lane's (pair, step) masks of dynamic
// obstacles / static-shape clusters (rp_kernels.h: near_mask_step); otherwise every shape is tested (static shapes
// after a wave-level bounding-circle rejection).
// Per lane, a bounding-circle rejection comes before the exact test (conservative: a small relative margin keeps
// it from ever rejecting a pair the exact test would accept).
// LDS_SLOTS (single-launch variants): `lds_slot` is the workgroup's LDS copy of ObsTables::slot; the walk over the static
// clusters reads member rows from there (broadcast reads, tens of cycles) instead of through chained scalar loads from device
// memory (cluster descriptor -> member rows: a few hundred cycles per cluster with nothing to overlap them with when a SIMD
// holds one or two wavefronts -- that chain, not the tests, was the cost of a road boundary on a small batch).
#ifndef RP_KEEP_ADDRESS_MATH_IN_LOOP
#define RP_KEEP_ADDRESS_MATH_IN_LOOP 1   // pose_collides: see the dynamic part
#endif
#ifndef RP_WALK_PRETEST
#define RP_WALK_PRETEST 1        // batched circle pre-test of a lane's mask of dynamic obstacles (pose_collides)
#endif
#ifndef RP_WALK_PRETEST_BATCH
#define RP_WALK_PRETEST_BATCH 4  // ... centres requested at a time (eight: 32 registers, the state-row variants spill 14 more -- cfg2 15.7 -> 17.6 us)
#endif
#ifndef RP_WALK_PRETEST_MIN
#define RP_WALK_PRETEST_MIN 2    // ... for wavefronts with a lane whose mask holds more bits than this
#endif
#ifdef RP_WALK_COUNT
static __device__ unsigned long long *rp_walk_dbg = nullptr;   // diagnostic: clusters walked / queries / exact tests of one wavefront
#define RP_WSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0); if (rp_walk_dbg && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) rp_walk_dbg[(k)] = t_; } while (0)
#else
#define RP_WSTAMP(k) do { } while (0)
#endif
// STATIC_MASKED: `near_static` holds this lane's bits of static clusters (the evaluation kernels: the pose's cell of the grid over
// the static shapes, whatever the plan; MASKED only says whether `near` is a mask of DYNAMIC obstacles); else every static shape
// is tested behind a wave-level bounding-circle rejection (the swept-volume check, whose boxes the grid was not built for).
template <bool MASKED, bool STATIC, bool LDS_SLOTS = false, bool STATIC_MASKED = MASKED, bool SIZES_IN_LOOP = true>
__device__ __forceinline__ bool pose_collides(const ObsTables &ob, const Obb &ego, double ego_r, int t, bool want, uint64_t near,
                                              uint64_t near_static, const double *lds_slot = nullptr) {
    bool hit = false;
    const gcdouble sobb = (gcdouble)ob.sobb, tri = (gcdouble)ob.tri, circ = (gcdouble)ob.circ, dyn = (gcdouble)ob.dyn;
    if (STATIC_MASKED && STATIC && LDS_SLOTS) {
        uint64_t mu = wave_or_u64(want ? near_static : 0);
        mu &= ob.n_clus >= 64 ? ~0ull : (1ull << ob.n_clus) - 1ull;
        const int per = ob.clus_per;
#ifdef RP_WALK_COUNT
        if (rp_walk_dbg && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) { rp_walk_dbg[0] += (unsigned long long)__popcll(mu); rp_walk_dbg[1] += 1; }
#endif
        RP_WSTAMP(8);
        auto exact = [&](const Obb &b, int tag, bool close) -> bool {   // tag: kind + 4 * index in the table of that kind (wave-uniform)
            const int kind = tag & 3, j = tag >> 2;
            if (kind == 0) return close && !obb_sep_on_b_axes(ego, b) && !obb_sep_on_a_axes(ego, b);   // strip's own axes first
            if (kind == 1) {
                const gcdouble o = tri + (size_t)j * 10;
                const double dx = o[6] - ego.cx, dy = o[7] - ego.cy, rr = ego_r + o[8];
                if (!(close && dx * dx + dy * dy <= rr * rr * 1.000001)) return false;
                const double tv[6] = {o[0], o[1], o[2], o[3], o[4], o[5]};
                return obb_tri(ego, tv);
            }
            const gcdouble o = circ + (size_t)j * 4;
            return close && obb_circ(ego, o[0], o[1], o[2]);
        };
        while (mu != 0) {   // wave-uniform
            const int c = __ffsll((unsigned long long)mu) - 1;
            mu &= mu - 1;
            // (explicitly an LDS pointer: through a generic one the rows come by flat loads, which count on the vector-memory
            //  counter as well -- and waiting for them then means waiting for the write-through row stores issued before)
            typedef const double __attribute__((address_space(3))) *lds_cdouble;
            const lds_cdouble row = (lds_cdouble)lds_slot + (size_t)(c * per) * RP_SLOT_ROW;
            for (int m = 0; m < per; m += 2) {
                const bool two = m + 1 < per;
                const lds_cdouble o0 = row + (size_t)m * RP_SLOT_ROW, o1 = row + (size_t)(two ? m + 1 : m) * RP_SLOT_ROW;
                const Obb b0 = {o0[0], o0[1], o0[2], o0[3], o0[4], o0[5]};
                const Obb b1 = {o1[0], o1[1], o1[2], o1[3], o1[4], o1[5]};
                const double g0 = o0[7], g1 = o1[7];
                // the shape's two slabs (slots without a member have half extents -inf: always far) with the ego's bounding circle
                // first; the exact test only when some lane of the wavefront comes that close
                const bool c0 = want && !obb_slab_far(b0, ego.cx, ego.cy, ego_r);
                const bool c1 = want && two && !obb_slab_far(b1, ego.cx, ego.cy, ego_r);
                if (__any(c0 || c1)) {
#ifdef RP_WALK_COUNT
                    { const int ne_ = (__any(c0) ? 1 : 0) + (__any(c1) ? 1 : 0); if (rp_walk_dbg && blockIdx.x == RP_STAMP_BLOCK && threadIdx.x == RP_STAMP_THREAD) rp_walk_dbg[2] += ne_; }
#endif
                    hit |= exact(b0, __builtin_amdgcn_readfirstlane((int)g0), c0);
                    hit |= exact(b1, __builtin_amdgcn_readfirstlane((int)g1), c1);
                }
            }
        }
        RP_WSTAMP(9);
    } else if (STATIC_MASKED && STATIC) {
        // static shapes: the clusters whose bit is set in the (pair, step) mask of ANY lane of the wavefront, walked
        // with wave-uniform control flow -- rows come through scalar loads (one 64-byte row per instruction, no
        // per-lane gathers, no dependent vector-memory round trips), every lane runs the cheap rejection test
        uint64_t mu = wave_or_u64(want ? near_static : 0);
        mu &= ob.n_clus >= 64 ? ~0ull : (1ull << ob.n_clus) - 1ull;
        const gcint info = (gcint)ob.clus_info;
        // The walk is a chain of scalar loads (cluster descriptor -> member rows -> test); the descriptor of the NEXT
        // cluster is requested before the rows of the current one are waited for, and two member rows are in flight
        // at a time, so that the round trips overlap instead of adding up.
        int kind = 0, first = 0, count = 0;
        if (mu != 0) {
            const int c0 = __ffsll((unsigned long long)mu) - 1;
            mu &= mu - 1;
            kind = info[4 * c0]; first = info[4 * c0 + 1]; count = info[4 * c0 + 2];
        }
        while (count > 0) {
            const bool more = mu != 0;
            const int cn = more ? __ffsll((unsigned long long)mu) - 1 : 0;
            if (more) mu &= mu - 1;
            const int nkind = info[4 * cn], nfirst = info[4 * cn + 1], ncount = more ? info[4 * cn + 2] : 0;   // next descriptor
            if (kind == 0) {
                for (int q = 0; q < count; q += 2) {
                    const gcdouble o0 = sobb + (size_t)(first + q) * OB_ROW;
                    const gcdouble o1 = sobb + (size_t)(first + (q + 1 < count ? q + 1 : q)) * OB_ROW;
                    const Obb b0 = {o0[OB_CX], o0[OB_CY], o0[OB_UX], o0[OB_UY], o0[OB_HL], o0[OB_HW]};
                    const Obb b1 = {o1[OB_CX], o1[OB_CY], o1[OB_UX], o1[OB_UY], o1[OB_HL], o1[OB_HW]};
                    // Most members are strips of a road boundary a few metres to the side: one slab (the strip's normal) with the
                    // ego's bounding circle rejects them in ~7 instructions; the exact test runs only when some lane of the
                    // wavefront comes that close (wave-uniform), the strip's own axes first.
                    const bool c0 = want && !obb_normal_far(b0, ego.cx, ego.cy, ego_r);
                    const bool c1 = want && q + 1 < count && !obb_normal_far(b1, ego.cx, ego.cy, ego_r);
                    if (__any(c0 || c1)) {
                        if (c0 && !obb_sep_on_b_axes(ego, b0)) hit |= !obb_sep_on_a_axes(ego, b0);
                        if (c1 && !obb_sep_on_b_axes(ego, b1)) hit |= !obb_sep_on_a_axes(ego, b1);
                    }
                }
            } else if (kind == 1) {
                for (int q = 0; q < count; ++q) {
                    const gcdouble o = tri + (size_t)(first + q) * 10;
                    const double dx = o[6] - ego.cx, dy = o[7] - ego.cy, rr = ego_r + o[8];
                    if (want && dx * dx + dy * dy <= rr * rr * 1.000001) {
                        const double tv[6] = {o[0], o[1], o[2], o[3], o[4], o[5]};
                        hit |= obb_tri(ego, tv);
                    }
                }
            } else {
                for (int q = 0; q < count; ++q) {
                    const gcdouble o = circ + (size_t)(first + q) * 4;
                    if (want) hit |= obb_circ(ego, o[0], o[1], o[2]);
                }
            }
            kind = nkind; first = nfirst; count = ncount;
        }
    } else if (STATIC) {
    WaveBound wb = {0.0, 0.0, 0.0};
    if (ob.n_sobb + ob.n_tri + ob.n_circ > 0) wb = ego_wave_bound(want, ego.cx, ego.cy, ego_r);   // wave-uniform
    for (int j = 0; j < ob.n_sobb; ++j) {
        const gcdouble o = sobb + j * OB_ROW;
        const double cx = o[OB_CX], cy = o[OB_CY], r = o[OB_R];
        if (!coarse_near(wb, cx, cy, r)) continue;   // wave-uniform
        const Obb b = {cx, cy, o[OB_UX], o[OB_UY], o[OB_HL], o[OB_HW]};
        if (want && !obb_sep_on_b_axes(ego, b)) hit |= !obb_sep_on_a_axes(ego, b);
    }
    for (int j = 0; j < ob.n_tri; ++j) {
        const gcdouble o = tri + j * 10;
        const double bx = o[6], by = o[7], r = o[8];
        if (!coarse_near(wb, bx, by, r)) continue;   // wave-uniform
        const double dx = bx - ego.cx, dy = by - ego.cy, rr = ego_r + r;
        if (want && dx * dx + dy * dy <= rr * rr * 1.000001) {
            const double tv[6] = {o[0], o[1], o[2], o[3], o[4], o[5]};
            hit |= obb_tri(ego, tv);
        }
    }
    for (int j = 0; j < ob.n_circ; ++j) {
        const gcdouble o = circ + j * 4;
        const double cx = o[0], cy = o[1], r = o[2];
        if (!coarse_near(wb, cx, cy, r)) continue;   // wave-uniform
        if (want) hit |= obb_circ(ego, cx, cy, r);
    }
    }
    // dynamic obstacles: only those whose bit is set in the (pair, step) mask of the longitudinal profile
    // (The address arithmetic of this part -- seven plane offsets, the bases of the centre and radius tables, the mask of valid bits --
    //  does not change over a kernel's step loop; hoisted out of it, it takes some thirty scalar registers the evaluation kernels do
    //  not have: they are spilled into lanes of a vector register and every later use is a v_readlane -- a VECTOR instruction, 84 of
    //  them in the 16-lane kernel's loop.  The two table sizes go through an empty asm here, so that what is derived from them stays
    //  where it is used, as scalar multiplies and adds: 118 -> 79 v_readlane in the costs-only 16-lane kernel, 143 -> 61 in the lane
    //  kernel with the query, cfg3 production kernel -1.9 % (same box); the variants that store state rows went the other way
    //  (46 -> 54) and keep the hoisted form: SIZES_IN_LOOP.)
    int nd_u = ob.n_dyn, ns_u = ob.n_steps;
    if (RP_KEEP_ADDRESS_MATH_IN_LOOP && SIZES_IN_LOOP) asm volatile("" : "+s"(nd_u), "+s"(ns_u));
    const int k = t - ob.dyn_t0;
    const bool k_ok = want && k >= 0 && k < ns_u;
    const int kc = k_ok ? k : 0;
    const size_t plane = (size_t)nd_u * (size_t)ns_u;
    uint64_t m = k_ok ? near : 0;
    if (!MASKED) {
        // no (pair, step) mask came with the pose (explicit-polynomial plans, the swept check): it is built here, from the table
        // the profile kernel builds its masks from -- one 16-byte centre load per obstacle and a scalar radius (the obstacle's
        // largest over its steps: a superset of the exact circle test below), eight obstacles requested before the first is
        // tested -- and the walk below then visits the few obstacles that are near.  Round 2 walked every obstacle here with
        // seven loads each, waited for one obstacle at a time (357 vector loads per wavefront and step block at 51 obstacles).
        m = 0;
        if (__any(k_ok)) {
            typedef double dbl2 __attribute__((ext_vector_type(2)));
            typedef const dbl2 __attribute__((address_space(4))) *gcdouble2;
            const gcdouble2 xy = (gcdouble2)(dyn + dyn_xy_offset(nd_u, ns_u));
            const gcdouble rmax = dyn + dyn_rmax_offset(nd_u, ns_u);
            constexpr int kB = 8;
            for (int j0 = 0; j0 < nd_u; j0 += kB) {
                dbl2 oc[kB];
#pragma unroll
                for (int u = 0; u < kB; ++u) {   // (a batch's tail repeats the last obstacle: same circle, same bit)
                    const int j = j0 + u < nd_u ? j0 + u : nd_u - 1;
                    oc[u] = xy[(size_t)j * ns_u + kc];
                }
#pragma unroll
                for (int u = 0; u < kB; ++u) {
                    const int j = j0 + u < nd_u ? j0 + u : nd_u - 1;
                    const double dx = oc[u].x - ego.cx, dy = oc[u].y - ego.cy, rr = ego_r + rmax[j];   // NaN centre: absent, no bit
                    m |= (uint64_t)(dx * dx + dy * dy <= rr * rr * 1.000001) << (j < 63 ? j : 63);
                }
            }
        }
        if (!k_ok) m = 0;
    }
    RP_WSTAMP(10);
    const bool overflow = (m >> 63) != 0 && nd_u > 63;
    m &= nd_u >= 63 ? ~(1ull << 63) : (1ull << nd_u) - 1ull;
    if (MASKED && RP_WALK_PRETEST && __any(__popcll((unsigned long long)m) > RP_WALK_PRETEST_MIN)) {
        // The (pair, step) masks are exact to the lateral range of the pair's candidates only (~7 m): in traffic a lane's mask holds
        // several obstacles, few of which its own pose comes near, and the walk below costs one memory round trip per bit of the
        // fullest mask of the wavefront.  Here the bits are thinned first, ALL AT ONCE: the centres of up to eight obstacles of the
        // mask requested together (one 16-byte load each, the table the masks were built from), tested against the ego's bounding
        // circle with the largest obstacle radius of the scene -- a superset of the walk's own, exact circle test, so the walk
        // visits the same obstacles it would have tested, and nothing else changes.  (cfg3: 4 - 8 bits per lane, 0 - 2 after this.)
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        typedef const dbl2 __attribute__((address_space(4))) *gcdouble2;
        const gcdouble2 xy = (gcdouble2)(dyn + dyn_xy_offset(nd_u, ns_u));
        const double rr1 = ego_r + ob.dyn_rmax_all;
        const double lim = rr1 * rr1 * 1.000001;
        constexpr int kB = RP_WALK_PRETEST_BATCH;
        uint64_t rest = m, keep = 0;
        while (__any(rest != 0)) {
            dbl2 oc[kB];
            uint64_t r1 = rest;
#pragma unroll
            for (int u = 0; u < kB; ++u) {   // (a lane with fewer bits repeats obstacle 0: tested, never kept)
                const int j = r1 != 0 ? __ffsll((unsigned long long)r1) - 1 : 0;
                r1 &= r1 - 1;
                oc[u] = xy[(size_t)j * ns_u + kc];
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                const uint64_t bit = rest & (0 - rest);   // lowest set bit (0 when none is left)
                rest &= rest - 1;
                const double dx = oc[u].x - ego.cx, dy = oc[u].y - ego.cy;
                if (dx * dx + dy * dy <= lim) keep |= bit;   // false for NaN (absent)
            }
        }
        m = keep;
    }
    while (__any(m != 0)) {   // wave-uniform trip count = largest number of near obstacles among the lanes
        const bool act = m != 0;
        const int j = act ? __ffsll((unsigned long long)m) - 1 : 0;
        m &= m - 1;           // (0 stays 0)
        const gcdouble o = dyn + (size_t)j * ns_u + kc;
        // all seven loads are independent and issue back to back; NaN centre = obstacle absent
        const double cx = o[0], cy = o[plane], ux = o[2 * plane], uy = o[3 * plane], hl = o[4 * plane], hw = o[5 * plane],
                     rr = ego_r + o[6 * plane];
        const double dx = cx - ego.cx, dy = cy - ego.cy;
        if (act && dx * dx + dy * dy <= rr * rr * 1.000001) {   // false for NaN
            Obb b = {cx, cy, ux, uy, hl, hw};
            hit |= obb_obb(ego, b);
        }
    }
    if (__any(overflow)) {    // more than 63 dynamic obstacles: the tail is tested one by one
        for (int j = 63; j < nd_u; ++j) {
            const gcdouble o = dyn + (size_t)j * ns_u + kc;
            const double cx = o[0], cy = o[plane], rr = ego_r + o[6 * plane];
            const double dx = cx - ego.cx, dy = cy - ego.cy;
            if (overflow && dx * dx + dy * dy <= rr * rr * 1.000001) {
                Obb b = {cx, cy, o[2 * plane], o[3 * plane], o[4 * plane], o[5 * plane]};
                hit |= obb_obb(ego, b);
            }
        }
    }
    return hit;
}
