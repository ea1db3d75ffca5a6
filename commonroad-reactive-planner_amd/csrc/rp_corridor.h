// rp_corridor.h -- host-side batch view of the adaptive sampling space: the candidate list of
// CorridorSampling.generate_trajectories_at_level (commonroad_rp/sampling.py:345-397) as coefficient arrays, without one Python
// object per candidate, per longitudinal sample or per corridor look-up.  Set-up work like rp_frontend.h: no GPU involved.
//
// The reference iterates over Python sets of floats -- set(np.linspace(low, up, n)), and for lateral intervals that straddle the
// reference path set(np.linspace(lo, hi, n)).union({0}) -- so the ORDER of the candidates is the iteration order of CPython's
// set: PySetF64 restates that container (Objects/setobject.c of CPython 3.8 .. 3.12: open addressing, 9 linear probes, then
// i = 5 i + 1 + perturb; growth x4 once fill * 5 >= mask * 3; float hash = value reduced modulo 2^61 - 1, Python/pyhash.c).
// The binding checks this restatement against the interpreter's own sets when it loads and keeps the Python batch view if they
// ever disagree (commonroad_rp_amd/sampling.py).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rpco {

// hash(float): _Py_HashDouble for finite values (Python/pyhash.c); inf / nan do not occur in sample sets
inline int64_t py_hash_double(double v) {
    constexpr int kBits = 61;
    constexpr uint64_t kMod = (1ull << kBits) - 1ull;
    if (!(std::fabs(v) <= 1.7976931348623157e308)) return v > 0 ? 314159 : (v < 0 ? -314159 : 0);
    int e;
    double m = std::frexp(v, &e);
    int sign = 1;
    if (m < 0) { sign = -1; m = -m; }
    uint64_t x = 0;
    while (m != 0.0) {
        x = ((x << 28) & kMod) | (x >> (kBits - 28));
        m *= 268435456.0;   // 2^28
        e -= 28;
        const uint64_t y = (uint64_t)m;
        m -= (double)y;
        x += y;
        if (x >= kMod) x -= kMod;
    }
    e = e >= 0 ? e % kBits : kBits - 1 - ((-1 - e) % kBits);
    x = ((x << e) & kMod) | (x >> (kBits - e));
    int64_t h = (int64_t)x * sign;
    if (h == -1) h = -2;
    return h;
}

// A CPython set of floats: insertion, copy (set_merge into an empty set) and iteration in table order.
class PySetF64 {
  public:
    PySetF64() : table_(8), mask_(7) {}
    void add(double key) { add_entry(key, py_hash_double(key)); }
    // set(iterable): one set_add_key per element
    static PySetF64 from_values(const double *v, int n) {
        PySetF64 s;
        for (int i = 0; i < n; ++i) s.add(v[i]);
        return s;
    }
    // s.union(other): set_copy(s) -- a new set merged from s (set_merge into an empty table) -- then other's keys one by one
    PySetF64 union_with(const double *other, int n_other) const {
        PySetF64 r;
        r.merge_from(*this);
        // set_update_internal(result, other) with other a set: set_merge again (resize check, then normal insertions)
        if ((r.fill_ + (size_t)n_other) * 5 >= r.mask_ * 3) r.resize((r.used_ + (size_t)n_other) * 2);
        for (int i = 0; i < n_other; ++i) r.add(other[i]);
        return r;
    }
    void values(std::vector<double> &out) const {
        for (const Entry &e : table_)
            if (e.used) out.push_back(e.key);
    }
    size_t size() const { return used_; }

  private:
    struct Entry { double key = 0.0; int64_t hash = 0; bool used = false; };
    static constexpr size_t kLinearProbes = 9;
    std::vector<Entry> table_;
    size_t mask_, fill_ = 0, used_ = 0;

    static void insert_clean(std::vector<Entry> &t, size_t mask, double key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask;
        for (;;) {
            if (!t[i].used) { t[i] = {key, hash, true}; return; }
            if (i + kLinearProbes <= mask)
                for (size_t j = 1; j <= kLinearProbes; ++j)
                    if (!t[i + j].used) { t[i + j] = {key, hash, true}; return; }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(size_t minused) {
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<Entry> nt(newsize);
        for (const Entry &e : table_)
            if (e.used) insert_clean(nt, newsize - 1, e.key, e.hash);
        table_.swap(nt);
        mask_ = newsize - 1;
        fill_ = used_;
    }
    void add_entry(double key, int64_t hash) {
        size_t perturb = (size_t)hash, i = (size_t)hash & mask_;
        for (;;) {
            const size_t probes = (i + kLinearProbes <= mask_) ? kLinearProbes : 0;
            for (size_t j = 0; j <= probes; ++j) {
                Entry &e = table_[i + j];
                if (!e.used) {
                    e = {key, hash, true};
                    ++fill_; ++used_;
                    if (fill_ * 5 >= mask_ * 3) resize(used_ > 50000 ? used_ * 2 : used_ * 4);
                    return;
                }
                if (e.hash == hash && e.key == key) return;   // already there (0.0 == -0.0, equal hashes)
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask_;
        }
    }
    // set_merge(this = empty, other)
    void merge_from(const PySetF64 &o) {
        if (o.used_ == 0) return;
        if ((fill_ + o.used_) * 5 >= mask_ * 3) resize((used_ + o.used_) * 2);
        if (fill_ == 0 && mask_ == o.mask_ && o.fill_ == o.used_) {   // same size, no dummies: the table is copied slot by slot
            table_ = o.table_;
            fill_ = o.fill_; used_ = o.used_;
            return;
        }
        fill_ = used_ = o.used_;
        for (const Entry &e : o.table_)
            if (e.used) insert_clean(table_, mask_, e.key, e.hash);
    }
};

// np.linspace(lo, hi, n): arange(n) * step + lo with step = (hi - lo) / (n - 1), last element = hi
inline void linspace(double lo, double hi, int n, std::vector<double> &out) {
    out.resize((size_t)n);
    if (n == 1) { out[0] = lo; return; }
    const double delta = hi - lo, div = (double)(n - 1), step = delta / div;
    if (step == 0.0 && delta != 0.0) {   // (NumPy: underflow of the step -> multiply first)
        for (int i = 0; i < n; ++i) out[i] = ((double)i / div) * delta + lo;
    } else {
        for (int i = 0; i < n; ++i) out[i] = (double)i * step + lo;
    }
    out[n - 1] = hi;
}

struct Box { double p_lon_min, p_lon_max, p_lat_min, p_lat_max, v_lon_min, v_lon_max; };

// Candidates of one sampling level.  Per time sample k: T[k], traj_len[k], velocity interval [v_low[k], v_up[k]], corridor nodes
// boxes[box_off[k] .. box_off[k + 1]).  Output: one row per candidate, in the reference's order.
struct Candidates {
    std::vector<double> lon, lat, T, v_end, d_end;   // [C][6], [C][6], [C], [C], [C]
    std::vector<int32_t> traj_len;
    // candidates that share a longitudinal polynomial -- the lateral samples of one (T, v) sample -- are adjacent: group[C] numbers
    // them (0, 1, ... in order of appearance), first[groups] is a group's first candidate
    std::vector<int32_t> group, first;
};

// candidates of the time samples [k_first, k_last), appended to `out` in the reference's order
inline void corridor_candidates(int k_first, int k_last, const double *T, const int32_t *traj_len, const double *v_low, const double *v_up,
                                const int32_t *box_off, const Box *boxes, int n, const double *x0_lon, const double *x0_lat,
                                Candidates &out) {
    std::vector<double> lin, vs, ds, lateral;
    std::vector<int> ids, comp;
    {   // room for the usual case up front (n velocity samples x (n + 1) lateral samples per connected part): the parts of a level are
        // worked out on several threads at once, and growing six vectors step by step is what they would contend for (the allocator)
        size_t nodes_max = 0;
        for (int k = k_first; k < k_last; ++k) nodes_max = std::max(nodes_max, (size_t)(box_off[k + 1] - box_off[k]));
        const size_t guess = out.T.size() + (size_t)(k_last - k_first) * (size_t)n * (size_t)(n + 1) * std::max<size_t>(nodes_max, 1);
        out.lon.reserve(6 * guess); out.lat.reserve(6 * guess);
        out.T.reserve(guess); out.v_end.reserve(guess); out.d_end.reserve(guess); out.traj_len.reserve(guess); out.group.reserve(guess);
    }
    const double s0 = x0_lon[0], sv0 = x0_lon[1], sa0 = x0_lon[2];
    const double p0 = x0_lat[0], v0 = x0_lat[1], a0 = x0_lat[2];
    const double zero = 0.0;
    for (int k = k_first; k < k_last; ++k) {
        const double t = T[k];
        linspace(v_low[k], v_up[k], n, lin);
        vs.clear();
        PySetF64::from_values(lin.data(), n).values(vs);                                   // sampling.py:367
        const Box *nodes = boxes + box_off[k];
        const int n_nodes = box_off[k + 1] - box_off[k];
        const double t2 = std::pow(t, 2.0), t3 = std::pow(t, 3.0), t4 = std::pow(t, 4.0), t5 = std::pow(t, 5.0);
        for (double v : vs) {
            // QuarticTrajectory (polynomial_trajectory.py:341-360): [[3 T^2, 4 T^3], [6 T, 12 T^2]] x = [v - v0 - a0 T, -a0], closed form
            const double bv = v - sv0 - sa0 * t, ba = -sa0;
            double c[6] = {s0, sv0, sa0 / 2.0, (3.0 * bv - t * ba) / (3.0 * t * t), (t * ba - 2.0 * bv) / (4.0 * t * t * t), 0.0};
            const double end = c[0] + c[1] * t + c[2] * t2 + c[3] * t3 + c[4] * t4 + c[5] * t5;   // :369
            ids.clear();
            for (int j = 0; j < n_nodes; ++j)
                if (nodes[j].p_lon_min <= end && end <= nodes[j].p_lon_max) ids.push_back(j);       // :374-375
            if (ids.empty()) continue;
            // connected parts: nodes whose lateral intervals overlap or touch, transitively; parts in the order of their first members
            const int m = (int)ids.size();
            comp.resize((size_t)m);
            for (int a = 0; a < m; ++a) comp[a] = a;
            auto find = [&](int a) { while (comp[a] != a) { comp[a] = comp[comp[a]]; a = comp[a]; } return a; };
            for (int a = 0; a < m; ++a)
                for (int b = a + 1; b < m; ++b)
                    if (nodes[ids[a]].p_lat_min <= nodes[ids[b]].p_lat_max && nodes[ids[b]].p_lat_min <= nodes[ids[a]].p_lat_max) {
                        const int ra = find(a), rb = find(b);
                        if (ra != rb) comp[ra > rb ? ra : rb] = ra < rb ? ra : rb;
                    }
            lateral.clear();
            for (int root = 0; root < m; ++root) {
                if (find(root) != root) continue;   // (roots in ascending order = order of first members)
                double lo = 0.0, hi = 0.0;
                bool first = true;
                for (int a = 0; a < m; ++a)
                    if (find(a) == root) {
                        const Box &nd = nodes[ids[a]];
                        lo = first ? nd.p_lat_min : (nd.p_lat_min < lo ? nd.p_lat_min : lo);
                        hi = first ? nd.p_lat_max : (nd.p_lat_max > hi ? nd.p_lat_max : hi);
                        first = false;
                    }
                linspace(lo, hi, n, lin);
                const PySetF64 base = PySetF64::from_values(lin.data(), n);
                if (lo < 0.0 && 0.0 < hi) base.union_with(&zero, 1).values(lateral);                 // :384-386
                else base.values(lateral);
            }
            if (!lateral.empty()) out.first.push_back((int32_t)out.T.size());
            const int32_t gid = (int32_t)out.first.size() - 1;
            for (double d : lateral) {
                // QuinticTrajectory to (d, 0, 0) over t (polynomial_trajectory.py:292-320), closed form
                const double T2 = t * t, T3 = T2 * t;
                const double bp = d - (p0 + v0 * t + 0.5 * a0 * T2), bvl = -(v0 + a0 * t), bal = -a0;
                const double q[6] = {p0, v0, 0.5 * a0, (20.0 * bp - 8.0 * t * bvl + T2 * bal) / (2.0 * T3),
                                     (-30.0 * bp + 14.0 * t * bvl - 2.0 * T2 * bal) / (2.0 * T3 * t),
                                     (12.0 * bp - 6.0 * t * bvl + T2 * bal) / (2.0 * T3 * T2)};
                out.lon.insert(out.lon.end(), c, c + 6);
                out.lat.insert(out.lat.end(), q, q + 6);
                out.T.push_back(t); out.v_end.push_back(v); out.d_end.push_back(d);
                out.traj_len.push_back(traj_len[k]);
                out.group.push_back(gid);
            }
        }
    }
}

}  // namespace rpco
