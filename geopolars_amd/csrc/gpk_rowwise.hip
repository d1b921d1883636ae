// gpk_rowwise.hip — 1-to-1 row-wise binary operators.
//   distance               geoseries.rs:141-146,248-251 (intended impl ops::distance::euclidean_distance,
//                          geoseries.rs:250) — geo 0.27 euclidean_distance.rs + geo-types private_utils.rs
//   contains / within /    north-star additions to the trait; semantics = geo's Contains / Intersects
//   intersects             as dispatched in spatial_index.rs:89-137
//
// Mapping: G lanes (power of two, picked from the mean vertex count of the non-point side) share one
// row.  Lane k takes segments k, k+G, ... of each ring, so a wave reads 64 consecutive vertices per
// load instruction (coalesced 16-byte loads) no matter how ragged the rows are; the per-row minimum,
// winding number and any-hit flags are folded with xor-shuffles inside the group.  That is the
// "bin on vertex count so lanes in a wave see similar work" rule of the north star applied per call.
#include <cfloat>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gpk_device.h"
#include "gpk_polypoly.h"
#include "gpk_contains.h"
#include "gpk_lineal.h"
#include "gpk_scan.h"

namespace gpk {

// ---- per-segment pieces -----------------------------------------------------------------------------
// geo-types private_utils::line_segment_distance, evaluated as a SQUARED distance kept as a fraction
// num / den, so that the per-segment work has no division and no hypot (both cost tens of f64
// instructions on the vector unit and made this kernel VALU-bound):
//     degenerate segment or r <= 0     -> |p - s|^2 / 1
//     r >= 1                           -> |p - e|^2 / 1
//     otherwise                        -> cross^2 / |e - s|^2          (upstream: |cross / d2| * hypot(dx, dy))
// r = dot / d2 is compared with 0 and 1 through dot <= 0 and dot >= d2 (same sign; at r ~ 1 the two
// formulas agree to O((1-r)^2)).  Fractions are compared by cross-multiplication; one divide + sqrt per
// row at the end.  Results agree with the upstream expression to a few ulps, inside the 1e-9 contract.
struct Frac {
    double num, den;
};
__device__ __forceinline__ bool frac_less(const Frac& a, const Frac& b) { return a.num * b.den < b.num * a.den; }
__device__ __forceinline__ double frac_sqrt(const Frac& f) { return f.num == INFINITY ? DBL_MAX : sqrt(f.num / f.den); }

__device__ __forceinline__ Frac segment_dist2(double px, double py, double sx, double sy, double ex, double ey, double& cross_out,
                                              double& dxdy_out) {
    const double dx = ex - sx, dy = ey - sy, qx = px - sx, qy = py - sy;
    const double d2 = dx * dx + dy * dy;
    const double dot = qx * dx + qy * dy;
    const double cross = qx * dy - qy * dx;  // == -((sy - py) * dx - (sx - px) * dy)
    cross_out = cross;
    dxdy_out = dx * dy;
    if (d2 == 0.0 || dot <= 0.0) return Frac{qx * qx + qy * qy, 1.0};
    if (dot >= d2) {
        const double rx = px - ex, ry = py - ey;
        return Frac{rx * rx + ry * ry, 1.0};
    }
    return Frac{cross * cross, d2};
}

// geo-types private_utils::line_string_contains_point, one segment (tolerance f64::EPSILON on |tx - ty|).
// tx - ty == cross / (dx * dy) up to ~3 ulps of O(1) quantities, so the two divisions are only needed when
// |cross| <= 8 eps |dx dy|; everywhere else the upstream predicate is certainly false.  Inside that band the
// upstream expression is evaluated verbatim, so the zero / non-zero outcome of `distance` is exact.
__device__ __forceinline__ bool segment_contains_eps(double px, double py, double sx, double sy, double ex, double ey, double cross,
                                                     double dxdy) {
    const double dx = ex - sx, dy = ey - sy;
    if (dx == 0.0 && dy == 0.0) return px == sx && py == sy;
    if (dy == 0.0) {
        if (py != sy) return false;
        const double t = (px - sx) / dx;
        return 0.0 <= t && t <= 1.0;
    }
    if (dx == 0.0) {
        if (px != sx) return false;
        const double t = (py - sy) / dy;
        return 0.0 <= t && t <= 1.0;
    }
    if (fabs(cross) > 1.7763568394002505e-15 * fabs(dxdy)) return false;  // 8 * 2^-52
    const double tx = (px - sx) / dx, ty = (py - sy) / dy;
    return fabs(tx - ty) <= DBL_EPSILON && 0.0 <= tx && tx <= 1.0;
}

template <int G>
__device__ __forceinline__ double gmin(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
template <int G>
__device__ __forceinline__ Frac gmin_frac(Frac v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const Frac w{__shfl_xor(v.num, o, 64), __shfl_xor(v.den, o, 64)};
        if (frac_less(w, v)) v = w;
    }
    return v;
}
template <int G>
__device__ __forceinline__ int gsum(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int G>
__device__ __forceinline__ int gor(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// One coordinate sequence against one point, G lanes cooperating.
struct SeqAcc {
    Frac dmin;      // min line_segment_distance, squared, as a fraction
    int wn;         // winding number (rings)
    int on_ring;    // coordinate_position boundary hit
    int eps_hit;    // line_string_contains_point (vertex equality or eps-collinear)
};
template <int G, bool WANT_DIST, bool WANT_POS>
__device__ __forceinline__ SeqAcc scan_sequence(const double2* __restrict__ xy, int c0, int c1, double px, double py,
                                                int lane) {
    SeqAcc a{Frac{INFINITY, 1.0}, 0, 0, 0};
    const int n = c1 - c0;
    if (n == 1) {
        const double2 p = xy[c0];
        const int eq = p.x == px && p.y == py;
        a.on_ring = eq;
        a.eps_hit = eq;
    }
    for (int i = c0 + lane; i + 1 < c1; i += G) {
        const double2 s = xy[i], e = xy[i + 1];
        if (WANT_POS) {
            int wn = 0;
            a.on_ring |= (int)dev::ring_edge(s.x, s.y, e.x, e.y, px, py, wn);
            a.wn += wn;
        }
        if (WANT_DIST) {
            double cross, dxdy;
            const Frac d = segment_dist2(px, py, s.x, s.y, e.x, e.y, cross, dxdy);
            if (frac_less(d, a.dmin)) a.dmin = d;
            a.eps_hit |= (int)((s.x == px && s.y == py) || (e.x == px && e.y == py) ||
                               segment_contains_eps(px, py, s.x, s.y, e.x, e.y, cross, dxdy));
        }
    }
    if (WANT_DIST) {
        a.dmin = gmin_frac<G>(a.dmin);
        a.eps_hit = gor<G>(a.eps_hit);
    }
    if (WANT_POS) {
        a.wn = gsum<G>(a.wn);
        a.on_ring = gor<G>(a.on_ring);
    }
    return a;
}
__device__ __forceinline__ int pos_of(const SeqAcc& a, int n) {
    if (n == 0) return dev::POS_OUTSIDE;
    if (a.on_ring) return dev::POS_BOUNDARY;
    return a.wn == 0 ? dev::POS_OUTSIDE : dev::POS_INSIDE;
}

// point_line_string_euclidean_distance
template <int G>
__device__ __forceinline__ double point_linestring_distance(const double2* xy, int c0, int c1, double px, double py,
                                                            int lane) {
    if (c1 == c0) return 0.0;
    const SeqAcc a = scan_sequence<G, true, false>(xy, c0, c1, px, py, lane);
    return a.eps_hit ? 0.0 : frac_sqrt(a.dmin);
}

// EuclideanDistance<Point, Polygon>: 0 if the polygon intersects the point (or its exterior is empty);
// else min over holes (as linestrings) and exterior segments.  Also returns the polygon position.
template <int G, bool WANT_DIST>
__device__ __forceinline__ double point_polygon(const DevGeo& b, int r0, int r1, double px, double py, int lane,
                                                int* pos_out) {
    *pos_out = dev::POS_OUTSIDE;
    if (r1 <= r0) return 0.0;
    const int e0 = b.ring_off[r0], e1 = b.ring_off[r0 + 1];
    if (e1 == e0) return 0.0;
    const SeqAcc ext = scan_sequence<G, WANT_DIST, true>(b.xy, e0, e1, px, py, lane);
    int pos = pos_of(ext, e1 - e0);
    double dh = DBL_MAX;
    bool resolved = pos != dev::POS_INSIDE;  // Outside / Boundary: holes do not change the position
    for (int r = r0 + 1; r < r1; ++r) {
        const int h0 = b.ring_off[r], h1 = b.ring_off[r + 1];
        if (!WANT_DIST && resolved) break;
        const SeqAcc h = scan_sequence<G, WANT_DIST, true>(b.xy, h0, h1, px, py, lane);
        if (!resolved) {
            const int ph = pos_of(h, h1 - h0);
            if (ph == dev::POS_BOUNDARY) {
                pos = dev::POS_BOUNDARY;
                resolved = true;
            } else if (ph == dev::POS_INSIDE) {
                pos = dev::POS_OUTSIDE;
                resolved = true;
            }
        }
        if (WANT_DIST) {
            const double d = (h1 == h0 || h.eps_hit) ? 0.0 : frac_sqrt(h.dmin);
            dh = d < dh ? d : dh;
        }
    }
    *pos_out = pos;
    if (!WANT_DIST) return 0.0;
    if (pos != dev::POS_OUTSIDE) return 0.0;
    const double de = frac_sqrt(ext.dmin);
    return dh < de ? dh : de;
}

// distance from one point to row j of b
// KIND selects the right-side geometry family at compile time (one instantiation per family keeps the hot
// kernel free of the other families' code and registers); KIND < 0 = decide at run time.
constexpr int KIND_ANY = -1;
template <int G, int KIND = KIND_ANY>
__device__ __forceinline__ double point_geom_distance(const DevGeo& b, int64_t j, double px, double py, int lane) {
    switch (KIND == KIND_ANY ? b.type : KIND) {
    case GPK_GEOM_POINT: {
        const double2 q = b.xy[j];
        return hypot(px - q.x, py - q.y);
    }
    case GPK_GEOM_MULTIPOINT: {
        double m = DBL_MAX;
        for (int i = b.geom_off[j] + lane; i < b.geom_off[j + 1]; i += G) {
            const double2 q = b.xy[i];
            const double d = hypot(px - q.x, py - q.y);
            m = d < m ? d : m;
        }
        return gmin<G>(m);
    }
    case GPK_GEOM_LINESTRING:
        return point_linestring_distance<G>(b.xy, b.geom_off[j], b.geom_off[j + 1], px, py, lane);
    case GPK_GEOM_MULTILINESTRING: {
        double m = DBL_MAX;
        for (int l = b.geom_off[j]; l < b.geom_off[j + 1]; ++l) {
            const double d = point_linestring_distance<G>(b.xy, b.ring_off[l], b.ring_off[l + 1], px, py, lane);
            m = d < m ? d : m;
        }
        return m;
    }
    default: {
        int p0, p1;
        dev::geom_parts(b, j, p0, p1);
        double m = DBL_MAX;
        for (int p = p0; p < p1; ++p) {
            int r0, r1, pos;
            dev::part_rings(b, p, r0, r1);
            const double d = point_polygon<G, true>(b, r0, r1, px, py, lane, &pos);
            m = d < m ? d : m;
        }
        return m;
    }
    }
}

// coordinates a row of `b` owns (the work estimate used to bin rows)
__device__ __forceinline__ int geom_work(const DevGeo& b, int64_t j) {
    switch (b.type) {
    case GPK_GEOM_POINT: return 1;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT: return b.geom_off[j + 1] - b.geom_off[j];
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING: return b.ring_off[b.geom_off[j + 1]] - b.ring_off[b.geom_off[j]];
    default: return b.ring_off[b.part_off[b.geom_off[j + 1]]] - b.ring_off[b.part_off[b.geom_off[j]]];
    }
}

// Row-wise distance.  Rows are ragged (C3: 4..256 segments, log-uniform), and a wave pays for its longest
// row, so each work-group first BINS its tile of rows by log2(vertex count) with a counting sort in LDS
// (the north star's "binning on vertex count so lanes in a wave see similar work"), then G-lane groups walk
// the tile in bin order, longest first.  Inputs and outputs stay in the tile's contiguous row range, so the
// permutation costs no extra HBM traffic.  Which group handles which row depends on atomic order, the value
// written for a row does not.
constexpr int DIST_TILE = 2048, DIST_BINS = 24;
template <int G, int KIND>
__global__ __launch_bounds__(256) void distance_kernel(DevGeo pts, DevGeo other, const uint32_t* __restrict__ rows,
                                                       double* __restrict__ out) {
    __shared__ uint16_t s_perm[DIST_TILE];
    __shared__ int s_cnt[DIST_BINS], s_start[DIST_BINS];
    const int tid = threadIdx.x, lane = tid & (G - 1);
    const int64_t n = pts.n_geoms;
    for (int64_t base = (int64_t)blockIdx.x * DIST_TILE; base < n; base += (int64_t)gridDim.x * DIST_TILE) {
        const int tile_rows = (int)(n - base < DIST_TILE ? n - base : DIST_TILE);
        if (tid < DIST_BINS) s_cnt[tid] = 0;
        __syncthreads();
        int bin[DIST_TILE / 256], rank[DIST_TILE / 256];
#pragma unroll
        for (int k = 0; k < DIST_TILE / 256; ++k) {
            const int li = k * 256 + tid;
            bin[k] = -1;
            if (li < tile_rows) {
                const int64_t i = base + li;
                const int64_t j = rows ? (int64_t)rows[i] : i;
                const int w = (uint64_t)j < (uint64_t)other.n_geoms ? geom_work(other, j) : 0;  // out-of-range map entry: a null row
                bin[k] = w <= 1 ? 0 : (32 - __clz(w - 1));  // ceil(log2(w)); w < 2^23 -> bin < DIST_BINS
                if (bin[k] >= DIST_BINS) bin[k] = DIST_BINS - 1;
                rank[k] = atomicAdd(&s_cnt[bin[k]], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {  // longest rows first
            int run = 0;
            for (int b = DIST_BINS - 1; b >= 0; --b) {
                s_start[b] = run;
                run += s_cnt[b];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DIST_TILE / 256; ++k)
            if (bin[k] >= 0) s_perm[s_start[bin[k]] + rank[k]] = (uint16_t)(k * 256 + tid);
        __syncthreads();
        for (int e = tid / G; e < tile_rows; e += 256 / G) {
            const int64_t i = base + s_perm[e];
            const int64_t j = rows ? (int64_t)rows[i] : i;
            const double2 p = pts.xy[i];
            double d;
            if (!dev::valid_row(pts.validity, i) || !dev::row_ok(other, j) || isnan(p.x) || isnan(p.y))
                d = NAN;
            else
                d = point_geom_distance<G, KIND>(other, j, p.x, p.y, lane);
            if (lane == 0) out[i] = d;
        }
        __syncthreads();
    }
}

// ---- distance, rows grouped by target (point x LINESTRING with a row map) ------------------------------
// When many rows point at the same linestring (C3: 10M rows -> 100k linestrings) the row-major kernel above
// re-reads every linestring ~100 times from L2/MALL (10 GB of gather traffic for 98 MB of coordinates) and
// spends lanes on per-row reductions.  Grouped: a counting sort orders the rows by target; one WAVE owns one
// target, stages its vertices in LDS once, and every lane walks the segments for its own point (LDS broadcast
// reads, no cross-lane reduction).  Which lane handles which row depends on atomic order; the value written
// for a row does not.
// Both passes aggregate equal keys inside a wave before touching memory (clustered row maps put many rows of
// one target in the same wave: 64 same-address atomics would serialise): up to two rounds of "lanes that share
// the first remaining lane's key go together", the rest falls back to one atomic per lane.
// Map entries >= L (no such target) are counted in the extra bucket L, which no work item covers; the scatter pass gives
// those rows their result directly: NaN, the answer of a null row.
template <bool SCATTER>
__global__ __launch_bounds__(256) void dist_sort_kernel(const uint32_t* __restrict__ rows, int64_t n, uint32_t L, int32_t* __restrict__ counter,
                                                        uint32_t* __restrict__ perm, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool todo = i < n;
    uint32_t key = todo ? rows[i] : 0u;
    if (key >= L) {
        key = L;
        if (SCATTER && todo) out[i] = NAN;
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const unsigned long long rest = __ballot(todo);
        if (!rest) break;
        const int leader = __ffsll((long long)rest) - 1;
        const uint32_t lkey = __shfl(key, leader, 64);
        const bool mine = todo && key == lkey;
        const unsigned long long same = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counter[lkey], (int)__popcll(same));
        base = __shfl(base, leader, 64);
        if (mine) {
            if (SCATTER) perm[base + (int)__popcll(same & ((1ull << lane) - 1ull))] = (uint32_t)i;
            todo = false;
        }
    }
    if (todo) {
        const int slot = atomicAdd(&counter[key], 1);
        if (SCATTER) perm[slot] = (uint32_t)i;
    }
}

// How local is the row map?  Counts the lanes whose key is within one of their right neighbour's (ascending or equal
// runs: `i mod L`, sorted maps, blocks of one target).  Local maps make the counting sort's atomics land on neighbouring
// counters (one cache line per wave); a random map makes every lane's atomic its own line, and a radix sort wins.
// A sample is enough: DIST_PROBE_BLOCKS chunks of 256 rows spread evenly over the map (one atomic per wave on a single
// word — sampling every row would cost 1.9 ms in that atomic alone).
constexpr int DIST_PROBE_BLOCKS = 256;
__global__ __launch_bounds__(256) void dist_probe_kernel(const uint32_t* __restrict__ rows, int64_t n, unsigned long long* __restrict__ local) {
    const int64_t i = (int64_t)blockIdx.x * (n / gridDim.x) + threadIdx.x;
    const uint32_t k = i < n ? rows[i] : 0u, kn = i + 1 < n ? rows[i + 1] : k;
    const unsigned long long m = __ballot(i + 1 < n && (kn - k <= 1u));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(local, (unsigned long long)__popcll(m));
}
// radix-sort grouping: row numbers, and the map clamped to L (entries without a target sort behind every real one and get NaN)
__global__ void iota_clamp_kernel(const uint32_t* __restrict__ rows, uint32_t L, uint32_t* __restrict__ iota, uint32_t* __restrict__ keys,
                                  double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    iota[i] = (uint32_t)i;
    const uint32_t k = rows[i];
    keys[i] = k < L ? k : L;
    if (k >= L) out[i] = NAN;
}
// off[t] = first position of key t in the sorted keys (t = 0..L), cnt[t] = rows of target t
__global__ void sorted_offsets_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t L, int32_t* __restrict__ off,
                                      int32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > L) return;
    auto lower = [&](uint32_t key) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    const int64_t a = lower((uint32_t)t);
    off[t] = (int32_t)a;
    if (t < L) cnt[t] = (int32_t)(lower((uint32_t)t + 1u) - a);
}

__global__ void dist_batches_kernel(const int32_t* __restrict__ cnt, int64_t L, int32_t* __restrict__ nb) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < L) nb[t] = (cnt[t] + 63) >> 6;
}

// squared distance to one segment with the start-point terms carried from the previous segment (|p - a|^2 of
// this segment is |p - b|^2 of the last one).  Explicit fma: this kernel's results only need the 1e-9 contract;
// the zero / non-zero outcome is decided separately by the exact re-walk below.
struct SegState {
    double ax, ay, qx, qy, na2;  // vertex a, p - a, |p - a|^2
};
__device__ __forceinline__ void seg_step(SegState& st, double2 b, double px, double py, Frac& best, int& maybe) {
    const double rx = px - b.x, ry = py - b.y;
    const double nb2 = __builtin_fma(rx, rx, ry * ry);
    const double dx = b.x - st.ax, dy = b.y - st.ay;
    const double d2 = __builtin_fma(dx, dx, dy * dy);
    const double dot = __builtin_fma(st.qx, dx, st.qy * dy);
    const double cross = __builtin_fma(st.qx, dy, -(st.qy * dx));
    Frac c;
    const bool at_a = d2 == 0.0 || dot <= 0.0, at_b = dot >= d2;
    c.num = at_a ? st.na2 : (at_b ? nb2 : cross * cross);
    c.den = (at_a || at_b) ? 1.0 : d2;
    if (frac_less(c, best)) best = c;
    // candidates for upstream's "point is on the linestring" short-circuit: a vertex hit, or |tx - ty| within reach
    // of f64::EPSILON (tx - ty == cross / (dx dy)); everything else is certainly not on the segment
    maybe |= (int)(st.na2 == 0.0 || nb2 == 0.0 || fabs(cross) <= 1.7763568394002505e-15 * fabs(dx * dy));
    st.ax = b.x;
    st.ay = b.y;
    st.qx = rx;
    st.qy = ry;
    st.na2 = nb2;
}

constexpr int DG_CHUNK = 256;  // vertices staged per wave (4 KB: keeps ~10 work-groups per CU resident); longer linestrings are streamed in passes
__global__ __launch_bounds__(256) void distance_grouped_kernel(DevGeo pts, DevGeo ls, const int32_t* __restrict__ off,
                                                               const int32_t* __restrict__ item_off,
                                                               const uint32_t* __restrict__ perm, double* __restrict__ out) {
    __shared__ double2 s_xy[4][DG_CHUNK + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double2* sv = s_xy[wave];
    const int64_t L = ls.n_geoms;
    const int64_t n_items = item_off[L];
    // one work item = (target, batch of 64 rows of that target)
    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < n_items; it += (int64_t)gridDim.x * 4) {
        int64_t lo = 0, hi = L;  // largest t with item_off[t] <= it
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)item_off[mid] <= it)
                lo = mid;
            else
                hi = mid;
        }
        const int64_t t = lo;
        const int r = off[t] + (int)(it - item_off[t]) * 64 + lane;
        const bool active = r < off[t + 1];
        const int c0 = ls.geom_off[t], nv = ls.geom_off[t + 1] - c0;
        const uint32_t i = active ? perm[r] : 0u;
        double2 p = make_double2(NAN, NAN);
        if (active && dev::valid_row(pts.validity, i)) p = pts.xy[i];
        Frac best{INFINITY, 1.0};
        int maybe = 0;
        SegState st{0, 0, 0, 0, 0};
        for (int cb = 0; cb < nv; cb += DG_CHUNK) {  // one pass unless the linestring exceeds DG_CHUNK + 1 vertices
            const int m = nv - cb < DG_CHUNK + 1 ? nv - cb : DG_CHUNK + 1;
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < m; k += 64) sv[k] = ls.xy[c0 + cb + k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (cb == 0) {
                const double2 a = sv[0];
                st.ax = a.x;
                st.ay = a.y;
                st.qx = p.x - a.x;
                st.qy = p.y - a.y;
                st.na2 = __builtin_fma(st.qx, st.qx, st.qy * st.qy);
                if (nv == 1) maybe |= (int)(st.na2 == 0.0);
            }
#pragma unroll 4
            for (int k = 1; k < m; ++k) seg_step(st, sv[k], p.x, p.y, best, maybe);
        }
        // exact replay of line_string_contains_point for the (rare) lanes that came close
        int eps_hit = 0;
        if (__any(maybe)) {
            if (nv <= DG_CHUNK + 1) {
                if (maybe) {
                    if (nv == 1) eps_hit = (int)(sv[0].x == p.x && sv[0].y == p.y);
                    for (int k = 0; k + 1 < nv && !eps_hit; ++k) {
                        const double2 a = sv[k], b = sv[k + 1];
                        const double cr = (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
                        eps_hit = (int)((a.x == p.x && a.y == p.y) || (b.x == p.x && b.y == p.y) ||
                                        segment_contains_eps(p.x, p.y, a.x, a.y, b.x, b.y, cr, (b.x - a.x) * (b.y - a.y)));
                    }
                }
            } else if (maybe) {  // streamed linestring: replay from global memory
                for (int k = 0; k + 1 < nv && !eps_hit; ++k) {
                    const double2 a = ls.xy[c0 + k], b = ls.xy[c0 + k + 1];
                    const double cr = (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
                    eps_hit = (int)((a.x == p.x && a.y == p.y) || (b.x == p.x && b.y == p.y) ||
                                    segment_contains_eps(p.x, p.y, a.x, a.y, b.x, b.y, cr, (b.x - a.x) * (b.y - a.y)));
                }
            }
        }
        if (active) {
            double d;
            if (!dev::valid_row(ls.validity, t) || isnan(p.x) || isnan(p.y))
                d = NAN;
            else if (nv == 0 || eps_hit)
                d = 0.0;
            else
                d = frac_sqrt(best);
            out[i] = d;
        }
    }
}

// ---- row-wise predicates: point x polygonal (cooperative), everything else one lane per row -------------
template <int G>
__global__ __launch_bounds__(256) void point_poly_predicate_kernel(DevGeo pts, DevGeo polys,
                                                                    const uint32_t* __restrict__ rows,
                                                                    bool rows_index_polys, bool boundary_counts,
                                                                    uint8_t* __restrict__ out, int64_t n_out) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t i = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; i < n_out; i += groups) {
        // row i of the left operand pairs with row rows[i] (or i) of the right operand
        const int64_t ip = rows_index_polys ? i : (rows ? (int64_t)rows[i] : i);
        const int64_t jp = rows_index_polys ? (rows ? (int64_t)rows[i] : i) : i;
        const bool ok = dev::row_ok(pts, ip) && dev::row_ok(polys, jp);
        const double2 p = ok ? pts.xy[ip] : make_double2(NAN, NAN);
        bool hit = false;
        if (ok && !isnan(p.x) && !isnan(p.y)) {
            int p0, p1;
            dev::geom_parts(polys, jp, p0, p1);
            for (int q = p0; q < p1 && !hit; ++q) {
                int r0, r1, pos;
                dev::part_rings(polys, q, r0, r1);
                (void)point_polygon<G, false>(polys, r0, r1, p.x, p.y, lane, &pos);
                hit = boundary_counts ? pos != dev::POS_OUTSIDE : pos == dev::POS_INSIDE;
            }
        }
        if (lane == 0) out[i] = hit;
    }
}

// row-wise intersects(polygon, polygon): 16 lanes per row (gpk_polypoly.h, same routine as the join's refine)
constexpr int PP_GS = 16;
__global__ __launch_bounds__(256) void poly_poly_intersects_kernel(DevGeo a, DevGeo b, const uint32_t* __restrict__ rows,
                                                                    uint8_t* __restrict__ out) {
    __shared__ double4 seg_lists[(256 / PP_GS) * 2 * PP_LIST];
    const int lane = threadIdx.x & (PP_GS - 1);
    double4* seg_list = seg_lists + (threadIdx.x / PP_GS) * 2 * PP_LIST;
    const int64_t groups = (int64_t)gridDim.x * (256 / PP_GS);
    for (int64_t i = (int64_t)blockIdx.x * (256 / PP_GS) + threadIdx.x / PP_GS; i < a.n_geoms; i += groups) {
        const int64_t j = rows ? (int64_t)rows[i] : i;
        bool hit = false;
        if (dev::valid_row(a.validity, i) && dev::row_ok(b, j))
            hit = polygonal_intersects_polygonal_group<PP_GS>(a, i, b, j, lane, seg_list);
        if (lane == 0) out[i] = hit;
    }
}

// row-wise contains(polygonal, polygonal) (gpk_contains.h); within(a, b) = contains(b, a): `swap` exchanges the roles
__global__ __launch_bounds__(256) void poly_poly_contains_kernel(DevGeo a, DevGeo b, const uint32_t* __restrict__ rows, bool swap,
                                                                  uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & (PP_GS - 1);
    const int64_t groups = (int64_t)gridDim.x * (256 / PP_GS);
    for (int64_t i = (int64_t)blockIdx.x * (256 / PP_GS) + threadIdx.x / PP_GS; i < a.n_geoms; i += groups) {
        const int64_t j = rows ? (int64_t)rows[i] : i;
        bool hit = false;
        if (dev::valid_row(a.validity, i) && dev::row_ok(b, j))
            hit = swap ? cont::polygonal_contains_polygonal_group<PP_GS>(b, j, a, i, lane)
                       : cont::polygonal_contains_polygonal_group<PP_GS>(a, i, b, j, lane);
        if (lane == 0) out[i] = hit;
    }
}

// row-wise contains(lineal, point) / within(point, lineal): one lane per row (gpk_lineal.h)
__global__ void lineal_point_contains_kernel(DevGeo a, DevGeo b, const uint32_t* __restrict__ rows, bool lineal_is_a,
                                             uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_geoms) return;
    const int64_t j = rows ? (int64_t)rows[i] : i;
    bool hit = false;
    if (dev::valid_row(a.validity, i) && dev::row_ok(b, j)) {
        const double2 p = lineal_is_a ? b.xy[j] : a.xy[i];
        if (p.x == p.x && p.y == p.y) hit = lineal_is_a ? lineal_contains_point(a, i, p.x, p.y) : lineal_contains_point(b, j, p.x, p.y);
    }
    out[i] = hit;
}

__global__ void point_point_equal_kernel(DevGeo a, DevGeo b, const uint32_t* __restrict__ rows, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_geoms) return;
    const int64_t j = rows ? (int64_t)rows[i] : i;
    const bool ok = dev::valid_row(a.validity, i) && dev::row_ok(b, j);
    const double2 p = a.xy[i], q = ok ? b.xy[j] : make_double2(NAN, NAN);
    out[i] = ok && p.x == q.x && p.y == q.y;
}

__global__ void fill_u8_kernel(uint8_t* out, int64_t n, uint8_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// ---- host side -----------------------------------------------------------------------------------
static int pick_group_rows(const DevGeo& g) {
    const double mean = g.n_geoms > 0 ? (double)g.n_coords / (double)g.n_geoms : 1.0;
    int G = 1;
    while (G < 64 && G * 2 * 8 <= mean) G <<= 1;  // ~8 segments per lane: short reductions, >= 64 B contiguous per group
    return G;
}
static dim3 coop_grid(int64_t n_rows, int G) {
    const int64_t per_block = 256 / G;
    int64_t blocks = (n_rows + per_block - 1) / per_block;
    const int64_t cap = (int64_t)cu_count() * 32;
    if (blocks > cap) blocks = cap;
    return dim3((unsigned)(blocks > 0 ? blocks : 1));
}

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_distance_rowwise(const gpk_geoarray* a, const gpk_geoarray* b, const uint32_t* b_rows, double* out,
                             int32_t out_space, void* stream) {
    if (!a || !b || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const gpk_geoarray *pts = a, *other = b;
    if (a->d.type != GPK_GEOM_POINT) {
        if (b->d.type != GPK_GEOM_POINT)
            return fail(GPK_ERR_MISMATCHED_GEOMETRY, "distance: one side must be a POINT array (found types %d, %d)",
                        a->d.type, b->d.type);
        if (b_rows) return fail(GPK_ERR_INVALID_ARGUMENT, "distance: b_rows requires the POINT array on the left");
        pts = b;
        other = a;
    }
    if (!b_rows && a->d.n_geoms != b->d.n_geoms)
        return fail(GPK_ERR_INVALID_ARGUMENT, "distance: row counts differ (%lld vs %lld)", (long long)a->d.n_geoms,
                    (long long)b->d.n_geoms);
    const int64_t n = pts->d.n_geoms;
    if (n == 0) return GPK_OK;
    const size_t ob = sizeof(double) * (size_t)n;
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const uint32_t* rows_dev = b_rows;
    double* out_dev = out;
    if (host_out) {
        GPK_TRY(workspace().begin(align256(ob) + (b_rows ? align256(sizeof(uint32_t) * (size_t)n) : 0) + 512));
        out_dev = (double*)workspace().take(ob);
        if (b_rows) {
            uint32_t* r = (uint32_t*)workspace().take(sizeof(uint32_t) * (size_t)n);
            GPK_HIP(hipMemcpyAsync(r, b_rows, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, s));
            rows_dev = r;
        }
    }
    // many rows per linestring: group the rows by target and stage each linestring in LDS once
    if (b_rows && other->d.type == GPK_GEOM_LINESTRING && other->d.n_geoms > 0 && n >= 8 * other->d.n_geoms) {
        const int64_t L = other->d.n_geoms;
        void* tmp = nullptr;  // off | cursor | cnt | batches | item_off (L+1 each) | perm (n) | scan totals
        const size_t ib = align256(sizeof(int32_t) * (size_t)(L + 1));
        const size_t total = 5 * ib + align256(sizeof(uint32_t) * (size_t)n) + align256(sizeof(unsigned long long) * (size_t)((L + 255) / 256 + 4));
        GPK_HIP(hipMalloc(&tmp, total));
        void* sort_tmp_free = nullptr;
        auto fin = [&](int32_t rc) {
            (void)hipStreamSynchronize(s);
            (void)hipFree(tmp);
            if (sort_tmp_free) (void)hipFree(sort_tmp_free);
            return rc;
        };
        char* base = (char*)tmp;
        int32_t* g_off = (int32_t*)base;
        int32_t* g_cur = (int32_t*)(base + ib);
        int32_t* g_cnt = (int32_t*)(base + 2 * ib);
        int32_t* g_nb = (int32_t*)(base + 3 * ib);
        int32_t* g_item = (int32_t*)(base + 4 * ib);
        uint32_t* perm = (uint32_t*)(base + 5 * ib);
        unsigned long long* btot = (unsigned long long*)(base + 5 * ib + align256(sizeof(uint32_t) * (size_t)n));
        void* sort_tmp = nullptr;  // radix-sort path only
        auto run = [&]() -> int32_t {
            const dim3 rg((unsigned)((n + 255) / 256));
            // rows grouped by target: counting sort for local row maps, radix sort (rocPRIM) for scattered ones
            unsigned long long h_local = 0, *d_local = btot;
            GPK_HIP(hipMemsetAsync(d_local, 0, sizeof(unsigned long long), s));
            const int probe_blocks = n >= 256 * DIST_PROBE_BLOCKS ? DIST_PROBE_BLOCKS : 1;
            GPK_LAUNCH("gpk_dist_probe", dist_probe_kernel, dim3((unsigned)probe_blocks), dim3(256), 0, s, rows_dev, n, d_local);
            GPK_HIP(hipMemcpyAsync(&h_local, d_local, sizeof h_local, hipMemcpyDeviceToHost, s));
            GPK_HIP(hipStreamSynchronize(s));
            const int64_t sampled = (int64_t)probe_blocks * 256 < n ? (int64_t)probe_blocks * 256 : n;
            if (2 * (int64_t)h_local >= sampled) {
                GPK_HIP(hipMemsetAsync(g_cnt, 0, sizeof(int32_t) * (size_t)(L + 1), s));
                GPK_LAUNCH("gpk_dist_hist", dist_sort_kernel<false>, rg, dim3(256), 0, s, rows_dev, n, (uint32_t)L, g_cnt, (uint32_t*)nullptr, (double*)nullptr);
                GPK_TRY(exclusive_scan_i32(g_cnt, L, g_off, g_cur, btot, s));  // g_cur[L] = g_off[L]: the bucket of map entries without a target
                GPK_HIP(hipMemcpyAsync(g_cur + L, g_off + L, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
                GPK_LAUNCH("gpk_dist_scatter", dist_sort_kernel<true>, rg, dim3(256), 0, s, rows_dev, n, (uint32_t)L, g_cur, perm, out_dev);
            } else {
                int bits = 1;
                while (bits < 32 && (1ll << bits) < L + 1) ++bits;  // keys 0..L (L = "no such target")
                size_t tb = 0;
                GPK_HIP(rocprim::radix_sort_pairs(nullptr, tb, rows_dev, (uint32_t*)nullptr, (const uint32_t*)nullptr, perm, (size_t)n, 0, bits, s));
                const size_t nbytes = align256(sizeof(uint32_t) * (size_t)n);
                GPK_HIP(hipMalloc(&sort_tmp, 3 * nbytes + tb + 256));
                uint32_t* iota = (uint32_t*)sort_tmp;
                uint32_t* keys_sorted = (uint32_t*)((char*)sort_tmp + nbytes);
                uint32_t* keys_in = (uint32_t*)((char*)sort_tmp + 2 * nbytes);
                void* rp_tmp = (char*)sort_tmp + 3 * nbytes;
                GPK_LAUNCH("gpk_dist_iota", iota_clamp_kernel, rg, dim3(256), 0, s, rows_dev, (uint32_t)L, iota, keys_in, out_dev, n);
                GPK_HIP(rocprim::radix_sort_pairs(rp_tmp, tb, (const uint32_t*)keys_in, keys_sorted, (const uint32_t*)iota, perm, (size_t)n, 0, bits, s));
                GPK_LAUNCH("gpk_dist_offsets", sorted_offsets_kernel, dim3((unsigned)((L + 256) / 256)), dim3(256), 0, s, (const uint32_t*)keys_sorted, n,
                           L, g_off, g_cnt);
            }
            GPK_LAUNCH("gpk_dist_batches", dist_batches_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, (const int32_t*)g_cnt, L, g_nb);
            GPK_TRY(exclusive_scan_i32(g_nb, L, g_item, nullptr, btot, s));
            int64_t blocks = (n / 64 + L + 3) / 4;  // upper bound on the number of (target, 64-row batch) items
            const int64_t cap = (int64_t)cu_count() * 16;
            if (blocks > cap) blocks = cap;
            GPK_LAUNCH("gpk_distance_grouped", distance_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pts->d, other->d,
                       (const int32_t*)g_off, (const int32_t*)g_item, (const uint32_t*)perm, out_dev);
            return GPK_OK;
        };
        const int32_t rc = run();
        sort_tmp_free = sort_tmp;
        if (rc != GPK_OK) return fin(rc);
        const int32_t rc2 = copy_out(out, out_space, out_dev, ob, s);
        return fin(rc2);
    }
    int G = other->d.type == GPK_GEOM_POINT ? 1 : pick_group_rows(other->d);
    G = G <= 1 ? 1 : (G <= 8 ? 8 : 32);  // instantiated group sizes
    int64_t n_tiles = (n + DIST_TILE - 1) / DIST_TILE;
    if (n_tiles > (int64_t)cu_count() * 16) n_tiles = (int64_t)cu_count() * 16;
    const dim3 grid((unsigned)n_tiles), block(256);
#define DIST_LAUNCH(GG, KK) GPK_LAUNCH("gpk_distance", (distance_kernel<GG, KK>), grid, block, 0, s, pts->d, other->d, rows_dev, out_dev)
#define DIST_BY_G(KK)                  \
    do {                               \
        if (G == 1)                    \
            DIST_LAUNCH(1, KK);        \
        else if (G == 8)               \
            DIST_LAUNCH(8, KK);        \
        else                           \
            DIST_LAUNCH(32, KK);       \
    } while (0)
    switch (other->d.type) {
    case GPK_GEOM_POINT: DIST_LAUNCH(1, GPK_GEOM_POINT); break;
    case GPK_GEOM_MULTIPOINT: DIST_BY_G(GPK_GEOM_MULTIPOINT); break;
    case GPK_GEOM_LINESTRING: DIST_BY_G(GPK_GEOM_LINESTRING); break;
    case GPK_GEOM_MULTILINESTRING: DIST_BY_G(GPK_GEOM_MULTILINESTRING); break;
    case GPK_GEOM_POLYGON: DIST_BY_G(GPK_GEOM_POLYGON); break;
    default: DIST_BY_G(GPK_GEOM_MULTIPOLYGON); break;
    }
#undef DIST_BY_G
#undef DIST_LAUNCH
    return copy_out(out, out_space, out_dev, ob, s);
}

int32_t gpk_predicate_rowwise(const gpk_geoarray* a, const gpk_geoarray* b, const uint32_t* b_rows, int32_t predicate,
                              uint8_t* out, int32_t out_space, void* stream) {
    if (!a || !b || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    if (predicate != GPK_PRED_INTERSECTS && predicate != GPK_PRED_CONTAINS && predicate != GPK_PRED_WITHIN)
        return fail(GPK_ERR_INVALID_ARGUMENT, "unknown predicate %d", predicate);
    GPK_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    if (!b_rows && a->d.n_geoms != b->d.n_geoms)
        return fail(GPK_ERR_INVALID_ARGUMENT, "predicate: row counts differ (%lld vs %lld)", (long long)a->d.n_geoms,
                    (long long)b->d.n_geoms);
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    const bool host_out = out_space != GPK_MEM_DEVICE;
    const uint32_t* rows_dev = b_rows;
    uint8_t* out_dev = out;
    if (host_out) {
        GPK_TRY(workspace().begin(align256((size_t)n) + (b_rows ? align256(sizeof(uint32_t) * (size_t)n) : 0) + 512));
        out_dev = (uint8_t*)workspace().take((size_t)n);
        if (b_rows) {
            uint32_t* r = (uint32_t*)workspace().take(sizeof(uint32_t) * (size_t)n);
            GPK_HIP(hipMemcpyAsync(r, b_rows, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, s));
            rows_dev = r;
        }
    }
    const int ta = a->d.type, tb = b->d.type;
    const dim3 block(256);
    const dim3 flat((unsigned)((n + 255) / 256));

    // contains(a, b): a polygonal, b point -> Inside.  within(a, b) == contains(b, a): a point, b polygonal.
    // intersects: either order, boundary counts.
    auto is_lineal = [](int t) { return t == GPK_GEOM_LINESTRING || t == GPK_GEOM_MULTILINESTRING; };
    const bool a_poly_b_pt = is_polygonal(ta) && tb == GPK_GEOM_POINT;
    const bool a_pt_b_poly = ta == GPK_GEOM_POINT && is_polygonal(tb);
    bool run_pp = false, boundary = false, rows_index_polys = false;
    const gpk_geoarray *pts = nullptr, *polys = nullptr;
    if (predicate == GPK_PRED_INTERSECTS && (a_poly_b_pt || a_pt_b_poly)) {
        run_pp = true;
        boundary = true;
    } else if (predicate == GPK_PRED_CONTAINS && a_poly_b_pt) {
        run_pp = true;
    } else if (predicate == GPK_PRED_WITHIN && a_pt_b_poly) {
        run_pp = true;
    }
    if (run_pp) {
        pts = a_pt_b_poly ? a : b;
        polys = a_pt_b_poly ? b : a;
        rows_index_polys = a_pt_b_poly;  // b_rows indexes b; b is the polygon side when a is the point side
        const int G = pick_group_rows(polys->d);
        const dim3 grid = coop_grid(n, G);
#define PP(GG)                                                                                                       \
    GPK_LAUNCH("gpk_point_poly_predicate", point_poly_predicate_kernel<GG>, grid, block, 0, s, pts->d, polys->d,    \
               rows_dev, rows_index_polys, boundary, out_dev, n)
        switch (G) {
        case 1: PP(1); break;
        case 2: PP(2); break;
        case 4: PP(4); break;
        case 8: PP(8); break;
        case 16: PP(16); break;
        case 32: PP(32); break;
        default: PP(64); break;
        }
#undef PP
    } else if (predicate == GPK_PRED_INTERSECTS && is_polygonal(ta) && is_polygonal(tb)) {
        GPK_LAUNCH("gpk_poly_poly_intersects", poly_poly_intersects_kernel, coop_grid(n, PP_GS), block, 0, s, a->d, b->d, rows_dev, out_dev);
    } else if (ta == GPK_GEOM_POINT && tb == GPK_GEOM_POINT) {
        GPK_LAUNCH("gpk_point_point_equal", point_point_equal_kernel, flat, block, 0, s, a->d, b->d, rows_dev, out_dev);
    } else if (is_polygonal(ta) && is_polygonal(tb)) {  // contains / within
        GPK_LAUNCH("gpk_poly_poly_contains", poly_poly_contains_kernel, coop_grid(n, PP_GS), block, 0, s, a->d, b->d, rows_dev,
                   predicate == GPK_PRED_WITHIN, out_dev);
    } else if ((predicate == GPK_PRED_CONTAINS && is_lineal(ta) && tb == GPK_GEOM_POINT) ||
               (predicate == GPK_PRED_WITHIN && ta == GPK_GEOM_POINT && is_lineal(tb))) {
        GPK_LAUNCH("gpk_lineal_point_contains", lineal_point_contains_kernel, flat, block, 0, s, a->d, b->d, rows_dev,
                   predicate == GPK_PRED_CONTAINS, out_dev);
    } else {
        // combinations the reference's dispatch table maps to `false` (spatial_index.rs:136)
        GPK_LAUNCH("gpk_fill_u8", fill_u8_kernel, flat, block, 0, s, out_dev, n, (uint8_t)0);
    }
    return copy_out(out, out_space, out_dev, (size_t)n, s);
}

}  // extern "C"
