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
// spends lanes on per-row reductions.  Grouped: the rows are ordered ONCE per row map (gpk_rowmap) by target, the
// targets themselves by descending vertex count, and the ordered rows are cut into chunks of 64.  One WAVE owns one
// chunk: its lanes' targets are a handful of neighbours in the length order (equally long linestrings, so every lane
// walks about the same number of segments and no lane idles behind a short batch), the wave stages a window of each
// of those linestrings in LDS once, and every lane walks the segments of ITS target for its own point with LDS reads
// (lanes of one target read one address: a broadcast) — no cross-lane reduction.  The staged coordinates are the
// "ring coordinates in LDS per work-group" of the north star.

// Pass 1 / 3 of the map build: histogram of the map over targets, then the scatter of the rows into target order.  Both
// passes aggregate equal keys inside a wave before touching memory (clustered row maps put many rows of one target in the
// same wave: 64 same-address atomics would serialise): up to two rounds of "lanes that share the first remaining lane's key
// go together", the rest falls back to one atomic per lane.  Map entries >= L (no such target) are counted in the extra
// bucket L, which sorts behind every real target and is never visited: those rows get NaN, the answer of a null row.
// SCATTER: counter[] holds the cursors of the targets IN LENGTH ORDER (cursor of target t at counter[rank[t]]).
template <bool SCATTER>
__global__ __launch_bounds__(256) void rowmap_sort_kernel(const uint32_t* __restrict__ rows, int64_t n, uint32_t L, const uint32_t* __restrict__ rank,
                                                          int32_t* __restrict__ counter, uint32_t* __restrict__ perm, uint32_t* __restrict__ tsorted) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool todo = i < n;
    uint32_t key = todo ? rows[i] : 0u;
    if (key >= L) key = L;
    const uint32_t slot_key = SCATTER ? (key < L ? rank[key] : L) : key;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const unsigned long long rest = __ballot(todo);
        if (!rest) break;
        const int leader = __ffsll((long long)rest) - 1;
        const uint32_t lkey = __shfl(slot_key, leader, 64);
        const bool mine = todo && slot_key == lkey;
        const unsigned long long same = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counter[lkey], (int)__popcll(same));
        base = __shfl(base, leader, 64);
        if (mine) {
            if (SCATTER) {
                const int at = base + (int)__popcll(same & ((1ull << lane) - 1ull));
                perm[at] = (uint32_t)i;
                tsorted[at] = key;
            }
            todo = false;
        }
    }
    if (todo) {
        const int at = atomicAdd(&counter[slot_key], 1);
        if (SCATTER) {
            perm[at] = (uint32_t)i;
            tsorted[at] = key;
        }
    }
}
// pass 2: sort keys of the targets — descending vertex count, ties by id (the key is unique, so the order is deterministic)
__global__ void rowmap_keys_kernel(DevGeo ls, uint32_t L, unsigned long long* __restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const uint32_t nv = (uint32_t)(ls.geom_off[t + 1] - ls.geom_off[t]);
    keys[t] = ((unsigned long long)(0xFFFFFFFFu - nv) << 32) | t;
}
// rank[target] = position in the length order; cnt_sorted[r] = rows of the r-th target in that order
__global__ void rowmap_rank_kernel(const unsigned long long* __restrict__ sorted, uint32_t L, const int32_t* __restrict__ cnt, uint32_t* __restrict__ rank,
                                   int32_t* __restrict__ cnt_sorted) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > L) return;
    if (r == L) {
        cnt_sorted[L] = cnt[L];  // rows without a target
        return;
    }
    const uint32_t t = (uint32_t)sorted[r];
    rank[t] = r;
    cnt_sorted[r] = cnt[t];
}
__global__ void rowmap_nan_kernel(const uint32_t* __restrict__ perm, int64_t from, int64_t n, double* __restrict__ out) {
    const int64_t i = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i]] = NAN;
}

// How local is the row map?  Counts the lanes whose key is within one of their right neighbour's (ascending or equal
// runs: `i mod L`, sorted maps, blocks of one target).  Local maps make the counting sort's atomics land on neighbouring
// counters (one cache line per wave); a random map makes every lane's atomic its own line, and a radix sort wins.
// A sample is enough: DIST_PROBE_BLOCKS chunks of 256 rows spread evenly over the map.
constexpr int DIST_PROBE_BLOCKS = 256;
__global__ __launch_bounds__(256) void dist_probe_kernel(const uint32_t* __restrict__ rows, int64_t n, unsigned long long* __restrict__ local) {
    const int64_t i = (int64_t)blockIdx.x * (n / gridDim.x) + threadIdx.x;
    const uint32_t k = i < n ? rows[i] : 0u, kn = i + 1 < n ? rows[i + 1] : k;
    const unsigned long long m = __ballot(i + 1 < n && (kn - k <= 1u));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(local, (unsigned long long)__popcll(m));
}
// radix path of the map build: sort keys = position of the row's target in the length order (L = no such target)
__global__ void rowmap_radix_keys_kernel(const uint32_t* __restrict__ rows, uint32_t L, const uint32_t* __restrict__ rank, uint32_t* __restrict__ keys,
                                         uint32_t* __restrict__ iota, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = rows[i];
    keys[i] = k < L ? rank[k] : L;
    iota[i] = (uint32_t)i;
}
// sorted rank -> target id (order[r] = the low word of the r-th sorted target key)
__global__ void rowmap_targets_kernel(const uint32_t* __restrict__ keys_sorted, const unsigned long long* __restrict__ order, uint32_t L, int64_t n,
                                      uint32_t* __restrict__ tsorted) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = keys_sorted[i];
    tsorted[i] = r < L ? (uint32_t)order[r] : L;
}

// One segment step of the grouped kernel.  The distance from p to a linestring is the smaller of (1) the distance to its
// nearest VERTEX and (2) the distance to the nearest segment whose interior holds p's projection — the same minimum the
// per-segment clamps of geo-types' line_segment_distance produce (an end point is never nearer than its segment), without
// the clamp selects: vertices go through one v_min_f64, interiors through a fraction cross^2 / d2 compared by
// cross-multiplication (no division, no sqrt per segment; one divide + sqrt per row at the end).  The start-point terms are
// carried from the previous segment.  Explicit fma: this kernel's results only need the 1e-9 contract; the zero / non-zero
// outcome is decided separately by the exact re-walk below.
struct SegState {
    double ax, ay, qx, qy;  // vertex a, p - a
};
__device__ __forceinline__ double vmin_f64(double a, double b) {  // one instruction (fmin() adds a canonicalising v_max per operand)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Round 6: the step by instruction count (the kernel is bound by f64 issue: 0.59 of the issue rate, PMC).  What does not depend on the
// point — the reciprocal of the segment's squared length — is computed ONCE per staged window by the lane that copied the vertex and
// read back from LDS with it; the distance to a segment is the distance to the clamped projection a + clamp(q.d / d.d, 0, 1) d, whose
// end points ARE the vertices (no separate vertex minimum, no cross-multiplied fraction compare): 12 vector instructions a segment
// (dx dy | dot | t | ex ey | e.e | min | next q) where the vertex / fraction form above took 17 - 21.  The clamp is the multiply's own
// output modifier.  A zero-length segment stores 0 for its reciprocal: t = 0, the distance to its vertex (geo-types' a == b arm).
__device__ __forceinline__ double mul_clamp01(double a, double b) {
    double r;
    asm("v_mul_f64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void seg_step(SegState& st, double2 b, double inv_d2, double px, double py, double& best) {
    const double dx = b.x - st.ax, dy = b.y - st.ay;
    const double dot = __builtin_fma(st.qx, dx, st.qy * dy);
    const double t = mul_clamp01(dot, inv_d2);
    const double ex = __builtin_fma(-t, dx, st.qx), ey = __builtin_fma(-t, dy, st.qy);
    best = vmin_f64(best, __builtin_fma(ex, ex, ey * ey));
    st.ax = b.x;
    st.ay = b.y;
    st.qx = px - b.x;
    st.qy = py - b.y;
}

constexpr int DC_K = 31;      // segments per staged window of one linestring (32 vertices)
constexpr int DC_SLOT = 33;   // LDS stride of a staged window in vertices (odd: windows of different targets sit on different banks)
constexpr int DC_MAXD = 8;    // targets staged at once per wave (a chunk with more distinct targets goes in several groups)
#ifndef GPK_DIST_ABLATE
#define GPK_DIST_ABLATE 0  // tuning builds only (answers wrong on purpose): 1 = points read and results written in map order (no gather, no scatter)
#endif
#ifndef GPK_DIST_ROWS
#define GPK_DIST_ROWS 1  // (2: measured 3 % slower on C3 — 122 registers halve the occupancy)
#endif
constexpr int DC_R = GPK_DIST_ROWS;       // rows per lane: a wave's chunk is 128 ordered rows (lane l holds rows l and 64 + l of it)
// what does not depend on the point, once per linestring column (kept with the row map): the reciprocal of every segment's squared
// length (0 for a zero-length one, and for the last vertex of the array) and every linestring's longest squared segment
__global__ __launch_bounds__(256) void seg_inv_kernel(const double2* __restrict__ xy, int64_t n_coords, double* __restrict__ inv) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_coords) return;
    double r = 0.0;
    if (j + 1 < n_coords) {
        const double2 a = xy[j], b = xy[j + 1];
        const double dx = b.x - a.x, dy = b.y - a.y, d2 = __builtin_fma(dx, dx, dy * dy);
        r = d2 > 0.0 ? 1.0 / d2 : 0.0;
    }
    inv[j] = r;
}
__global__ __launch_bounds__(256) void seg_max_kernel(DevGeo ls, const double* __restrict__ inv, double* __restrict__ max_d2) {
    // 8 lanes per linestring: the smallest non-zero reciprocal is the longest segment
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    double lo = INFINITY;
    if (t < ls.n_geoms) {
        const int c0 = ls.geom_off[t], c1 = ls.geom_off[t + 1];
        for (int j = c0 + sub; j + 1 < c1; j += 8) {
            const double r = inv[j];
            if (r > 0.0 && r < lo) lo = r;
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const double w = __shfl_xor(lo, o, 8);
        lo = w < lo ? w : lo;
    }
    if (t < ls.n_geoms && sub == 0) max_d2[t] = lo == INFINITY ? 0.0 : 1.0 / lo;
}
__global__ __launch_bounds__(256) void distance_grouped_kernel(DevGeo pts, DevGeo ls, const uint32_t* __restrict__ perm,
                                                               const uint32_t* __restrict__ tsorted, const double* __restrict__ seg_inv,
                                                               const double* __restrict__ seg_max, int64_t n_valid, double* __restrict__ out) {
    __shared__ double2 s_xy[4][DC_MAXD * DC_SLOT];
    __shared__ double s_inv[4][DC_MAXD * DC_SLOT];  // 1 / |v[j + 1] - v[j]|^2 of a staged window's segment j
    __shared__ int s_run[4][DC_MAXD][2];            // the runs of the current group: first coordinate, vertex count
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double2* sv = s_xy[wave];
    double* sinv = s_inv[wave];
    int(*srun)[2] = s_run[wave];
    constexpr int CHUNK = 64 * DC_R;
    const int64_t n_chunks = (n_valid + CHUNK - 1) / CHUNK;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + wave; ch < n_chunks; ch += (int64_t)gridDim.x * 4) {
        bool active[DC_R];
        uint32_t t[DC_R], i[DC_R];
        double2 p[DC_R];
        int c0[DC_R], nv[DC_R], my_run[DC_R];
        double best[DC_R], mx[DC_R];
        SegState st[DC_R];
        unsigned long long heads[DC_R];
        uint32_t t_last = 0xFFFFFFFFu;  // the target of the row before (lane 63 of the row set before)
        int runs_before = 0;
#pragma unroll
        for (int r = 0; r < DC_R; ++r) {
            const int64_t pos = ch * CHUNK + r * 64 + lane;
            active[r] = pos < n_valid;
            t[r] = active[r] ? tsorted[pos] : 0xFFFFFFFFu;
            i[r] = active[r] ? perm[pos] : 0u;
            p[r] = make_double2(NAN, NAN);
            if (active[r] && dev::valid_row(pts.validity, i[r])) p[r] = pts.xy[GPK_DIST_ABLATE ? (uint32_t)pos : i[r]];
            c0[r] = nv[r] = 0;
            mx[r] = 0.0;
            if (active[r]) {
                c0[r] = ls.geom_off[t[r]];
                nv[r] = ls.geom_off[t[r] + 1] - c0[r];
                mx[r] = seg_max[t[r]];
            }
            // runs of equal targets over the 128 ordered rows: the k-th run of the chunk is staged in LDS slot k (mod DC_MAXD)
            uint32_t t_prev = __shfl_up(t[r], 1, 64);
            if (lane == 0) t_prev = t_last;
            heads[r] = __ballot(active[r] && ((r == 0 && lane == 0) || t[r] != t_prev));
            my_run[r] = runs_before + (int)__popcll(heads[r] & ((2ull << lane) - 1ull)) - 1;
            runs_before += (int)__popcll(heads[r]);
            t_last = __shfl(t[r], 63, 64);
            best[r] = INFINITY;
            st[r] = SegState{0, 0, 0, 0};
        }
        const int n_runs = runs_before;
        for (int g0 = 0; g0 < n_runs; g0 += DC_MAXD) {
            // the group's runs: every run's first row says where its linestring starts and how long it is
            __builtin_amdgcn_wave_barrier();
            int nv_max = 0;
#pragma unroll
            for (int r = 0; r < DC_R; ++r) {
                const bool head = ((heads[r] >> lane) & 1ull) != 0ull;
                if (head && my_run[r] >= g0 && my_run[r] < g0 + DC_MAXD) {
                    srun[my_run[r] - g0][0] = c0[r];
                    srun[my_run[r] - g0][1] = nv[r];
                }
                const bool mine = active[r] && my_run[r] >= g0 && my_run[r] < g0 + DC_MAXD;
                nv_max = mine && nv[r] > nv_max ? nv[r] : nv_max;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int w = __shfl_xor(nv_max, o, 64);
                nv_max = w > nv_max ? w : nv_max;
            }
            const int g_runs = n_runs - g0 < DC_MAXD ? n_runs - g0 : DC_MAXD;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int w0 = 0; w0 < (nv_max > 1 ? nv_max - 1 : 1); w0 += DC_K) {
                __builtin_amdgcn_wave_barrier();
                {   // lane j of each half wave copies vertex j of a window and its segment's reciprocal: two runs per round
                    const int half = lane >> 5, j = lane & 31;
                    for (int sl = 0; sl < g_runs; sl += 2) {
                        const int q = sl + half;
                        if (q < g_runs) {
                            const int s_c0 = srun[q][0], s_nv = srun[q][1];
                            if (w0 + j < s_nv) {
                                sv[q * DC_SLOT + j] = ls.xy[s_c0 + w0 + j];
                                sinv[q * DC_SLOT + j] = seg_inv[s_c0 + w0 + j];
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < DC_R; ++r) {
                    const bool mine = active[r] && my_run[r] >= g0 && my_run[r] < g0 + DC_MAXD;
                    if (mine && nv[r] > 0 && w0 < (nv[r] > 1 ? nv[r] - 1 : 1)) {
                        const int slot = (my_run[r] - g0) * DC_SLOT;
                        if (w0 == 0) {
                            const double2 a = sv[slot];
                            st[r].ax = a.x;
                            st[r].ay = a.y;
                            st[r].qx = p[r].x - a.x;
                            st[r].qy = p[r].y - a.y;
                            // a one-vertex linestring is at f64::MAX unless the point IS the vertex (upstream's fold); with segments, their
                            // end points cover the vertices
                            if (nv[r] == 1 && st[r].qx == 0.0 && st[r].qy == 0.0) best[r] = 0.0;
                        }
                        const int kmax = nv[r] - 1 - w0 < DC_K ? nv[r] - 1 - w0 : DC_K;
                        // two segments per trip, the next trip's first vertex requested before this trip's arithmetic; the reads past the
                        // last vertex stay inside the window's slot (DC_SLOT = DC_K + 2 entries)
                        int k = 1;
                        double2 b0 = sv[slot + 1];
                        double i0 = sinv[slot];
                        for (; k + 1 <= kmax; k += 2) {
                            const double2 b1 = sv[slot + k + 1];
                            const double i1 = sinv[slot + k];
                            const double2 n0 = sv[slot + k + 2];
                            const double in0 = sinv[slot + k + 1];
                            seg_step(st[r], b0, i0, p[r].x, p[r].y, best[r]);
                            seg_step(st[r], b1, i1, p[r].x, p[r].y, best[r]);
                            b0 = n0;
                            i0 = in0;
                        }
                        if (k <= kmax) seg_step(st[r], b0, i0, p[r].x, p[r].y, best[r]);
                    }
                }
            }
        }
        // Upstream short-circuits "the point is on the linestring => 0" with |tx - ty| <= f64::EPSILON per segment
        // (tx - ty == cross / (dx dy)), which only a point within ~6 eps * |segment| of a segment can satisfy: squared,
        // best <= 64 eps^2 * (longest segment)^2.  Such lanes — and exact vertex hits, best == 0 — replay the upstream test
        // verbatim from global memory, so the zero / non-zero outcome stays exact; everybody else skips it.
#pragma unroll
        for (int r = 0; r < DC_R; ++r) {
            int eps_hit = 0;
            const bool maybe = active[r] && nv[r] > 0 && best[r] <= 3.1554436208840472e-30 * mx[r];  // 64 * 2^-104
            if (maybe) {
                if (nv[r] == 1) eps_hit = 1;  // best == 0 above: the point equals the vertex
                for (int k = 0; k + 1 < nv[r] && !eps_hit; ++k) {
                    const double2 a = ls.xy[c0[r] + k], b = ls.xy[c0[r] + k + 1];
                    const double cr = (p[r].x - a.x) * (b.y - a.y) - (p[r].y - a.y) * (b.x - a.x);
                    eps_hit = (int)((a.x == p[r].x && a.y == p[r].y) || (b.x == p[r].x && b.y == p[r].y) ||
                                    segment_contains_eps(p[r].x, p[r].y, a.x, a.y, b.x, b.y, cr, (b.x - a.x) * (b.y - a.y)));
                }
            }
            if (active[r]) {
                double d;
                if (!dev::valid_row(ls.validity, t[r]) || isnan(p[r].x) || isnan(p[r].y))
                    d = NAN;
                else if (nv[r] == 0 || eps_hit)
                    d = 0.0;
                else
                    d = best[r] == INFINITY ? DBL_MAX : sqrt(best[r]);
                out[GPK_DIST_ABLATE ? (uint32_t)(ch * CHUNK + r * 64 + lane) : i[r]] = d;
            }
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


// ---- row map: the rows of a point column ordered for the grouped distance kernel ---------------------------------------
}  // namespace gpk
struct gpk_rowmap {
    int device;
    int64_t n, n_valid, n_targets;
    uint32_t* perm;     // left rows in (target length order, target) order; rows without a target last
    uint32_t* tsorted;  // their targets, same order
    double* seg_inv;    // per coordinate of the linestring column: 1 / squared length of the segment that starts there (seg_inv_kernel)
    double* seg_max;    // per linestring: its longest squared segment
    int64_t nbytes;
};
namespace gpk {

// rows_dev: the map in device memory.  Synchronous (the build reads two counts back); buffers live until gpk_rowmap_free.
static int32_t rowmap_build_dev(const gpk_geoarray* ls, const uint32_t* rows_dev, int64_t n, hipStream_t s, gpk_rowmap** out) {
    *out = nullptr;
    const int64_t L = ls->d.n_geoms;
    if (n > (int64_t)INT32_MAX) return fail(GPK_ERR_INVALID_ARGUMENT, "row map: more than 2^31 - 1 rows (split the column)");
    gpk_rowmap* m = new gpk_rowmap;
    memset(m, 0, sizeof *m);
    m->n = n;
    m->n_targets = L;
    (void)hipGetDevice(&m->device);
    void* tmp = nullptr;
    void* sort_tmp = nullptr;
    auto fin = [&](int32_t rc) {
        (void)hipStreamSynchronize(s);
        // (the map's buffers and the build's scratch come from the library's block cache: four hipMalloc + two hipFree were
        // half of a 0.8 ms build on this runtime)
        if (tmp) cached_free(tmp);
        if (sort_tmp) cached_free(sort_tmp);
        if (rc != GPK_OK) {
            gpk_rowmap_free(m);
            m = nullptr;
        }
        *out = m;
        return rc;
    };
    const size_t nb = align256(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    {
        const hipError_t e1 = cached_malloc((void**)&m->perm, nb), e2 = cached_malloc((void**)&m->tsorted, nb);
        if (e1 != hipSuccess || e2 != hipSuccess) return fin(fail(GPK_ERR_OOM, "row map: hipMalloc failed"));
    }
    m->nbytes = (int64_t)(2 * nb);
    {   // what the distance kernel reads per segment and per linestring besides the coordinates
        const int64_t nc = ls->d.n_coords;
        const size_t ib8 = align256(sizeof(double) * (size_t)(nc > 0 ? nc : 1)), mb8 = align256(sizeof(double) * (size_t)(L > 0 ? L : 1));
        if (cached_malloc((void**)&m->seg_inv, ib8) != hipSuccess || cached_malloc((void**)&m->seg_max, mb8) != hipSuccess)
            return fin(fail(GPK_ERR_OOM, "row map: hipMalloc failed"));
        m->nbytes += (int64_t)(ib8 + mb8);
        auto pre = [&]() -> int32_t {
            if (nc > 0) GPK_LAUNCH("gpk_rowmap_seg_inv", seg_inv_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, ls->d.xy, nc, m->seg_inv);
            if (L > 0) GPK_LAUNCH("gpk_rowmap_seg_max", seg_max_kernel, dim3((unsigned)((L * 8 + 255) / 256)), dim3(256), 0, s, ls->d, (const double*)m->seg_inv, m->seg_max);
            return GPK_OK;
        };
        const int32_t prc = pre();
        if (prc != GPK_OK) return fin(prc);
    }
    if (n == 0) return fin(GPK_OK);  // an empty batch: an empty map (n_valid = 0), no launch (a grid of zero blocks is an error)
    // scratch: cnt | cnt_sorted | off | cursor (L+2 i32 each) | rank (L u32) | keys, sorted (L u64 each) | scan totals
    const size_t ib = align256(sizeof(int32_t) * (size_t)(L + 2)), kb = align256(sizeof(unsigned long long) * (size_t)(L + 1));
    size_t sort_bytes = 0;
    {
        const hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)L, 0, 64, s);
        if (e != hipSuccess) return fin(fail(GPK_ERR_DEVICE, "row map: %s", hipGetErrorString(e)));
    }
    const size_t total = 5 * ib + 2 * kb + align256(sort_bytes + 256) + align256(sizeof(unsigned long long) * (size_t)((L + 256) / 256 + 4));
    if (cached_malloc(&tmp, total) != hipSuccess) return fin(fail(GPK_ERR_OOM, "row map: hipMalloc(%zu) failed", total));
    char* base = (char*)tmp;
    int32_t* cnt = (int32_t*)base;
    int32_t* cnt_sorted = (int32_t*)(base + ib);
    int32_t* off = (int32_t*)(base + 2 * ib);
    int32_t* cursor = (int32_t*)(base + 3 * ib);
    uint32_t* rank = (uint32_t*)(base + 4 * ib);
    unsigned long long* keys = (unsigned long long*)(base + 5 * ib);
    unsigned long long* sorted = (unsigned long long*)(base + 5 * ib + kb);
    void* rp_tmp = base + 5 * ib + 2 * kb;
    unsigned long long* btot = (unsigned long long*)(base + 5 * ib + 2 * kb + align256(sort_bytes + 256));
    auto run = [&]() -> int32_t {
        const dim3 rg((unsigned)((n + 255) / 256)), lg((unsigned)((L + 256) / 256));
        // targets in descending vertex-count order
        GPK_LAUNCH("gpk_rowmap_keys", rowmap_keys_kernel, lg, dim3(256), 0, s, ls->d, (uint32_t)L, keys);
        GPK_HIP(rocprim::radix_sort_keys(rp_tmp, sort_bytes, (const unsigned long long*)keys, sorted, (size_t)L, 0, 64, s));
        // how local is the map?  (decides counting sort vs radix sort of the rows)
        unsigned long long h_local = 0, *d_local = btot;
        GPK_HIP(hipMemsetAsync(d_local, 0, sizeof(unsigned long long), s));
        const int probe_blocks = n >= 256 * DIST_PROBE_BLOCKS ? DIST_PROBE_BLOCKS : 1;
        GPK_LAUNCH("gpk_dist_probe", dist_probe_kernel, dim3((unsigned)probe_blocks), dim3(256), 0, s, rows_dev, n, d_local);
        GPK_HIP(hipMemcpyAsync(&h_local, d_local, sizeof h_local, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(L + 2), s));
        GPK_LAUNCH("gpk_rowmap_hist", rowmap_sort_kernel<false>, rg, dim3(256), 0, s, rows_dev, n, (uint32_t)L, (const uint32_t*)nullptr, cnt, (uint32_t*)nullptr,
                   (uint32_t*)nullptr);
        GPK_LAUNCH("gpk_rowmap_rank", rowmap_rank_kernel, lg, dim3(256), 0, s, (const unsigned long long*)sorted, (uint32_t)L, (const int32_t*)cnt, rank, cnt_sorted);
        GPK_TRY(exclusive_scan_i32(cnt_sorted, L + 1, off, cursor, btot, s));  // off[L] = rows with a target = start of the rest
        int32_t n_valid = 0;
        GPK_HIP(hipMemcpyAsync(&n_valid, off + L, sizeof n_valid, hipMemcpyDeviceToHost, s));
        GPK_HIP(hipStreamSynchronize(s));
        m->n_valid = n_valid;
        const int64_t sampled = (int64_t)probe_blocks * 256 < n ? (int64_t)probe_blocks * 256 : n;
        if (2 * (int64_t)h_local >= sampled) {
            GPK_LAUNCH("gpk_rowmap_scatter", rowmap_sort_kernel<true>, rg, dim3(256), 0, s, rows_dev, n, (uint32_t)L, (const uint32_t*)rank, cursor, m->perm,
                       m->tsorted);
        } else {  // scattered map: every lane's atomic would be its own cache line; a radix sort of (length rank, row) pairs wins
            int bits = 1;
            while (bits < 32 && (1ll << bits) < L + 1) ++bits;  // keys 0..L (L = "no such target")
            size_t tb = 0;
            GPK_HIP(rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, m->perm, (size_t)n, 0, bits, s));
            GPK_HIP(cached_malloc(&sort_tmp, 3 * nb + tb + 256));
            uint32_t* iota = (uint32_t*)sort_tmp;
            uint32_t* keys_in = (uint32_t*)((char*)sort_tmp + nb);
            uint32_t* keys_sorted = (uint32_t*)((char*)sort_tmp + 2 * nb);
            void* rp2 = (char*)sort_tmp + 3 * nb;
            GPK_LAUNCH("gpk_rowmap_radix_keys", rowmap_radix_keys_kernel, rg, dim3(256), 0, s, rows_dev, (uint32_t)L, (const uint32_t*)rank, keys_in, iota, n);
            GPK_HIP(rocprim::radix_sort_pairs(rp2, tb, (const uint32_t*)keys_in, keys_sorted, (const uint32_t*)iota, m->perm, (size_t)n, 0, bits, s));
            GPK_LAUNCH("gpk_rowmap_targets", rowmap_targets_kernel, rg, dim3(256), 0, s, (const uint32_t*)keys_sorted, (const unsigned long long*)sorted, (uint32_t)L, n,
                       m->tsorted);
        }
        return GPK_OK;
    };
    return fin(run());
}

static int32_t distance_rowmap_dev(const gpk_geoarray* pts, const gpk_geoarray* ls, const gpk_rowmap* map, double* out_dev, hipStream_t s) {
    const int64_t n = map->n, nv = map->n_valid;
    if (nv < n)  // map entries without a target: the answer of a null row
        GPK_LAUNCH("gpk_rowmap_nan", rowmap_nan_kernel, dim3((unsigned)((n - nv + 255) / 256)), dim3(256), 0, s, (const uint32_t*)map->perm, nv, n, out_dev);
    if (nv == 0) return GPK_OK;
    int64_t blocks = ((nv + 64 * DC_R - 1) / (64 * DC_R) + 3) / 4;  // one wave per chunk of 128 ordered rows
    const int64_t cap = (int64_t)cu_count() * 16;
    if (blocks > cap) blocks = cap;
    GPK_LAUNCH("gpk_distance_grouped", distance_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pts->d, ls->d, (const uint32_t*)map->perm,
               (const uint32_t*)map->tsorted, (const double*)map->seg_inv, (const double*)map->seg_max, nv, out_dev);
    return GPK_OK;
}

}  // namespace gpk

using namespace gpk;

extern "C" {

int32_t gpk_rowmap_free(gpk_rowmap* m) {
    if (!m) return GPK_OK;
    if (m->perm || m->tsorted || m->seg_inv) {  // (cached blocks: wait once, on the owning device, for whatever still reads the map)
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != m->device) (void)hipSetDevice(m->device);
        (void)hipDeviceSynchronize();
        if (cur >= 0 && cur != m->device) (void)hipSetDevice(cur);
    }
    if (m->perm) cached_free(m->perm);
    if (m->tsorted) cached_free(m->tsorted);
    if (m->seg_inv) cached_free(m->seg_inv);
    if (m->seg_max) cached_free(m->seg_max);
    delete m;
    return GPK_OK;
}

int32_t gpk_rowmap_build(const gpk_geoarray* b, const uint32_t* b_rows, int64_t n_rows, int32_t rows_space, void* stream, gpk_rowmap** out) {
    if (!b || !out || (!b_rows && n_rows > 0) || n_rows < 0) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    GPK_TRY(require_device());
    if (b->d.type != GPK_GEOM_LINESTRING)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "row map: the grouped schedule serves LINESTRING right sides (found type %d)", b->d.type);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t* rows_dev = b_rows;
    void* staged = nullptr;
    if (rows_space != GPK_MEM_DEVICE && n_rows > 0) {
        GPK_HIP(device_malloc(&staged, sizeof(uint32_t) * (size_t)n_rows));
        const hipError_t e = hipMemcpyAsync(staged, b_rows, sizeof(uint32_t) * (size_t)n_rows, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) {
            (void)hipFree(staged);
            return fail(GPK_ERR_DEVICE, "row map: %s", hipGetErrorString(e));
        }
        rows_dev = (const uint32_t*)staged;
    }
    const int32_t rc = rowmap_build_dev(b, rows_dev, n_rows, s, out);  // synchronises before it returns
    if (staged) (void)hipFree(staged);
    return rc;
}

int32_t gpk_rowmap_nbytes(const gpk_rowmap* m, int64_t* out_bytes) {
    if (!m || !out_bytes) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_bytes = m->nbytes;
    return GPK_OK;
}

int32_t gpk_distance_rowmap(const gpk_geoarray* a, const gpk_geoarray* b, const gpk_rowmap* map, double* out, int32_t out_space, void* stream) {
    if (!a || !b || !map || !out) return fail(GPK_ERR_INVALID_ARGUMENT, "NULL argument");
    GPK_TRY(require_device());
    if (a->d.type != GPK_GEOM_POINT || b->d.type != GPK_GEOM_LINESTRING)
        return fail(GPK_ERR_MISMATCHED_GEOMETRY, "distance with a row map: POINT x LINESTRING (found types %d, %d)", a->d.type, b->d.type);
    if (map->n != a->d.n_geoms || map->n_targets != b->d.n_geoms)
        return fail(GPK_ERR_INVALID_ARGUMENT, "row map was built for %lld rows x %lld targets, called with %lld x %lld", (long long)map->n,
                    (long long)map->n_targets, (long long)a->d.n_geoms, (long long)b->d.n_geoms);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = a->d.n_geoms;
    if (n == 0) return GPK_OK;
    const size_t ob = sizeof(double) * (size_t)n;
    double* out_dev = out;
    if (out_space != GPK_MEM_DEVICE) {
        GPK_TRY(workspace().begin(align256(ob) + 512));
        out_dev = (double*)workspace().take(ob);
    }
    GPK_TRY(distance_rowmap_dev(a, b, map, out_dev, s));
    return copy_out(out, out_space, out_dev, ob, s);
}

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
    // many rows per linestring: order the rows by target once (the row map), stage each linestring in LDS per chunk
    if (b_rows && other->d.type == GPK_GEOM_LINESTRING && other->d.n_geoms > 0 && n >= 8 * other->d.n_geoms) {
        gpk_rowmap* map = nullptr;
        GPK_TRY(rowmap_build_dev(other, rows_dev, n, s, &map));
        int32_t rc = distance_rowmap_dev(pts, other, map, out_dev, s);
        if (rc == GPK_OK) rc = copy_out(out, out_space, out_dev, ob, s);
        (void)hipStreamSynchronize(s);  // the map's buffers are released below
        gpk_rowmap_free(map);
        return rc;
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
