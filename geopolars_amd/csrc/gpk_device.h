// gpk_device.h — device-side geometry predicates for gfx950.
//
// Semantics are those of the crates the reference operator surface delegates to (geo 0.27 /
// robust 1.1, Cargo.lock:986-1004,2251; call sites geopolars/src/spatial_index.rs:89-137).  All
// arithmetic is IEEE f64 with contraction OFF (the library is built -ffp-contract=off); the only
// fused operations are the explicit fma() of the error-free products.  Boolean results are exact:
// every orientation sign is either certified by Shewchuk's stage-A bound or recomputed with
// expansion arithmetic on the input coordinates.
#pragma once

#include <hip/hip_runtime.h>

#include "gpk_common.h"

namespace gpk {
namespace dev {

constexpr int POS_OUTSIDE = 0;
constexpr int POS_BOUNDARY = 1;
constexpr int POS_INSIDE = 2;

// Streaming accesses (read-once inputs, write-once outputs) carry the non-temporal hint so that they do not
// evict the small gather tables (raster, slabs, directory) from the XCD's 4 MB L2.
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_stream(const double2* p) {
    const v2f64 v = __builtin_nontemporal_load(reinterpret_cast<const v2f64*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void store_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ bool valid_row(const uint8_t* validity, int64_t i) {
    return !validity || ((validity[i >> 3] >> (i & 7)) & 1);
}
// a row a caller's index map points at: out-of-range indices behave like null rows (the gather semantics of
// gpk_take_*; include/geopolars_hip.h states the contract for `b_rows`)
__device__ __forceinline__ bool row_ok(const DevGeo& g, int64_t j) {
    return (uint64_t)j < (uint64_t)g.n_geoms && valid_row(g.validity, j);
}

// ---- exact orientation -------------------------------------------------------------------------
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    const double bv = s - a;
    const double av = s - bv;
    e = (a - av) + (b - bv);
}
__device__ __forceinline__ void two_prod(double a, double b, double& p, double& e) {
    p = a * b;
    e = __builtin_fma(a, b, -p);
}

// Exact sign of ax*by - ax*cy - cx*by - ay*bx + ay*cx + cy*bx (== orient2d determinant of the INPUT
// coordinates; the cx*cy terms cancel).  Six error-free products -> 12 components, summed exactly by
// repeated grow-expansion; the sign of a non-overlapping expansion is the sign of its most
// significant non-zero component.  Rare path (points within ~1e-16 relative of an edge's line): kept
// out of line and rolled so it costs the hot kernels no registers.
__device__ __noinline__ int orient2d_exact(double ax, double ay, double bx, double by, double cx,
                                           double cy) {
    double t[12];
    two_prod(ax, by, t[0], t[1]);
    two_prod(-ax, cy, t[2], t[3]);
    two_prod(-cx, by, t[4], t[5]);
    two_prod(-ay, bx, t[6], t[7]);
    two_prod(ay, cx, t[8], t[9]);
    two_prod(cy, bx, t[10], t[11]);
    double e[12];
    int n = 0;
#pragma unroll 1
    for (int i = 0; i < 12; ++i) {
        double q = t[i];
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            double s, err;
            two_sum(q, e[j], s, err);
            e[j] = err;
            q = s;
        }
        e[n++] = q;
    }
    int sign = 0;
#pragma unroll 1
    for (int i = 0; i < 12; ++i) {  // most significant non-zero component is the last one
        if (e[i] > 0.0) sign = 1;
        if (e[i] < 0.0) sign = -1;
    }
    return sign;
}

// robust::orient2d sign: +1 counter-clockwise, -1 clockwise, 0 collinear.
__device__ __forceinline__ int orient2d(double ax, double ay, double bx, double by, double cx,
                                        double cy) {
    const double detleft = (ax - cx) * (by - cy);
    const double detright = (ay - cy) * (bx - cx);
    const double det = detleft - detright;
    // Stage A.  When detleft and detright have opposite signs (or one is zero) det is sign-exact; the
    // bound below is then trivially satisfied or det == 0 with both zero, so one test covers all arms
    // except the exact-zero ones handled by `certain`.
    const double detsum = fabs(detleft) + fabs(detright);
    const double errbound = 3.3306690738754716e-16 * detsum;  // (3 + 16 eps) eps, eps = 2^-53
    const bool opposite = (detleft > 0.0 && detright <= 0.0) || (detleft < 0.0 && detright >= 0.0) ||
                          detleft == 0.0;
    const bool certain = opposite || fabs(det) >= errbound;
    if (__builtin_expect(certain, 1)) return (det > 0.0) - (det < 0.0);
    return orient2d_exact(ax, ay, bx, by, cx, cy);
}

__device__ __forceinline__ bool value_in_between(double v, double a, double b) {
    return a > b ? (v >= b && v <= a) : (v >= a && v <= b);
}

// One edge of coord_pos_relative_to_ring (geo 0.27 coordinate_position.rs): updates the winding
// number, returns true when the coordinate lies ON the edge.
//
// Upstream evaluates orient2d for every edge whose y-range straddles c.y.  Here the determinant is only
// evaluated when c.x also lies within the edge's closed x-range: outside it the sign is known (strictly
// left of both endpoints => left of the directed edge; strictly right => right of it) and the
// collinear / on-boundary arm cannot fire, so the result is identical.  The upward and downward arms share
// ONE orientation evaluation (they are mutually exclusive), which matters on a 64-wide wave where every
// divergent arm is paid by all lanes.
__device__ __forceinline__ bool ring_edge(double sx, double sy, double ex, double ey, double cx,
                                          double cy, int& wn) {
    const bool up = sy <= cy && ey >= cy;   // upward (or horizontal at c.y): includes start, excludes end
    const bool down = sy > cy && ey <= cy;  // downward: excludes start, includes end
    if (!(up || down)) return false;
    const double lo = fmin(sx, ex), hi = fmax(sx, ex);
    if (cx < lo) {  // strictly left of the whole edge: it is crossed unless it ends level with c (upward arm)
        wn += up ? (ey != cy ? 1 : 0) : -1;
        return false;
    }
    if (!(cx <= hi)) return false;  // strictly right (or NaN): never counted, cannot be on it
    const int o = orient2d(sx, sy, ex, ey, cx, cy);
    if (o == 0) return true;  // collinear and within the closed x-range: on the boundary
    if (up)
        wn += (o > 0 && ey != cy) ? 1 : 0;
    else
        wn -= (o < 0) ? 1 : 0;
    return false;
}

// orient2d_exact without a call and without scratch memory (every loop unrolled: the two expansions live in registers), for
// code that must not contain a call (the tile kernels of gpk_join.hip settle their rare rows in place)
__device__ __forceinline__ int orient2d_exact_unrolled(double ax, double ay, double bx, double by, double cx, double cy) {
    double t[12];
    two_prod(ax, by, t[0], t[1]);
    two_prod(-ax, cy, t[2], t[3]);
    two_prod(-cx, by, t[4], t[5]);
    two_prod(-ay, bx, t[6], t[7]);
    two_prod(ay, cx, t[8], t[9]);
    two_prod(cy, bx, t[10], t[11]);
    double e[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        double q = t[i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            double s, err;
            two_sum(q, e[j], s, err);
            e[j] = err;
            q = s;
        }
        e[i] = q;
    }
    int sign = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {  // most significant non-zero component is the last one
        if (e[i] > 0.0) sign = 1;
        if (e[i] < 0.0) sign = -1;
    }
    return sign;
}
// dev::ring_edge with the exact arm inlined (same arms, same results)
__device__ __forceinline__ bool ring_edge_inline(double sx, double sy, double ex, double ey, double cx, double cy, int& wn) {
    const bool up = sy <= cy && ey >= cy;
    const bool down = sy > cy && ey <= cy;
    if (!(up || down)) return false;
    const double lo = fmin(sx, ex), hi = fmax(sx, ex);
    if (cx < lo) {
        wn += up ? (ey != cy ? 1 : 0) : -1;
        return false;
    }
    if (!(cx <= hi)) return false;
    const double detleft = (sx - cx) * (ey - cy);
    const double detright = (sy - cy) * (ex - cx);
    const double det = detleft - detright;
    const double detsum = fabs(detleft) + fabs(detright);
    const double errbound = 3.3306690738754716e-16 * detsum;
    const bool opposite = (detleft > 0.0 && detright <= 0.0) || (detleft < 0.0 && detright >= 0.0) || detleft == 0.0;
    int o;
    if (opposite || fabs(det) >= errbound)
        o = (det > 0.0) - (det < 0.0);
    else
        o = orient2d_exact_unrolled(sx, sy, ex, ey, cx, cy);
    if (o == 0) return true;
    if (up)
        wn += (o > 0 && ey != cy) ? 1 : 0;
    else
        wn -= (o < 0) ? 1 : 0;
    return false;
}

// ring_edge for kernels that keep the expansion arithmetic out of their code (pip_tile_chain, gpk_join.hip): the same arms, but an
// orientation that Shewchuk's stage-A bound cannot certify is not recomputed here — `unsure` is set and the caller hands the
// point to the exact walk.  (wn and the return value are then meaningless for this edge.)
__device__ __forceinline__ bool ring_edge_filtered(double sx, double sy, double ex, double ey, double cx, double cy, int& wn, bool& unsure) {
    const bool up = sy <= cy && ey >= cy;
    const bool down = sy > cy && ey <= cy;
    if (!(up || down)) return false;
    const double lo = fmin(sx, ex), hi = fmax(sx, ex);
    if (cx < lo) {
        wn += up ? (ey != cy ? 1 : 0) : -1;
        return false;
    }
    if (!(cx <= hi)) return false;
    const double detleft = (sx - cx) * (ey - cy);
    const double detright = (sy - cy) * (ex - cx);
    const double det = detleft - detright;
    const double detsum = fabs(detleft) + fabs(detright);
    const double errbound = 3.3306690738754716e-16 * detsum;
    const bool opposite = (detleft > 0.0 && detright <= 0.0) || (detleft < 0.0 && detright >= 0.0) || detleft == 0.0;
    if (!(opposite || fabs(det) >= errbound)) {
        unsure = true;
        return false;
    }
    const int o = (det > 0.0) - (det < 0.0);
    if (o == 0) return true;
    if (up)
        wn += (o > 0 && ey != cy) ? 1 : 0;
    else
        wn -= (o < 0) ? 1 : 0;
    return false;
}

// coord_pos_relative_to_ring over a closed ring in global memory.
__device__ inline int coord_pos_ring(const double2* __restrict__ ring, int n, double cx, double cy) {
    if (n == 0) return POS_OUTSIDE;
    double2 s = ring[0];
    if (n == 1) return (cx == s.x && cy == s.y) ? POS_BOUNDARY : POS_OUTSIDE;
    int wn = 0;
    bool on = false;
    for (int i = 1; i < n; ++i) {
        const double2 e = ring[i];
        on |= ring_edge(s.x, s.y, e.x, e.y, cx, cy, wn);
        s = e;
    }
    if (on) return POS_BOUNDARY;
    return wn == 0 ? POS_OUTSIDE : POS_INSIDE;
}

// ---- normalised geometry accessors ---------------------------------------------------------
__device__ __forceinline__ void geom_parts(const DevGeo& a, int64_t g, int& p0, int& p1) {
    if (a.type == GPK_GEOM_MULTIPOLYGON) {
        p0 = a.geom_off[g];
        p1 = a.geom_off[g + 1];
    } else {
        p0 = (int)g;
        p1 = (int)g + 1;
    }
}
__device__ __forceinline__ void part_rings(const DevGeo& a, int p, int& r0, int& r1) {
    const int32_t* off = a.type == GPK_GEOM_MULTIPOLYGON ? a.part_off : a.geom_off;
    r0 = off[p];
    r1 = off[p + 1];
}

// Polygon::coordinate_position
__device__ inline int polygon_pos(const DevGeo& a, int r0, int r1, double cx, double cy) {
    if (r1 <= r0) return POS_OUTSIDE;
    int c0 = a.ring_off[r0], c1 = a.ring_off[r0 + 1];
    if (c1 == c0) return POS_OUTSIDE;
    const int pe = coord_pos_ring(a.xy + c0, c1 - c0, cx, cy);
    if (pe != POS_INSIDE) return pe;
    for (int r = r0 + 1; r < r1; ++r) {
        c0 = a.ring_off[r];
        c1 = a.ring_off[r + 1];
        const int ph = coord_pos_ring(a.xy + c0, c1 - c0, cx, cy);
        if (ph == POS_BOUNDARY) return POS_BOUNDARY;
        if (ph == POS_INSIDE) return POS_OUTSIDE;
    }
    return POS_INSIDE;
}

// Contains<Point> (boundary excluded) / Intersects<Point> (boundary included) for a polygonal row:
// true when ANY member polygon satisfies it.
template <bool BOUNDARY_COUNTS>
__device__ inline bool polygonal_hits_point(const DevGeo& a, int64_t g, double cx, double cy) {
    int p0, p1;
    geom_parts(a, g, p0, p1);
    for (int p = p0; p < p1; ++p) {
        int r0, r1;
        part_rings(a, p, r0, r1);
        const int pos = polygon_pos(a, r0, r1, cx, cy);
        if (BOUNDARY_COUNTS ? pos != POS_OUTSIDE : pos == POS_INSIDE) return true;
    }
    return false;
}

// ---- all-reduce over G consecutive lanes with DPP row operations --------------------------------
// G = 2, 4, 8, 16 (a group never straddles a 16-lane DPP row): quad permutes for xor 1 / xor 2, row_half_mirror to join
// the two quads of an 8-lane half row, row_mirror to join the two halves of a row.  One VALU mov per 32-bit
// word and step, instead of a ds_bpermute round trip through the LDS crossbar.  All lanes of the group must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
}
template <int G, typename T, typename Op>
__device__ __forceinline__ T group_allreduce(T v, Op op) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group_allreduce: G lanes within one DPP row");
    if constexpr (G >= 2) v = op(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v = op(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v = op(v, dpp_mov<0x141>(v));   // row_half_mirror: lane i <-> 7 - i of its half row
    if constexpr (G == 16) v = op(v, dpp_mov<0x140>(v));  // row_mirror: lane i <-> 15 - i
    return v;  // every step pairs two lanes symmetrically, so a commutative op leaves the same bits on all lanes
}
template <int G>
__device__ __forceinline__ int group_sum(int v) { return group_allreduce<G>(v, [](int a, int b) { return a + b; }); }
template <int G>
__device__ __forceinline__ double group_sum(double v) { return group_allreduce<G>(v, [](double a, double b) { return a + b; }); }
template <int G>
__device__ __forceinline__ int group_or(int v) { return group_allreduce<G>(v, [](int a, int b) { return a | b; }); }
template <int G>
__device__ __forceinline__ double group_min(double v) { return group_allreduce<G>(v, [](double a, double b) { return fmin(a, b); }); }
template <int G>
__device__ __forceinline__ double group_max(double v) { return group_allreduce<G>(v, [](double a, double b) { return fmax(a, b); }); }

// ---- wave64 helpers ------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dev
}  // namespace gpk
