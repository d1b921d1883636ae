// gpk_contains.h — polygon `contains` polygon, reached from the join dispatch geopolars/src/spatial_index.rs:99-101
// (Polygon x Polygon) and :107-111 (MultiPolygon x Polygon) and from the row-wise north-star predicates
// contains / within.  geo 0.27 evaluates it as `relate(..).is_contains()` (DE-9IM T*****FF*); for valid operands
// that is "B is not empty and B is a subset of A", decided here with exact orientations only:
//   (1) no vertex of B is Outside A and no piece of an edge of B leaves A.  An edge is cut into pieces by its
//       touch points with a ring; the side of the piece next to a touch point is read off the direction of the
//       edge there — against the sector of the two ring edges at a ring vertex, against the ring edge when an end
//       point of the edge lies inside it; a proper crossing has a piece on either side.  Leaving A = outside its
//       exterior or inside one of its holes, so the test runs ring by ring;
//   (2) no hole of A is swallowed by B: given (1) the hole's interior misses every ring of B, so it is inside or
//       outside each of them as a whole; swallowed = inside (or equal to) B's exterior and outside all B's holes.
// G lanes work on one (A, B) pair: the ring being tested against is strided over the lanes, the edge or vertex
// tested is the same on all of them, verdicts are group-wide ORs (every branch below is group-uniform, as the DPP
// reductions of gpk_device.h need).  Unclosed rings, rings with fewer than 4 coordinates or without a turning
// extreme vertex make the operand invalid: the answer is false (oracle/gpk_oracle.c takes the same decisions).
#pragma once

#include "gpk_device.h"
#include "gpk_polypoly.h"

namespace gpk {
namespace cont {

enum { DIR_IN = 1, DIR_OUT = 2 };  // bits: a piece strictly inside / strictly outside the ring
enum { REL_IN = 0, REL_OUT = 1, REL_SAME = 2 };

struct Ring {
    const double2* v;  // closed: v[m] == v[0]
    int m;             // number of edges
    int ccw;           // +1 counter-clockwise, -1 clockwise
};

__device__ __forceinline__ bool same_xy(double2 a, double2 b) { return a.x == b.x && a.y == b.y; }
__device__ __forceinline__ int orient(double2 a, double2 b, double2 c) { return dev::orient2d(a.x, a.y, b.x, b.y, c.x, c.y); }

__device__ inline int prev_distinct(const Ring& r, int i) {
    const double2 v = r.v[i];
    for (int k = 1; k < r.m; ++k) {
        const int j = (i - k + r.m) % r.m;
        if (!same_xy(r.v[j], v)) return j;
    }
    return -1;
}
__device__ inline int next_distinct(const Ring& r, int i) {
    const double2 v = r.v[i];
    for (int k = 1; k < r.m; ++k) {
        const int j = (i + k) % r.m;
        if (!same_xy(r.v[j], v)) return j;
    }
    return -1;
}

// orientation of a simple ring = the turn at its lexicographically smallest vertex (always a convex corner)
template <int G>
__device__ inline bool ring_init(Ring& r, const double2* xy, int n, int lane) {
    if (n < 4) return false;
    if (!same_xy(xy[0], xy[n - 1])) return false;
    r.v = xy;
    r.m = n - 1;
    double bx = INFINITY, by = INFINITY;
    int bi = 0x7fffffff, nan = 0;
    for (int i = lane; i < r.m; i += G) {
        const double2 p = xy[i];
        nan |= (p.x != p.x) | (p.y != p.y);
        if (p.x < bx || (p.x == bx && p.y < by)) {  // strict: the lane keeps its first minimum
            bx = p.x;
            by = p.y;
            bi = i;
        }
    }
    const double gx = dev::group_min<G>(bx);
    const double gy = dev::group_min<G>(bx == gx ? by : INFINITY);
    const int k = dev::group_allreduce<G>((bx == gx && by == gy) ? bi : 0x7fffffff, [](int a, int b) { return a < b ? a : b; });
    if (dev::group_or<G>(nan) || k == 0x7fffffff) return false;  // NaN coordinates
    const int p = prev_distinct(r, k), q = next_distinct(r, k);
    if (p < 0 || q < 0) return false;
    r.ccw = orient(xy[p], xy[k], xy[q]);
    return r.ccw != 0;
}

// w lies on the line through v and t (t != v, w != v): on the same side of v as t?
__device__ __forceinline__ bool same_ray(double2 v, double2 t, double2 w) {
    if (t.x != v.x) return (t.x > v.x) == (w.x > v.x);
    return (t.y > v.y) == (w.y > v.y);
}

// the piece of the segment vertex_i -> w next to vertex_i: strictly inside the ring (DIR_IN), strictly outside
// (DIR_OUT), or running along one of the two incident ring edges (0)
__device__ inline int dir_at_vertex(const Ring& r, int i, double2 w) {
    const double2 v = r.v[i];
    int ip = prev_distinct(r, i), iq = next_distinct(r, i);
    if (r.ccw < 0) {  // walk the ring with its inside on the left
        const int t = ip;
        ip = iq;
        iq = t;
    }
    const double2 p = r.v[ip], q = r.v[iq];
    const int o1 = orient(p, v, w), o2 = orient(v, q, w);
    if (o1 == 0 && same_ray(v, p, w)) return 0;
    if (o2 == 0 && same_ray(v, q, w)) return 0;
    const int turn = orient(p, v, q);
    bool in;
    if (turn > 0)
        in = o1 > 0 && o2 > 0;  // convex corner: between the two edges
    else if (turn < 0)
        in = o1 > 0 || o2 > 0;  // reflex corner
    else
        in = o1 > 0;  // straight through
    return in ? DIR_IN : DIR_OUT;
}

__device__ __forceinline__ bool strictly_between(double2 p, double2 a, double2 b) {
    return !same_xy(p, a) && !same_xy(p, b) && dev::value_in_between(p.x, a.x, b.x) && dev::value_in_between(p.y, a.y, b.y);
}

// DIR_IN / DIR_OUT bits of the pieces of segment pq next to its touch points with the ring edges first, first+step, ...
__device__ inline int edge_ring_flags(double2 p, double2 q, const Ring& r, int first, int step) {
    if (same_xy(p, q)) return 0;
    const double lx = fmin(p.x, q.x), hx = fmax(p.x, q.x), ly = fmin(p.y, q.y), hy = fmax(p.y, q.y);
    int fl = 0;
    for (int i = first; i < r.m; i += step) {
        const double2 a = r.v[i], b = r.v[i + 1];
        if (fmax(a.x, b.x) < lx || fmin(a.x, b.x) > hx || fmax(a.y, b.y) < ly || fmin(a.y, b.y) > hy) continue;
        const int oa = orient(p, q, a);
        if (oa == 0 && dev::value_in_between(a.x, p.x, q.x) && dev::value_in_between(a.y, p.y, q.y)) {
            if (!same_xy(a, q)) fl |= dir_at_vertex(r, i, q);
            if (!same_xy(a, p)) fl |= dir_at_vertex(r, i, p);
        }
        if (same_xy(a, b)) continue;
        const int ob = orient(p, q, b);
        const int op = orient(a, b, p) * r.ccw, oq = orient(a, b, q) * r.ccw;
        if (op == 0 && strictly_between(p, a, b)) fl |= oq > 0 ? DIR_IN : (oq < 0 ? DIR_OUT : 0);
        if (oq == 0 && strictly_between(q, a, b)) fl |= op > 0 ? DIR_IN : (op < 0 ? DIR_OUT : 0);
        if (oa * ob < 0 && op * oq < 0) fl |= DIR_IN | DIR_OUT;
    }
    return fl;
}

// where the interior of ring h lies relative to ring r, given that it does not meet r: the first vertex of h off r
// decides; when every vertex lies on r, the first edge with a piece off r; REL_SAME when h runs along r throughout
template <int G>
__device__ inline int ring_rel(const double2* h, int hm, const Ring& r, int lane) {
    for (int i = 0; i < hm; ++i) {
        const double2 c = h[i];
        const int pos = coord_pos_ring_group<G>(r.v, r.m + 1, c.x, c.y, lane);
        if (pos == dev::POS_INSIDE) return REL_IN;
        if (pos == dev::POS_OUTSIDE) return REL_OUT;
    }
    for (int i = 0; i < hm; ++i) {
        const int fl = dev::group_or<G>(edge_ring_flags(h[i], h[i + 1], r, lane, G));
        if (fl & DIR_IN) return REL_IN;
        if (fl & DIR_OUT) return REL_OUT;
    }
    return REL_SAME;
}

template <int G>
__device__ inline double4 ring_bbox(const double2* xy, int n, int lane) {
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = lane; i < n; i += G) {
        const double2 p = xy[i];
        mnx = fmin(mnx, p.x); mny = fmin(mny, p.y); mxx = fmax(mxx, p.x); mxy = fmax(mxy, p.y);
    }
    return make_double4(dev::group_min<G>(mnx), dev::group_min<G>(mny), dev::group_max<G>(mxx), dev::group_max<G>(mxy));
}

// every ring of the polygon is a usable closed ring (empty holes are skipped everywhere)
template <int G>
__device__ inline bool rings_valid(const DevGeo& a, int r0, int r1, int lane) {
    for (int r = r0; r < r1; ++r) {
        const int c0 = a.ring_off[r], n = a.ring_off[r + 1] - c0;
        if (n == 0 && r > r0) continue;
        Ring t;
        if (!ring_init<G>(t, a.xy + c0, n, lane)) return false;
    }
    return true;
}

// Contains<Polygon> for Polygon: rings [ar0, ar1) of a hold rings [br0, br1) of b.  Same value on every lane.
template <int G>
__device__ inline bool polygon_contains_polygon_group(const DevGeo& a, int ar0, int ar1, const DevGeo& b, int br0, int br1, int lane) {
    if (ar1 <= ar0 || br1 <= br0) return false;
    const int ae0 = a.ring_off[ar0], aen = a.ring_off[ar0 + 1] - ae0;
    const int be0 = b.ring_off[br0], ben = b.ring_off[br0 + 1] - be0;
    if (aen == 0 || ben == 0) return false;  // an empty polygon holds nothing and is held by nothing
    {
        const double4 ba = ring_bbox<G>(a.xy + ae0, aen, lane), bb = ring_bbox<G>(b.xy + be0, ben, lane);
        if (bb.x < ba.x || bb.y < ba.y || bb.z > ba.z || bb.w > ba.w) return false;
    }
    if (!rings_valid<G>(a, ar0, ar1, lane) || !rings_valid<G>(b, br0, br1, lane)) return false;
    // (1) vertices of b: none Outside a
    for (int rb = br0; rb < br1; ++rb) {
        const int c0 = b.ring_off[rb], m = b.ring_off[rb + 1] - c0 - 1;
        for (int i = 0; i < m; ++i) {
            const double2 c = b.xy[c0 + i];
            if (polygon_pos_group<G>(a, ar0, ar1, c.x, c.y, lane) == dev::POS_OUTSIDE) return false;
        }
    }
    // (1) edges of b: no piece outside a's exterior or inside one of a's holes
    for (int ra = ar0; ra < ar1; ++ra) {
        const int a0 = a.ring_off[ra], an = a.ring_off[ra + 1] - a0;
        if (an == 0) continue;
        Ring r;
        (void)ring_init<G>(r, a.xy + a0, an, lane);
        const int bad = ra == ar0 ? DIR_OUT : DIR_IN;
        for (int rb = br0; rb < br1; ++rb) {
            const int c0 = b.ring_off[rb], m = b.ring_off[rb + 1] - c0 - 1;
            int fl = 0;
            for (int i = 0; i < m; ++i) fl |= edge_ring_flags(b.xy[c0 + i], b.xy[c0 + i + 1], r, lane, G);
            if (dev::group_or<G>(fl) & bad) return false;
        }
    }
    // (2) holes of a: none swallowed by b
    if (ar1 - ar0 > 1) {
        Ring eb;
        (void)ring_init<G>(eb, b.xy + be0, ben, lane);
        for (int ra = ar0 + 1; ra < ar1; ++ra) {
            const int a0 = a.ring_off[ra], an = a.ring_off[ra + 1] - a0;
            if (an == 0) continue;
            if (ring_rel<G>(a.xy + a0, an - 1, eb, lane) == REL_OUT) continue;
            bool swallowed = true;
            for (int rb = br0 + 1; rb < br1 && swallowed; ++rb) {
                const int c0 = b.ring_off[rb], bn = b.ring_off[rb + 1] - c0;
                if (bn == 0) continue;
                Ring hb;
                (void)ring_init<G>(hb, b.xy + c0, bn, lane);
                if (ring_rel<G>(a.xy + a0, an - 1, hb, lane) != REL_OUT) swallowed = false;
            }
            if (swallowed) return false;
        }
    }
    return true;
}

// geometry level: every non-empty member of b lies in one member of a, and b has a non-empty member
template <int G>
__device__ inline bool polygonal_contains_polygonal_group(const DevGeo& a, int64_t ia, const DevGeo& b, int64_t ib, int lane) {
    int a0, a1, b0, b1;
    dev::geom_parts(a, ia, a0, a1);
    dev::geom_parts(b, ib, b0, b1);
    int members = 0;
    for (int q = b0; q < b1; ++q) {
        int br0, br1;
        dev::part_rings(b, q, br0, br1);
        if (br1 <= br0 || b.ring_off[br0 + 1] == b.ring_off[br0]) continue;  // an empty member adds nothing to the set
        bool inside = false;
        for (int p = a0; p < a1 && !inside; ++p) {
            int ar0, ar1;
            dev::part_rings(a, p, ar0, ar1);
            inside = polygon_contains_polygon_group<G>(a, ar0, ar1, b, br0, br1, lane);
        }
        if (!inside) return false;
        ++members;
    }
    return members > 0;
}

}  // namespace cont
}  // namespace gpk
