// gpk_polypoly.h — polygon x polygon `intersects` (geo 0.27 algorithm/intersects/{line,polygon}.rs),
// reached from the join dispatch geopolars/src/spatial_index.rs:102-104,112-123 and from the row-wise
// north-star predicate.  One lane evaluates one pair; every orientation is exact (gpk_device.h).
#pragma once

#include "gpk_device.h"

namespace gpk {

// Intersects<Line> for Line (geo 0.27 intersects/line.rs)
__device__ __forceinline__ bool point_in_rect(double px, double py, double ax, double ay, double bx, double by) {
    return dev::value_in_between(px, ax, bx) && dev::value_in_between(py, ay, by);
}
__device__ inline bool line_intersects_line(double2 a0, double2 a1, double2 b0, double2 b1) {
    if (a0.x == a1.x && a0.y == a1.y)
        return dev::orient2d(b0.x, b0.y, b1.x, b1.y, a0.x, a0.y) == 0 && point_in_rect(a0.x, a0.y, b0.x, b0.y, b1.x, b1.y);
    const int c11 = dev::orient2d(a0.x, a0.y, a1.x, a1.y, b0.x, b0.y);
    const int c12 = dev::orient2d(a0.x, a0.y, a1.x, a1.y, b1.x, b1.y);
    if (c11 != c12) {
        const int c21 = dev::orient2d(b0.x, b0.y, b1.x, b1.y, a0.x, a0.y);
        const int c22 = dev::orient2d(b0.x, b0.y, b1.x, b1.y, a1.x, a1.y);
        return c21 != c22;
    }
    if (c11 == 0)
        return point_in_rect(b0.x, b0.y, a0.x, a0.y, a1.x, a1.y) || point_in_rect(b1.x, b1.y, a0.x, a0.y, a1.x, a1.y) ||
               point_in_rect(a1.x, a1.y, b0.x, b0.y, b1.x, b1.y) || point_in_rect(a0.x, a0.y, b0.x, b0.y, b1.x, b1.y);
    return false;
}

__device__ inline bool exterior_bbox(const DevGeo& a, int r0, int r1, double4* bb) {
    if (r1 <= r0) return false;
    const int c0 = a.ring_off[r0], c1 = a.ring_off[r0 + 1];
    if (c1 == c0) return false;
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = c0; i < c1; ++i) {
        const double2 p = a.xy[i];
        mnx = p.x < mnx ? p.x : mnx;
        mny = p.y < mny ? p.y : mny;
        mxx = p.x > mxx ? p.x : mxx;
        mxy = p.y > mxy ? p.y : mxy;
    }
    *bb = make_double4(mnx, mny, mxx, mxy);
    return true;
}

// bbox over ALL rings of a polygon: used only to prune segment pairs (invalid input may have holes
// poking out of the exterior, and pruning must never change the answer)
__device__ inline double4 all_rings_bbox(const DevGeo& a, int r0, int r1) {
    double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = a.ring_off[r0]; i < a.ring_off[r1]; ++i) {
        const double2 p = a.xy[i];
        mnx = p.x < mnx ? p.x : mnx;
        mny = p.y < mny ? p.y : mny;
        mxx = p.x > mxx ? p.x : mxx;
        mxy = p.y > mxy ? p.y : mxy;
    }
    return make_double4(mnx, mny, mxx, mxy);
}

// Intersects<Polygon> for Polygon (geo 0.27 intersects/polygon.rs), one lane.  Equivalent boolean:
// bboxes not disjoint AND (some ring segment pair intersects OR a vertex of B is not Outside A OR a
// vertex of A's exterior is not Outside B).
__device__ inline bool polygon_intersects_polygon(const DevGeo& a, int ar0, int ar1, const DevGeo& b, int br0, int br1) {
    double4 ba, bb;
    if (!exterior_bbox(a, ar0, ar1, &ba) || !exterior_bbox(b, br0, br1, &bb)) return false;
    if (ba.z < bb.x || ba.w < bb.y || bb.z < ba.x || bb.w < ba.y) return false;
    const double4 fa = all_rings_bbox(a, ar0, ar1);
    for (int rb = br0; rb < br1; ++rb) {
        const int b0 = b.ring_off[rb], b1 = b.ring_off[rb + 1];
        for (int j = b0; j + 1 < b1; ++j) {
            const double2 q0 = b.xy[j], q1 = b.xy[j + 1];
            // cheap reject of this segment against A's bbox keeps the O(n*m) loop short in practice
            const double qlx = fmin(q0.x, q1.x), qhx = fmax(q0.x, q1.x), qly = fmin(q0.y, q1.y), qhy = fmax(q0.y, q1.y);
            if (qhx < fa.x || qlx > fa.z || qhy < fa.y || qly > fa.w) continue;
            for (int ra = ar0; ra < ar1; ++ra) {
                const int a0 = a.ring_off[ra], a1 = a.ring_off[ra + 1];
                for (int i = a0; i + 1 < a1; ++i) {
                    const double2 p0 = a.xy[i], p1 = a.xy[i + 1];
                    if (fmax(p0.x, p1.x) < qlx || fmin(p0.x, p1.x) > qhx || fmax(p0.y, p1.y) < qly || fmin(p0.y, p1.y) > qhy)
                        continue;  // disjoint segment boxes cannot intersect (closed test)
                    if (line_intersects_line(p0, p1, q0, q1)) return true;
                }
            }
        }
    }
    // no boundary crossing: containment of one in the other (every vertex is tested, as upstream does)
    for (int rb = br0; rb < br1; ++rb) {
        const int b0 = b.ring_off[rb], b1 = b.ring_off[rb + 1];
        for (int j = b0; j < b1; ++j) {
            const double2 q = b.xy[j];
            if (dev::polygon_pos(a, ar0, ar1, q.x, q.y) != dev::POS_OUTSIDE) return true;
        }
    }
    {
        const int a0 = a.ring_off[ar0], a1 = a.ring_off[ar0 + 1];
        for (int i = a0; i < a1; ++i) {
            const double2 p = a.xy[i];
            if (dev::polygon_pos(b, br0, br1, p.x, p.y) != dev::POS_OUTSIDE) return true;
        }
    }
    return false;
}

__device__ inline bool polygonal_intersects_polygonal(const DevGeo& a, int64_t ia, const DevGeo& b, int64_t ib) {
    int a0, a1, b0, b1;
    dev::geom_parts(a, ia, a0, a1);
    dev::geom_parts(b, ib, b0, b1);
    for (int p = a0; p < a1; ++p) {
        int ar0, ar1;
        dev::part_rings(a, p, ar0, ar1);
        for (int q = b0; q < b1; ++q) {
            int br0, br1;
            dev::part_rings(b, q, br0, br1);
            if (polygon_intersects_polygon(a, ar0, ar1, b, br0, br1)) return true;
        }
    }
    return false;
}



// ---- G lanes cooperating on one (A, B) pair ----------------------------------------------------------
// Same boolean as polygonal_intersects_polygonal.  Lane k owns segments k, k+G, ... of B (flattened over B's
// rings) and tests each against every segment of A with box pruning; a group-wide OR after every round allows
// early exit.  If no boundary pair touches, a ring never changes side of the other polygon's boundary, so ONE
// vertex per ring decides containment (upstream tests every endpoint; the outcome is identical).
template <int G>
__device__ __forceinline__ bool group_any(bool v) {
    int x = v ? 1 : 0;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) x |= __shfl_xor(x, o, 64);
    return x != 0;
}

template <int G>
__device__ inline bool polygon_intersects_polygon_group(const DevGeo& a, int ar0, int ar1, const DevGeo& b, int br0, int br1, int lane) {
    if (ar1 <= ar0 || br1 <= br0) return false;
    const int a_c0 = a.ring_off[ar0], a_c1 = a.ring_off[ar1];
    const int b_c0 = b.ring_off[br0], b_c1 = b.ring_off[br1];
    if (a.ring_off[ar0 + 1] == a_c0 || b.ring_off[br0 + 1] == b_c0) return false;  // empty exterior
    // exterior bboxes, cooperatively (has_disjoint_bboxes)
    double amnx = INFINITY, amny = INFINITY, amxx = -INFINITY, amxy = -INFINITY;
    for (int i = a_c0 + lane; i < a.ring_off[ar0 + 1]; i += G) {
        const double2 p = a.xy[i];
        amnx = fmin(amnx, p.x); amny = fmin(amny, p.y); amxx = fmax(amxx, p.x); amxy = fmax(amxy, p.y);
    }
    double bmnx = INFINITY, bmny = INFINITY, bmxx = -INFINITY, bmxy = -INFINITY;
    for (int i = b_c0 + lane; i < b.ring_off[br0 + 1]; i += G) {
        const double2 p = b.xy[i];
        bmnx = fmin(bmnx, p.x); bmny = fmin(bmny, p.y); bmxx = fmax(bmxx, p.x); bmxy = fmax(bmxy, p.y);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        amnx = fmin(amnx, __shfl_xor(amnx, o, 64)); amny = fmin(amny, __shfl_xor(amny, o, 64));
        amxx = fmax(amxx, __shfl_xor(amxx, o, 64)); amxy = fmax(amxy, __shfl_xor(amxy, o, 64));
        bmnx = fmin(bmnx, __shfl_xor(bmnx, o, 64)); bmny = fmin(bmny, __shfl_xor(bmny, o, 64));
        bmxx = fmax(bmxx, __shfl_xor(bmxx, o, 64)); bmxy = fmax(bmxy, __shfl_xor(bmxy, o, 64));
    }
    if (amxx < bmnx || amxy < bmny || bmxx < amnx || bmxy < amny) return false;

    // boundary x boundary: coordinate index j of B starts a segment unless it is the last coordinate of its ring
    for (int j0 = b_c0; j0 < b_c1; j0 += G) {
        const int j = j0 + lane;
        bool found = false;
        if (j + 1 < b_c1) {
            // ring of j: B has few rings; find by scan
            int rb = br0;
            while (rb + 1 < br1 && b.ring_off[rb + 1] <= j) ++rb;
            if (j + 1 < b.ring_off[rb + 1]) {
                const double2 q0 = b.xy[j], q1 = b.xy[j + 1];
                const double qlx = fmin(q0.x, q1.x), qhx = fmax(q0.x, q1.x), qly = fmin(q0.y, q1.y), qhy = fmax(q0.y, q1.y);
                for (int ra = ar0; ra < ar1 && !found; ++ra) {
                    const int a0 = a.ring_off[ra], a1 = a.ring_off[ra + 1];
                    for (int i = a0; i + 1 < a1; ++i) {
                        const double2 p0 = a.xy[i], p1 = a.xy[i + 1];
                        if (fmax(p0.x, p1.x) < qlx || fmin(p0.x, p1.x) > qhx || fmax(p0.y, p1.y) < qly || fmin(p0.y, p1.y) > qhy) continue;
                        if (line_intersects_line(p0, p1, q0, q1)) {
                            found = true;
                            break;
                        }
                    }
                }
            }
        }
        if (group_any<G>(found)) return true;
    }
    // containment: one vertex per ring of B against A, then A's exterior against B
    bool inside = false;
    for (int rb = br0 + lane; rb < br1; rb += G) {
        const int c = b.ring_off[rb];
        if (b.ring_off[rb + 1] > c) {
            const double2 q = b.xy[c];
            inside |= dev::polygon_pos(a, ar0, ar1, q.x, q.y) != dev::POS_OUTSIDE;
        }
    }
    if (lane == 0) {
        const double2 p = a.xy[a_c0];
        inside |= dev::polygon_pos(b, br0, br1, p.x, p.y) != dev::POS_OUTSIDE;
    }
    return group_any<G>(inside);
}

template <int G>
__device__ inline bool polygonal_intersects_polygonal_group(const DevGeo& a, int64_t ia, const DevGeo& b, int64_t ib, int lane) {
    int a0, a1, b0, b1;
    dev::geom_parts(a, ia, a0, a1);
    dev::geom_parts(b, ib, b0, b1);
    for (int p = a0; p < a1; ++p) {
        int ar0, ar1;
        dev::part_rings(a, p, ar0, ar1);
        for (int q = b0; q < b1; ++q) {
            int br0, br1;
            dev::part_rings(b, q, br0, br1);
            if (polygon_intersects_polygon_group<G>(a, ar0, ar1, b, br0, br1, lane)) return true;
        }
    }
    return false;
}

}  // namespace gpk
