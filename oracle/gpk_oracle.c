/*
 * gpk_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see gpk_oracle.h for the provenance and
 * the "parity unpinned" statement).  Build: oracle/Makefile -> oracle/libgpk_oracle.so.
 *
 * Each function names the upstream behaviour it restates (crate version pinned by the reference's
 * Cargo.lock:986-1004,2251) and the reference call site that reaches it.
 */
#include "gpk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * A.0  Orientation kernel — robust::orient2d (robust 1.1.0, Cargo.lock:2251) as used by
 * geo::kernels::RobustKernel.  Stage A is Shewchuk's floating-point filter; when it cannot decide
 * the sign we evaluate the determinant of the INPUT coordinates exactly with expansion arithmetic
 * (six error-free products, exact expansion sum).  Any exact method returns the same sign, so the
 * B/C adaptive stages of the upstream code (a speed optimisation) are not restated.
 * ---------------------------------------------------------------------------------------------- */
static int64_t g_exact_calls = 0;

static inline void two_sum(double a, double b, double* s, double* e) {
    double x = a + b;
    double bv = x - a;
    double av = x - bv;
    *s = x;
    *e = (a - av) + (b - bv);
}
static inline void two_prod(double a, double b, double* p, double* e) {
    double x = a * b;
    *p = x;
    *e = fma(a, b, -x); /* exact barring over/underflow */
}
/* grow-expansion: h = e + b, both non-overlapping, increasing magnitude; n components -> n+1 */
static inline int grow_expansion(const double* e, int n, double b, double* h) {
    double q = b;
    for (int i = 0; i < n; ++i) {
        double s, err;
        two_sum(q, e[i], &s, &err);
        h[i] = err;
        q = s;
    }
    h[n] = q;
    return n + 1;
}

static int orient2d_exact(double ax, double ay, double bx, double by, double cx, double cy) {
    /* det = ax*by - ax*cy - cx*by - ay*bx + ay*cx + cy*bx   (the cx*cy terms cancel) */
    double t[12];
    two_prod(ax, by, &t[0], &t[1]);
    two_prod(-ax, cy, &t[2], &t[3]);
    two_prod(-cx, by, &t[4], &t[5]);
    two_prod(-ay, bx, &t[6], &t[7]);
    two_prod(ay, cx, &t[8], &t[9]);
    two_prod(cy, bx, &t[10], &t[11]);
    double e[13], h[13];
    int n = 0;
    for (int i = 0; i < 12; ++i) {
        n = grow_expansion(e, n, t[i], h);
        memcpy(e, h, sizeof(double) * (size_t)n);
    }
    for (int i = n - 1; i >= 0; --i) {
        if (e[i] > 0.0) return 1;
        if (e[i] < 0.0) return -1;
    }
    return 0;
}

int32_t gpko_orient2d(double ax, double ay, double bx, double by, double cx, double cy) {
    const double detleft = (ax - cx) * (by - cy);
    const double detright = (ay - cy) * (bx - cx);
    const double det = detleft - detright;
    double detsum;
    if (detleft > 0.0) {
        if (detright <= 0.0) return det > 0.0 ? 1 : (det < 0.0 ? -1 : 0);
        detsum = detleft + detright;
    } else if (detleft < 0.0) {
        if (detright >= 0.0) return det > 0.0 ? 1 : (det < 0.0 ? -1 : 0);
        detsum = -detleft - detright;
    } else {
        return det > 0.0 ? 1 : (det < 0.0 ? -1 : 0);
    }
    /* ccwerrboundA = (3 + 16 eps) eps, eps = 2^-53 */
    const double eps = 1.1102230246251565e-16;
    const double errbound = (3.0 + 16.0 * eps) * eps * detsum;
    if (det >= errbound || -det >= errbound) return det > 0.0 ? 1 : -1;
#pragma omp atomic
    g_exact_calls++;
    return orient2d_exact(ax, ay, bx, by, cx, cy);
}
int64_t gpko_orient2d_exact_calls(void) { return g_exact_calls; }

/* ------------------------------------------------------------------------------------------------
 * A.1  coord_pos_relative_to_ring — geo 0.27 algorithm/coordinate_position.rs.  Winding number
 * with on-boundary short circuit; edge rules: upward edge includes start / excludes end, downward
 * edge excludes start / includes end, horizontal edges never counted, the crossing must be strictly
 * right of the coord (decided by the exact orientation, never by an intersection x).
 * Reached from spatial_index.rs:91-96 via Polygon::contains(Point).
 * RING RULE: non-zero winding.  It agrees with even-odd ray crossing on every simple ring.
 * ---------------------------------------------------------------------------------------------- */
static inline int value_in_between(double v, double a, double b) {
    return a > b ? (v >= b && v <= a) : (v >= a && v <= b);
}

int32_t gpko_coord_pos_ring(double cx, double cy, const double* xy, int64_t n) {
    if (n == 0) return GPKO_OUTSIDE;
    if (n == 1) return (cx == xy[0] && cy == xy[1]) ? GPKO_BOUNDARY : GPKO_OUTSIDE;
    int wn = 0;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double sx = xy[2 * i], sy = xy[2 * i + 1];
        const double ex = xy[2 * i + 2], ey = xy[2 * i + 3];
        if (sy <= cy) {
            if (ey >= cy) {
                const int o = gpko_orient2d(sx, sy, ex, ey, cx, cy);
                if (o > 0 && ey != cy)
                    wn += 1;
                else if (o == 0 && value_in_between(cx, sx, ex))
                    return GPKO_BOUNDARY;
            }
        } else if (ey <= cy) {
            const int o = gpko_orient2d(sx, sy, ex, ey, cx, cy);
            if (o < 0)
                wn -= 1;
            else if (o == 0 && value_in_between(cx, sx, ex))
                return GPKO_BOUNDARY;
        }
    }
    return wn == 0 ? GPKO_OUTSIDE : GPKO_INSIDE;
}

/* ---- GeoArrow accessors ------------------------------------------------------------------- */
typedef struct {
    int64_t r0, r1; /* ring range of one polygon */
} ring_span;

static inline int is_polygonal(const gpk_geoarrow_desc* a) {
    return a->geom_type == GPK_GEOM_POLYGON || a->geom_type == GPK_GEOM_MULTIPOLYGON;
}
static inline int is_valid_row(const gpk_geoarrow_desc* a, int64_t i) {
    return !a->validity || ((a->validity[i >> 3] >> (i & 7)) & 1);
}
/* polygons (parts) of geometry g */
static inline void geom_parts(const gpk_geoarrow_desc* a, int64_t g, int64_t* p0, int64_t* p1) {
    if (a->geom_type == GPK_GEOM_MULTIPOLYGON) {
        *p0 = a->geom_offsets[g];
        *p1 = a->geom_offsets[g + 1];
    } else {
        *p0 = g;
        *p1 = g + 1;
    }
}
static inline ring_span part_rings(const gpk_geoarrow_desc* a, int64_t p) {
    ring_span s;
    if (a->geom_type == GPK_GEOM_MULTIPOLYGON) {
        s.r0 = a->part_offsets[p];
        s.r1 = a->part_offsets[p + 1];
    } else {
        s.r0 = a->geom_offsets[p];
        s.r1 = a->geom_offsets[p + 1];
    }
    return s;
}
static inline const double* ring_xy(const gpk_geoarrow_desc* a, int64_t r, int64_t* n) {
    *n = a->ring_offsets[r + 1] - a->ring_offsets[r];
    return a->xy + 2 * (int64_t)a->ring_offsets[r];
}

/* Polygon::coordinate_position (geo 0.27): exterior Outside -> Outside; OnBoundary -> OnBoundary;
 * Inside -> holes: OnBoundary -> OnBoundary, Inside -> Outside; else Inside.  Empty polygon (no
 * rings / empty exterior) -> Outside. */
static int polygon_pos(const gpk_geoarrow_desc* a, ring_span s, double cx, double cy) {
    if (s.r1 <= s.r0) return GPKO_OUTSIDE;
    int64_t n;
    const double* ext = ring_xy(a, s.r0, &n);
    if (n == 0) return GPKO_OUTSIDE;
    const int pe = gpko_coord_pos_ring(cx, cy, ext, n);
    if (pe != GPKO_INSIDE) return pe;
    for (int64_t r = s.r0 + 1; r < s.r1; ++r) {
        const double* h = ring_xy(a, r, &n);
        const int ph = gpko_coord_pos_ring(cx, cy, h, n);
        if (ph == GPKO_BOUNDARY) return GPKO_BOUNDARY;
        if (ph == GPKO_INSIDE) return GPKO_OUTSIDE;
    }
    return GPKO_INSIDE;
}

/* MultiPolygon::coordinate_position with the mod-2 boundary rule (helper; predicates use any()) */
int32_t gpko_coord_pos_geom(const gpk_geoarrow_desc* a, int64_t g, double cx, double cy) {
    if (!is_polygonal(a)) return -1;
    int64_t p0, p1;
    geom_parts(a, g, &p0, &p1);
    int inside = 0, bcount = 0;
    for (int64_t p = p0; p < p1 && !inside; ++p) {
        const int pos = polygon_pos(a, part_rings(a, p), cx, cy);
        if (pos == GPKO_INSIDE) inside = 1;
        if (pos == GPKO_BOUNDARY) bcount++;
    }
    if (bcount % 2 == 1) return GPKO_BOUNDARY;
    return inside ? GPKO_INSIDE : GPKO_OUTSIDE;
}

/* Contains<Point> for Polygon == position Inside; for MultiPolygon == any member contains. */
static int polygonal_contains_point(const gpk_geoarrow_desc* a, int64_t g, double cx, double cy) {
    int64_t p0, p1;
    geom_parts(a, g, &p0, &p1);
    for (int64_t p = p0; p < p1; ++p)
        if (polygon_pos(a, part_rings(a, p), cx, cy) == GPKO_INSIDE) return 1;
    return 0;
}
/* Intersects<Point> for Polygon == position != Outside; MultiPolygon == any. */
static int polygonal_intersects_point(const gpk_geoarrow_desc* a, int64_t g, double cx, double cy) {
    int64_t p0, p1;
    geom_parts(a, g, &p0, &p1);
    for (int64_t p = p0; p < p1; ++p)
        if (polygon_pos(a, part_rings(a, p), cx, cy) != GPKO_OUTSIDE) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * A.2  Intersects — geo 0.27 algorithm/intersects/{line,polygon,coordinate}.rs.
 * Reached from spatial_index.rs:102-104,112-123.
 * ---------------------------------------------------------------------------------------------- */
static inline int point_in_rect(double px, double py, double ax, double ay, double bx, double by) {
    return value_in_between(px, ax, bx) && value_in_between(py, ay, by);
}
/* Intersects<Coord> for Line */
static int line_intersects_coord(const double s[2], const double e[2], double px, double py) {
    return gpko_orient2d(s[0], s[1], e[0], e[1], px, py) == 0 &&
           point_in_rect(px, py, s[0], s[1], e[0], e[1]);
}
int32_t gpko_line_intersects_line(const double a0[2], const double a1[2], const double b0[2],
                                  const double b1[2]) {
    if (a0[0] == a1[0] && a0[1] == a1[1]) return line_intersects_coord(b0, b1, a0[0], a0[1]);
    const int c11 = gpko_orient2d(a0[0], a0[1], a1[0], a1[1], b0[0], b0[1]);
    const int c12 = gpko_orient2d(a0[0], a0[1], a1[0], a1[1], b1[0], b1[1]);
    if (c11 != c12) {
        const int c21 = gpko_orient2d(b0[0], b0[1], b1[0], b1[1], a0[0], a0[1]);
        const int c22 = gpko_orient2d(b0[0], b0[1], b1[0], b1[1], a1[0], a1[1]);
        return c21 != c22;
    } else if (c11 == 0) {
        /* collinear: any endpoint inside the other's closed bounding box */
        return point_in_rect(b0[0], b0[1], a0[0], a0[1], a1[0], a1[1]) ||
               point_in_rect(b1[0], b1[1], a0[0], a0[1], a1[0], a1[1]) ||
               point_in_rect(a1[0], a1[1], b0[0], b0[1], b1[0], b1[1]) ||
               point_in_rect(a0[0], a0[1], b0[0], b0[1], b1[0], b1[1]);
    }
    return 0;
}

static int geom_bbox(const gpk_geoarrow_desc* a, int64_t g, double bb[4]);

/* Intersects<Line> for Polygon: any ring segment intersects the line, or an endpoint of the line
 * is not Outside the polygon. */
static int polygon_intersects_line(const gpk_geoarrow_desc* a, ring_span s, const double l0[2],
                                   const double l1[2]) {
    for (int64_t r = s.r0; r < s.r1; ++r) {
        int64_t n;
        const double* xy = ring_xy(a, r, &n);
        for (int64_t i = 0; i + 1 < n; ++i)
            if (gpko_line_intersects_line(xy + 2 * i, xy + 2 * i + 2, l0, l1)) return 1;
    }
    return polygon_pos(a, s, l0[0], l0[1]) != GPKO_OUTSIDE ||
           polygon_pos(a, s, l1[0], l1[1]) != GPKO_OUTSIDE;
}
/* Intersects<LineString> for Polygon: any line of the linestring intersects the polygon */
static int polygon_intersects_ring(const gpk_geoarrow_desc* a, ring_span s,
                                   const gpk_geoarrow_desc* b, int64_t rb) {
    int64_t n;
    const double* xy = ring_xy(b, rb, &n);
    for (int64_t i = 0; i + 1 < n; ++i)
        if (polygon_intersects_line(a, s, xy + 2 * i, xy + 2 * i + 2)) return 1;
    return 0;
}
static int span_bbox(const gpk_geoarrow_desc* a, ring_span s, double bb[4]) {
    /* Polygon::bounding_rect scans the exterior only */
    if (s.r1 <= s.r0) return 0;
    int64_t n;
    const double* xy = ring_xy(a, s.r0, &n);
    if (n == 0) return 0;
    bb[0] = bb[2] = xy[0];
    bb[1] = bb[3] = xy[1];
    for (int64_t i = 1; i < n; ++i) {
        bb[0] = fmin(bb[0], xy[2 * i]);
        bb[1] = fmin(bb[1], xy[2 * i + 1]);
        bb[2] = fmax(bb[2], xy[2 * i]);
        bb[3] = fmax(bb[3], xy[2 * i + 1]);
    }
    return 1;
}
static inline int bbox_disjoint(const double a[4], const double b[4]) {
    return a[2] < b[0] || a[3] < b[1] || b[2] < a[0] || b[3] < a[1];
}
/* Intersects<Polygon> for Polygon */
static int polygon_intersects_polygon(const gpk_geoarrow_desc* a, ring_span sa,
                                      const gpk_geoarrow_desc* b, ring_span sb) {
    double ba[4], bbx[4];
    const int ha = span_bbox(a, sa, ba), hb = span_bbox(b, sb, bbx);
    if (ha && hb && bbox_disjoint(ba, bbx)) return 0;
    if (!ha || !hb) return 0; /* an empty polygon intersects nothing */
    /* self intersects polygon.exterior() || any polygon.interior || polygon intersects self.exterior() */
    for (int64_t r = sb.r0; r < sb.r1; ++r)
        if (polygon_intersects_ring(a, sa, b, r)) return 1;
    return polygon_intersects_ring(b, sb, a, sa.r0);
}
static int polygonal_intersects_polygonal(const gpk_geoarrow_desc* a, int64_t ia,
                                          const gpk_geoarrow_desc* b, int64_t ib) {
    int64_t a0, a1, b0, b1;
    geom_parts(a, ia, &a0, &a1);
    geom_parts(b, ib, &b0, &b1);
    for (int64_t p = a0; p < a1; ++p)
        for (int64_t q = b0; q < b1; ++q)
            if (polygon_intersects_polygon(a, part_rings(a, p), b, part_rings(b, q))) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Contains<Polygon> for Polygon / MultiPolygon — reached from spatial_index.rs:99-101 and :107-111.
 * geo 0.27 implements it as `self.relate(rhs).is_contains()` (DE-9IM pattern T*****FF*: interiors
 * meet, nothing of rhs in the exterior of self).  For VALID operands that is the set statement
 * "rhs is not empty and rhs is a subset of self (both closed)", which is what this section decides,
 * with exact orientations only (no constructed intersection points):
 *   (1) no vertex of rhs is Outside self, and no piece of an rhs edge leaves self.  The pieces of an
 *       edge between two consecutive touch points with a ring lie on one side of that ring, and the
 *       side is visible at the touch point from the direction of the edge: against the sector of the
 *       ring's two incident edges when the touch point is a ring vertex, against the ring edge when
 *       an end point of the rhs edge lies inside a ring edge; a proper crossing has a piece on
 *       either side.  Leaving self = outside its exterior ring or inside one of its holes, so the
 *       test is ring by ring.
 *   (2) no hole of self is swallowed by rhs: with (1) a hole's interior misses every ring of rhs, so
 *       it lies inside or outside each of them as a whole; it is swallowed when it is inside (or
 *       equal to) the exterior of rhs and outside every hole of rhs.
 * A MultiPolygon contains a (valid, hence connected) polygon iff one member does; a multipolygon rhs
 * (row-wise predicate only) is contained iff it is not empty and every member is.
 * PARITY: unpinned against the upstream relate (no Rust here, no golden vector in the reference);
 * pinned to the set statement by tests/test_oracle_rational.py (rational edge splitting).  Rings
 * that are unclosed, have fewer than 4 coordinates or no turning extreme vertex make the operand
 * invalid: the answer is then false.
 * ---------------------------------------------------------------------------------------------- */
enum { DIR_IN = 1, DIR_OUT = 2 }; /* bits: a piece strictly inside / strictly outside the ring */
enum { REL_IN = 0, REL_OUT = 1, REL_SAME = 2 };

typedef struct {
    const double* xy; /* closed ring: xy[m] == xy[0] */
    int64_t m;        /* number of edges */
    int ccw;          /* +1 counter-clockwise, -1 clockwise */
} cring;

static inline int same_xy(const double* a, const double* b) { return a[0] == b[0] && a[1] == b[1]; }
static inline const double* cr_v(const cring* r, int64_t i) { return r->xy + 2 * i; }
static int64_t cr_prev_distinct(const cring* r, int64_t i) {
    for (int64_t k = 1; k < r->m; ++k) {
        const int64_t j = (i - k % r->m + r->m) % r->m;
        if (!same_xy(cr_v(r, j), cr_v(r, i))) return j;
    }
    return -1;
}
static int64_t cr_next_distinct(const cring* r, int64_t i) {
    for (int64_t k = 1; k < r->m; ++k) {
        const int64_t j = (i + k) % r->m;
        if (!same_xy(cr_v(r, j), cr_v(r, i))) return j;
    }
    return -1;
}
static inline int orient_pts(const double* a, const double* b, const double* c) {
    return gpko_orient2d(a[0], a[1], b[0], b[1], c[0], c[1]);
}
/* orientation of a simple ring = turn at its lexicographically smallest vertex (always convex) */
static int cring_init(cring* r, const double* xy, int64_t n) {
    if (n < 4 || !same_xy(xy, xy + 2 * (n - 1))) return 0;
    r->xy = xy;
    r->m = n - 1;
    int64_t k = 0;
    for (int64_t i = 0; i < r->m; ++i) {
        if (isnan(xy[2 * i]) || isnan(xy[2 * i + 1])) return 0;
        if (xy[2 * i] < xy[2 * k] || (xy[2 * i] == xy[2 * k] && xy[2 * i + 1] < xy[2 * k + 1])) k = i;
    }
    const int64_t p = cr_prev_distinct(r, k), q = cr_next_distinct(r, k);
    if (p < 0 || q < 0) return 0;
    r->ccw = orient_pts(cr_v(r, p), cr_v(r, k), cr_v(r, q));
    return r->ccw != 0;
}
/* w lies on the line through v and t (t != v, w != v): on the same side of v as t? */
static inline int same_ray(const double* v, const double* t, const double* w) {
    if (t[0] != v[0]) return (t[0] > v[0]) == (w[0] > v[0]);
    return (t[1] > v[1]) == (w[1] > v[1]);
}
/* the piece of the segment vertex_i -> w next to vertex_i: strictly inside the ring (DIR_IN),
 * strictly outside (DIR_OUT), or running along one of the two incident edges (0) */
static int dir_at_vertex(const cring* r, int64_t i, const double* w) {
    const double* v = cr_v(r, i);
    int64_t ip = cr_prev_distinct(r, i), iq = cr_next_distinct(r, i);
    if (r->ccw < 0) { const int64_t t = ip; ip = iq; iq = t; } /* walk it with the inside on the left */
    const double *p = cr_v(r, ip), *q = cr_v(r, iq);
    const int o1 = orient_pts(p, v, w), o2 = orient_pts(v, q, w);
    if (o1 == 0 && same_ray(v, p, w)) return 0;
    if (o2 == 0 && same_ray(v, q, w)) return 0;
    const int turn = orient_pts(p, v, q);
    int in;
    if (turn > 0) in = o1 > 0 && o2 > 0;      /* convex corner: between the two edges */
    else if (turn < 0) in = o1 > 0 || o2 > 0; /* reflex corner */
    else in = o1 > 0;                          /* straight through */
    return in ? DIR_IN : DIR_OUT;
}
static inline int strictly_between(const double* p, const double* a, const double* b) {
    return !same_xy(p, a) && !same_xy(p, b) && value_in_between(p[0], a[0], b[0]) && value_in_between(p[1], a[1], b[1]);
}
/* DIR_IN / DIR_OUT bits of the pieces of segment pq next to its touch points with ring r */
static int edge_ring_flags(const double* p, const double* q, const cring* r) {
    if (same_xy(p, q)) return 0;
    const double lx = fmin(p[0], q[0]), hx = fmax(p[0], q[0]), ly = fmin(p[1], q[1]), hy = fmax(p[1], q[1]);
    int fl = 0;
    for (int64_t i = 0; i < r->m; ++i) {
        const double *a = cr_v(r, i), *b = r->xy + 2 * (i + 1);
        if (fmax(a[0], b[0]) < lx || fmin(a[0], b[0]) > hx || fmax(a[1], b[1]) < ly || fmin(a[1], b[1]) > hy) continue;
        const int oa = orient_pts(p, q, a);
        if (oa == 0 && value_in_between(a[0], p[0], q[0]) && value_in_between(a[1], p[1], q[1])) {
            if (!same_xy(a, q)) fl |= dir_at_vertex(r, i, q);
            if (!same_xy(a, p)) fl |= dir_at_vertex(r, i, p);
        }
        if (same_xy(a, b)) continue;
        const int ob = orient_pts(p, q, b);
        const int op = orient_pts(a, b, p) * r->ccw, oq = orient_pts(a, b, q) * r->ccw;
        if (op == 0 && strictly_between(p, a, b)) fl |= oq > 0 ? DIR_IN : (oq < 0 ? DIR_OUT : 0);
        if (oq == 0 && strictly_between(q, a, b)) fl |= op > 0 ? DIR_IN : (op < 0 ? DIR_OUT : 0);
        if (oa * ob < 0 && op * oq < 0) fl |= DIR_IN | DIR_OUT;
    }
    return fl;
}
/* where the interior of ring h lies relative to ring r, given that it does not meet r */
static int ring_rel(const cring* h, const cring* r) {
    for (int64_t i = 0; i < h->m; ++i) {
        const int pos = gpko_coord_pos_ring(h->xy[2 * i], h->xy[2 * i + 1], r->xy, r->m + 1);
        if (pos == GPKO_INSIDE) return REL_IN;
        if (pos == GPKO_OUTSIDE) return REL_OUT;
    }
    for (int64_t i = 0; i < h->m; ++i) {
        const int fl = edge_ring_flags(cr_v(h, i), h->xy + 2 * (i + 1), r);
        if (fl & DIR_IN) return REL_IN;
        if (fl & DIR_OUT) return REL_OUT;
    }
    return REL_SAME;
}
/* rings of one polygon as crings (empty holes skipped); 0 = empty or invalid polygon */
static int64_t polygon_crings(const gpk_geoarrow_desc* a, ring_span s, cring** out) {
    *out = NULL;
    if (s.r1 <= s.r0) return 0;
    cring* rs = (cring*)malloc(sizeof(cring) * (size_t)(s.r1 - s.r0));
    int64_t k = 0;
    for (int64_t r = s.r0; r < s.r1; ++r) {
        int64_t n;
        const double* xy = ring_xy(a, r, &n);
        if (n == 0 && r > s.r0) continue;
        if (!cring_init(&rs[k], xy, n)) {
            free(rs);
            return 0;
        }
        ++k;
    }
    *out = rs;
    return k;
}
static int polygon_contains_polygon(const gpk_geoarrow_desc* a, ring_span sa, const gpk_geoarrow_desc* b, ring_span sb) {
    double ba[4], bbx[4];
    if (!span_bbox(a, sa, ba) || !span_bbox(b, sb, bbx)) return 0;
    if (bbx[0] < ba[0] || bbx[1] < ba[1] || bbx[2] > ba[2] || bbx[3] > ba[3]) return 0;
    cring *ra, *rb;
    const int64_t na = polygon_crings(a, sa, &ra);
    const int64_t nb = na ? polygon_crings(b, sb, &rb) : 0;
    int ok = na > 0 && nb > 0;
    /* (1) the boundary of b stays in a */
    for (int64_t k = 0; ok && k < nb; ++k)
        for (int64_t i = 0; ok && i < rb[k].m; ++i)
            if (polygon_pos(a, sa, rb[k].xy[2 * i], rb[k].xy[2 * i + 1]) == GPKO_OUTSIDE) ok = 0;
    for (int64_t k = 0; ok && k < nb; ++k)
        for (int64_t i = 0; ok && i < rb[k].m; ++i)
            for (int64_t j = 0; ok && j < na; ++j) {
                const int fl = edge_ring_flags(cr_v(&rb[k], i), rb[k].xy + 2 * (i + 1), &ra[j]);
                if (fl & (j == 0 ? DIR_OUT : DIR_IN)) ok = 0;
            }
    /* (2) no hole of a is swallowed by b */
    for (int64_t j = 1; ok && j < na; ++j) {
        if (ring_rel(&ra[j], &rb[0]) == REL_OUT) continue;
        int swallowed = 1;
        for (int64_t k = 1; swallowed && k < nb; ++k)
            if (ring_rel(&ra[j], &rb[k]) != REL_OUT) swallowed = 0;
        if (swallowed) ok = 0;
    }
    if (na) free(ra);
    if (nb) free(rb);
    return ok;
}
static inline int span_is_empty(const gpk_geoarrow_desc* a, ring_span s) {
    return s.r1 <= s.r0 || a->ring_offsets[s.r0 + 1] == a->ring_offsets[s.r0];
}
static int polygonal_contains_polygonal(const gpk_geoarrow_desc* a, int64_t ia, const gpk_geoarrow_desc* b, int64_t ib) {
    int64_t a0, a1, b0, b1;
    geom_parts(a, ia, &a0, &a1);
    geom_parts(b, ib, &b0, &b1);
    int members = 0;
    for (int64_t q = b0; q < b1; ++q) {
        const ring_span sb = part_rings(b, q);
        if (span_is_empty(b, sb)) continue; /* an empty member adds nothing to the set */
        int inside = 0;
        for (int64_t p = a0; p < a1 && !inside; ++p) inside = polygon_contains_polygon(a, part_rings(a, p), b, sb);
        if (!inside) return 0;
        ++members;
    }
    return members > 0;
}

/* Contains<Coord> for Line / LineString (geo 0.27 algorithm/contains/{line,line_string}.rs) —
 * reached from spatial_index.rs:126-135 */
static int line_contains_coord(const double s[2], const double e[2], double px, double py) {
    if (s[0] == e[0] && s[1] == e[1]) return s[0] == px && s[1] == py;
    if ((px == s[0] && py == s[1]) || (px == e[0] && py == e[1])) return 0;
    return line_intersects_coord(s, e, px, py);
}
static int linestring_contains_coord(const double* xy, int64_t n, double px, double py) {
    if (n == 0) return 0;
    const int closed = xy[0] == xy[2 * (n - 1)] && xy[1] == xy[2 * (n - 1) + 1];
    if ((px == xy[0] && py == xy[1]) || (px == xy[2 * (n - 1)] && py == xy[2 * (n - 1) + 1]))
        return closed;
    for (int64_t i = 0; i + 1 < n; ++i) {
        if (line_contains_coord(xy + 2 * i, xy + 2 * i + 2, px, py)) return 1;
        if (i > 0 && px == xy[2 * i] && py == xy[2 * i + 1]) return 1;
    }
    return 0;
}
static int lineal_contains_point(const gpk_geoarrow_desc* a, int64_t g, double px, double py) {
    if (a->geom_type == GPK_GEOM_LINESTRING) {
        const int64_t c0 = a->geom_offsets[g], c1 = a->geom_offsets[g + 1];
        return linestring_contains_coord(a->xy + 2 * c0, c1 - c0, px, py);
    }
    /* MULTILINESTRING: any member contains */
    for (int64_t l = a->geom_offsets[g]; l < a->geom_offsets[g + 1]; ++l) {
        int64_t n;
        const double* xy = ring_xy(a, l, &n);
        if (linestring_contains_coord(xy, n, px, py)) return 1;
    }
    return 0;
}

static inline int point_is_empty(const gpk_geoarrow_desc* a, int64_t i) {
    return isnan(a->xy[2 * i]) || isnan(a->xy[2 * i + 1]);
}

/* Dispatch table of spatial_index.rs:89-137 (join) extended to the row-wise north-star predicates.
 * Point <-> polygonal: poly.contains(point) for ANY join predicate (spatial_index.rs:91-96). */
int32_t gpko_predicate_pair(const gpk_geoarrow_desc* a, int64_t ia, const gpk_geoarrow_desc* b,
                            int64_t ib, int32_t predicate) {
    if (!is_valid_row(a, ia) || !is_valid_row(b, ib)) return 0;
    const int ta = a->geom_type, tb = b->geom_type;
    if (ta == GPK_GEOM_POINT && is_polygonal(b)) {
        if (point_is_empty(a, ia)) return 0;
        return polygonal_contains_point(b, ib, a->xy[2 * ia], a->xy[2 * ia + 1]);
    }
    if (is_polygonal(a) && tb == GPK_GEOM_POINT) {
        if (point_is_empty(b, ib)) return 0;
        return polygonal_contains_point(a, ia, b->xy[2 * ib], b->xy[2 * ib + 1]);
    }
    if (is_polygonal(a) && is_polygonal(b)) {
        if (predicate == GPK_PRED_INTERSECTS) return polygonal_intersects_polygonal(a, ia, b, ib);
        /* contains: only (Multi)Polygon x Polygon has an arm (spatial_index.rs:99-101,107-111) */
        if (predicate == GPK_PRED_CONTAINS && tb == GPK_GEOM_POLYGON) return polygonal_contains_polygonal(a, ia, b, ib);
        return 0; /* `_ => false`, spatial_index.rs:136 */
    }
    if (ta == GPK_GEOM_POINT && (tb == GPK_GEOM_LINESTRING || tb == GPK_GEOM_MULTILINESTRING)) {
        if (point_is_empty(a, ia)) return 0;
        return lineal_contains_point(b, ib, a->xy[2 * ia], a->xy[2 * ia + 1]);
    }
    if (tb == GPK_GEOM_POINT && (ta == GPK_GEOM_LINESTRING || ta == GPK_GEOM_MULTILINESTRING)) {
        if (point_is_empty(b, ib)) return 0;
        return lineal_contains_point(a, ia, b->xy[2 * ib], b->xy[2 * ib + 1]);
    }
    return 0; /* `_ => false`, spatial_index.rs:136 */
}

/* Row-wise north-star predicates use geo's trait semantics directly:
 *   contains(poly, pt) = Inside ; within(pt, poly) = contains(poly, pt) ;
 *   intersects(poly, pt) = not Outside ; intersects(poly, poly) as A.2. */
static int rowwise_pair(const gpk_geoarrow_desc* a, int64_t ia, const gpk_geoarrow_desc* b,
                        int64_t ib, int32_t predicate) {
    if (!is_valid_row(a, ia) || !is_valid_row(b, ib)) return 0;
    const int ta = a->geom_type, tb = b->geom_type;
    if (predicate == GPK_PRED_WITHIN) return rowwise_pair(b, ib, a, ia, GPK_PRED_CONTAINS);
    if (predicate == GPK_PRED_CONTAINS) {
        if (is_polygonal(a) && tb == GPK_GEOM_POINT) {
            if (point_is_empty(b, ib)) return 0;
            return polygonal_contains_point(a, ia, b->xy[2 * ib], b->xy[2 * ib + 1]);
        }
        if ((ta == GPK_GEOM_LINESTRING || ta == GPK_GEOM_MULTILINESTRING) && tb == GPK_GEOM_POINT) {
            if (point_is_empty(b, ib)) return 0;
            return lineal_contains_point(a, ia, b->xy[2 * ib], b->xy[2 * ib + 1]);
        }
        if (ta == GPK_GEOM_POINT && tb == GPK_GEOM_POINT)
            return !point_is_empty(a, ia) && a->xy[2 * ia] == b->xy[2 * ib] &&
                   a->xy[2 * ia + 1] == b->xy[2 * ib + 1];
        if (is_polygonal(a) && is_polygonal(b)) return polygonal_contains_polygonal(a, ia, b, ib);
        return 0;
    }
    /* intersects */
    if (ta == GPK_GEOM_POINT && is_polygonal(b)) {
        if (point_is_empty(a, ia)) return 0;
        return polygonal_intersects_point(b, ib, a->xy[2 * ia], a->xy[2 * ia + 1]);
    }
    if (is_polygonal(a) && tb == GPK_GEOM_POINT) {
        if (point_is_empty(b, ib)) return 0;
        return polygonal_intersects_point(a, ia, b->xy[2 * ib], b->xy[2 * ib + 1]);
    }
    if (is_polygonal(a) && is_polygonal(b)) return polygonal_intersects_polygonal(a, ia, b, ib);
    if (ta == GPK_GEOM_POINT && tb == GPK_GEOM_POINT)
        return !point_is_empty(a, ia) && a->xy[2 * ia] == b->xy[2 * ib] &&
               a->xy[2 * ia + 1] == b->xy[2 * ib + 1];
    return 0;
}

int32_t gpko_predicate_rowwise(const gpk_geoarrow_desc* a, const gpk_geoarrow_desc* b,
                               const uint32_t* b_rows, int32_t predicate, uint8_t* out,
                               int32_t n_threads) {
    if (!b_rows && a->n_geoms != b->n_geoms) return GPK_ERR_INVALID_ARGUMENT;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads > 0 ? n_threads : 1)
    for (int64_t i = 0; i < a->n_geoms; ++i)
        out[i] = (uint8_t)rowwise_pair(a, i, b, b_rows ? b_rows[i] : i, predicate);
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A.3  Bounds — geo 0.27 algorithm/bounding_rect.rs.  Polygon/MultiPolygon scan exteriors only.
 * Used by NodeEnvelope (spatial_index.rs:212-276) and GeoSeries::envelope (geoseries.rs:28-33).
 * ---------------------------------------------------------------------------------------------- */
static void bbox_acc(double bb[4], const double* xy, int64_t n, int* have) {
    for (int64_t i = 0; i < n; ++i) {
        const double x = xy[2 * i], y = xy[2 * i + 1];
        if (!*have) {
            bb[0] = bb[2] = x;
            bb[1] = bb[3] = y;
            *have = 1;
        } else {
            /* geo's get_bounding_rect: plain < / > comparisons */
            if (x < bb[0]) bb[0] = x;
            if (y < bb[1]) bb[1] = y;
            if (x > bb[2]) bb[2] = x;
            if (y > bb[3]) bb[3] = y;
        }
    }
}
static int geom_bbox(const gpk_geoarrow_desc* a, int64_t g, double bb[4]) {
    int have = 0;
    switch (a->geom_type) {
    case GPK_GEOM_POINT:
        if (point_is_empty(a, g)) return 0;
        bbox_acc(bb, a->xy + 2 * g, 1, &have);
        break;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        bbox_acc(bb, a->xy + 2 * (int64_t)a->geom_offsets[g],
                 a->geom_offsets[g + 1] - a->geom_offsets[g], &have);
        break;
    case GPK_GEOM_MULTILINESTRING:
        for (int64_t l = a->geom_offsets[g]; l < a->geom_offsets[g + 1]; ++l) {
            int64_t n;
            const double* xy = ring_xy(a, l, &n);
            bbox_acc(bb, xy, n, &have);
        }
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTIPOLYGON: {
        int64_t p0, p1;
        geom_parts(a, g, &p0, &p1);
        for (int64_t p = p0; p < p1; ++p) {
            ring_span s = part_rings(a, p);
            if (s.r1 > s.r0) {
                int64_t n;
                const double* xy = ring_xy(a, s.r0, &n);
                bbox_acc(bb, xy, n, &have);
            }
        }
        break;
    }
    default:
        return 0;
    }
    return have;
}
int32_t gpko_bounds(const gpk_geoarrow_desc* a, double* out4) {
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        double bb[4];
        if (is_valid_row(a, g) && geom_bbox(a, g, bb))
            memcpy(out4 + 4 * g, bb, sizeof bb);
        else
            out4[4 * g] = out4[4 * g + 1] = out4[4 * g + 2] = out4[4 * g + 3] = NAN;
    }
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A.5  Area — geo 0.27 algorithm/area.rs.  twice_signed_ring_area: < 3 coords or not closed -> 0;
 * coordinates shifted by the first one; sum of start.x*end.y - start.y*end.x.
 * Polygon signed area = sign(ext) * (|ext| - sum |holes|).  GeoSeries::area -> geoseries.rs:14-16.
 * ---------------------------------------------------------------------------------------------- */
static double twice_signed_ring_area(const double* xy, int64_t n) {
    if (n < 3) return 0.0;
    if (xy[0] != xy[2 * (n - 1)] || xy[1] != xy[2 * (n - 1) + 1]) return 0.0;
    const double shx = xy[0], shy = xy[1];
    double tmp = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double sx = xy[2 * i] - shx, sy = xy[2 * i + 1] - shy;
        const double ex = xy[2 * i + 2] - shx, ey = xy[2 * i + 3] - shy;
        tmp = tmp + (sx * ey - sy * ex);
    }
    return tmp;
}
static double polygon_signed_area(const gpk_geoarrow_desc* a, ring_span s) {
    if (s.r1 <= s.r0) return 0.0;
    int64_t n;
    const double* xy = ring_xy(a, s.r0, &n);
    double area = twice_signed_ring_area(xy, n) / 2.0;
    const int neg = area < 0.0;
    area = fabs(area);
    for (int64_t r = s.r0 + 1; r < s.r1; ++r) {
        xy = ring_xy(a, r, &n);
        area -= fabs(twice_signed_ring_area(xy, n) / 2.0);
    }
    return neg ? -area : area;
}
int32_t gpko_area(const gpk_geoarrow_desc* a, double* out, int32_t is_signed) {
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        double v = 0.0;
        if (!is_valid_row(a, g)) {
            out[g] = NAN;
            continue;
        }
        if (is_polygonal(a)) {
            int64_t p0, p1;
            geom_parts(a, g, &p0, &p1);
            for (int64_t p = p0; p < p1; ++p) {
                const double sa = polygon_signed_area(a, part_rings(a, p));
                /* MultiPolygon: signed = sum of signed; unsigned = sum of |signed| */
                v += is_signed ? sa : fabs(sa);
            }
        }
        out[g] = v;
    }
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A.5  Centroid — geo 0.27 algorithm/centroid.rs (CentroidOperation / WeightedCentroid):
 * dimension-aware accumulation, the highest-dimensional parts win.  GeoSeries::centroid ->
 * geoseries.rs:18-21.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int dim; /* -1 empty, 0 points, 1 lines, 2 areas */
    double w, ax, ay;
} wcentroid;

static void wc_add(wcentroid* c, int dim, double cx, double cy, double w) {
    if (dim > c->dim) {
        c->dim = dim;
        c->w = w;
        c->ax = cx * w;
        c->ay = cy * w;
    } else if (dim == c->dim) {
        c->w += w;
        c->ax += cx * w;
        c->ay += cy * w;
    }
}
static void wc_merge(wcentroid* c, const wcentroid* o) {
    if (o->dim < 0) return;
    if (o->dim > c->dim)
        *c = *o;
    else if (o->dim == c->dim) {
        c->w += o->w;
        c->ax += o->ax;
        c->ay += o->ay;
    }
}
static void wc_add_line(wcentroid* c, const double s[2], const double e[2]) {
    if (s[0] == e[0] && s[1] == e[1])
        wc_add(c, 0, s[0], s[1], 1.0);
    else
        wc_add(c, 1, (s[0] + e[0]) / 2.0, (s[1] + e[1]) / 2.0, hypot(e[0] - s[0], e[1] - s[1]));
}
static void wc_add_linestring(wcentroid* c, const double* xy, int64_t n) {
    if (n == 1) wc_add(c, 0, xy[0], xy[1], 1.0);
    for (int64_t i = 0; i + 1 < n; ++i) wc_add_line(c, xy + 2 * i, xy + 2 * i + 2);
}
static void wc_add_ring(wcentroid* c, const double* xy, int64_t n) {
    const double area = twice_signed_ring_area(xy, n) / 2.0;
    if (area == 0.0) {
        if (n == 0) return;
        if (n == 1)
            wc_add(c, 0, xy[0], xy[1], 1.0);
        else
            wc_add_linestring(c, xy, n);
        return;
    }
    const double shx = xy[0], shy = xy[1];
    double accx = 0.0, accy = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double sx = xy[2 * i] - shx, sy = xy[2 * i + 1] - shy;
        const double ex = xy[2 * i + 2] - shx, ey = xy[2 * i + 3] - shy;
        const double tmp = sx * ey - sy * ex;
        accx += (ex + sx) * tmp;
        accy += (ey + sy) * tmp;
    }
    wc_add(c, 2, accx / (6.0 * area) + shx, accy / (6.0 * area) + shy, fabs(area));
}
static void wc_add_polygon(wcentroid* c, const gpk_geoarrow_desc* a, ring_span s) {
    if (s.r1 <= s.r0) return;
    wcentroid ext = {-1, 0, 0, 0}, in = {-1, 0, 0, 0};
    int64_t n;
    const double* xy = ring_xy(a, s.r0, &n);
    wc_add_ring(&ext, xy, n);
    for (int64_t r = s.r0 + 1; r < s.r1; ++r) {
        int64_t m;
        const double* h = ring_xy(a, r, &m);
        wc_add_ring(&in, h, m);
    }
    if (ext.dim < 0) return;
    if (in.dim >= 0 && in.dim == ext.dim) {
        /* sub_assign of equal-dimension weighted centroids */
        ext.w -= in.w;
        ext.ax -= in.ax;
        ext.ay -= in.ay;
        if (ext.w == 0.0) {
            wc_add_linestring(c, xy, n);
            return;
        }
    }
    wc_merge(c, &ext);
}
int32_t gpko_centroid(const gpk_geoarrow_desc* a, double* out_xy, uint8_t* out_valid) {
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        wcentroid c = {-1, 0, 0, 0};
        if (is_valid_row(a, g)) {
            switch (a->geom_type) {
            case GPK_GEOM_POINT:
                if (!point_is_empty(a, g)) wc_add(&c, 0, a->xy[2 * g], a->xy[2 * g + 1], 1.0);
                break;
            case GPK_GEOM_MULTIPOINT:
                for (int64_t i = a->geom_offsets[g]; i < a->geom_offsets[g + 1]; ++i)
                    wc_add(&c, 0, a->xy[2 * i], a->xy[2 * i + 1], 1.0);
                break;
            case GPK_GEOM_LINESTRING:
                wc_add_linestring(&c, a->xy + 2 * (int64_t)a->geom_offsets[g],
                                  a->geom_offsets[g + 1] - a->geom_offsets[g]);
                break;
            case GPK_GEOM_MULTILINESTRING:
                for (int64_t l = a->geom_offsets[g]; l < a->geom_offsets[g + 1]; ++l) {
                    int64_t n;
                    const double* xy = ring_xy(a, l, &n);
                    wc_add_linestring(&c, xy, n);
                }
                break;
            default: {
                int64_t p0, p1;
                geom_parts(a, g, &p0, &p1);
                for (int64_t p = p0; p < p1; ++p) wc_add_polygon(&c, a, part_rings(a, p));
            }
            }
        }
        if (c.dim < 0) {
            out_xy[2 * g] = out_xy[2 * g + 1] = NAN;
            if (out_valid) out_valid[g] = 0;
        } else {
            out_xy[2 * g] = c.ax / c.w;
            out_xy[2 * g + 1] = c.ay / c.w;
            if (out_valid) out_valid[g] = 1;
        }
    }
    return GPK_OK;
}

/* euclidean_length (geoseries.rs:35-41): lines = sum of segment hypot; polygons = exterior ring */
static double ls_length(const double* xy, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i)
        s += hypot(xy[2 * i + 2] - xy[2 * i], xy[2 * i + 3] - xy[2 * i + 1]);
    return s;
}
int32_t gpko_euclidean_length(const gpk_geoarrow_desc* a, double* out) {
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        double v = 0.0;
        if (!is_valid_row(a, g)) {
            out[g] = NAN;
            continue;
        }
        switch (a->geom_type) {
        case GPK_GEOM_LINESTRING:
            v = ls_length(a->xy + 2 * (int64_t)a->geom_offsets[g],
                          a->geom_offsets[g + 1] - a->geom_offsets[g]);
            break;
        case GPK_GEOM_MULTILINESTRING:
            for (int64_t l = a->geom_offsets[g]; l < a->geom_offsets[g + 1]; ++l) {
                int64_t n;
                const double* xy = ring_xy(a, l, &n);
                v += ls_length(xy, n);
            }
            break;
        case GPK_GEOM_POLYGON:
        case GPK_GEOM_MULTIPOLYGON: {
            int64_t p0, p1;
            geom_parts(a, g, &p0, &p1);
            for (int64_t p = p0; p < p1; ++p) {
                ring_span s = part_rings(a, p);
                if (s.r1 > s.r0) {
                    int64_t n;
                    const double* xy = ring_xy(a, s.r0, &n);
                    v += ls_length(xy, n);
                }
            }
            break;
        }
        default:
            v = 0.0;
        }
        out[g] = v;
    }
    return GPK_OK;
}

/* A.6 Affine — geo 0.27 AffineTransform::apply: x' = a*x + b*y + xoff ; y' = d*x + e*y + yoff,
 * evaluated left to right with no fused multiply-add (Rust never contracts).  This file is built
 * with -ffp-contract=off so the oracle is bit-reproducible.  geoseries.rs:11-12. */
int32_t gpko_affine_transform(const gpk_geoarrow_desc* a, const double m[6], double* out_xy) {
    for (int64_t i = 0; i < a->n_coords; ++i) {
        const double x = a->xy[2 * i], y = a->xy[2 * i + 1];
        out_xy[2 * i] = (m[0] * x + m[1] * y) + m[2];
        out_xy[2 * i + 1] = (m[3] * x + m[4] * y) + m[5];
    }
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A.4  Euclidean distance — geo 0.27 algorithm/euclidean_distance.rs + geo-types 0.7.12
 * private_utils.rs (line_segment_distance, line_string_contains_point,
 * point_line_string_euclidean_distance).  GeoSeries::distance -> geoseries.rs:141-146,248-251.
 * ---------------------------------------------------------------------------------------------- */
static double line_segment_distance(double px, double py, const double s[2], const double e[2]) {
    if (s[0] == e[0] && s[1] == e[1]) return hypot(s[0] - px, s[1] - py);
    const double dx = e[0] - s[0], dy = e[1] - s[1];
    const double d2 = dx * dx + dy * dy;
    const double r = ((px - s[0]) * dx + (py - s[1]) * dy) / d2;
    if (r <= 0.0) return hypot(s[0] - px, s[1] - py);
    if (r >= 1.0) return hypot(e[0] - px, e[1] - py);
    const double q = ((s[1] - py) * dx - (s[0] - px) * dy) / d2;
    return fabs(q) * hypot(dx, dy);
}
/* geo-types private_utils::line_string_contains_point (tolerance DBL_EPSILON on |tx - ty|) */
static int ls_contains_point_eps(const double* xy, int64_t n, double px, double py) {
    if (n == 0) return 0;
    if (n == 1) return xy[0] == px && xy[1] == py;
    for (int64_t i = 0; i < n; ++i)
        if (xy[2 * i] == px && xy[2 * i + 1] == py) return 1;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double sx = xy[2 * i], sy = xy[2 * i + 1];
        const double dx = xy[2 * i + 2] - sx, dy = xy[2 * i + 3] - sy;
        int contains;
        if (dx == 0.0 && dy == 0.0) {
            contains = (px == sx && py == sy);
        } else if (dy == 0.0) {
            const double t = (px - sx) / dx;
            contains = (py == sy) && 0.0 <= t && t <= 1.0;
        } else if (dx == 0.0) {
            const double t = (py - sy) / dy;
            contains = (px == sx) && 0.0 <= t && t <= 1.0;
        } else {
            const double tx = (px - sx) / dx, ty = (py - sy) / dy;
            contains = fabs(tx - ty) <= DBL_EPSILON && 0.0 <= tx && tx <= 1.0;
        }
        if (contains) return 1;
    }
    return 0;
}
static double point_linestring_distance(double px, double py, const double* xy, int64_t n) {
    if (n == 0 || ls_contains_point_eps(xy, n, px, py)) return 0.0;
    double m = DBL_MAX;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double d = line_segment_distance(px, py, xy + 2 * i, xy + 2 * i + 2);
        m = d < m ? d : m; /* f64::min */
    }
    return m;
}
static double point_polygon_distance(double px, double py, const gpk_geoarrow_desc* a, ring_span s) {
    if (s.r1 <= s.r0) return 0.0;
    int64_t n;
    const double* ext = ring_xy(a, s.r0, &n);
    if (n == 0 || polygon_pos(a, s, px, py) != GPKO_OUTSIDE) return 0.0;
    double m = DBL_MAX;
    for (int64_t r = s.r0 + 1; r < s.r1; ++r) {
        int64_t k;
        const double* h = ring_xy(a, r, &k);
        const double d = point_linestring_distance(px, py, h, k);
        m = d < m ? d : m;
    }
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double d = line_segment_distance(px, py, ext + 2 * i, ext + 2 * i + 2);
        m = d < m ? d : m;
    }
    return m;
}
static double point_geom_distance(double px, double py, const gpk_geoarrow_desc* b, int64_t ib) {
    switch (b->geom_type) {
    case GPK_GEOM_POINT:
        return hypot(px - b->xy[2 * ib], py - b->xy[2 * ib + 1]);
    case GPK_GEOM_MULTIPOINT: {
        double m = DBL_MAX;
        for (int64_t i = b->geom_offsets[ib]; i < b->geom_offsets[ib + 1]; ++i) {
            const double d = hypot(px - b->xy[2 * i], py - b->xy[2 * i + 1]);
            m = d < m ? d : m;
        }
        return m;
    }
    case GPK_GEOM_LINESTRING:
        return point_linestring_distance(px, py, b->xy + 2 * (int64_t)b->geom_offsets[ib],
                                         b->geom_offsets[ib + 1] - b->geom_offsets[ib]);
    case GPK_GEOM_MULTILINESTRING: {
        double m = DBL_MAX;
        for (int64_t l = b->geom_offsets[ib]; l < b->geom_offsets[ib + 1]; ++l) {
            int64_t n;
            const double* xy = ring_xy(b, l, &n);
            const double d = point_linestring_distance(px, py, xy, n);
            m = d < m ? d : m;
        }
        return m;
    }
    default: {
        int64_t p0, p1;
        geom_parts(b, ib, &p0, &p1);
        double m = DBL_MAX;
        for (int64_t p = p0; p < p1; ++p) {
            const double d = point_polygon_distance(px, py, b, part_rings(b, p));
            m = d < m ? d : m;
        }
        return m;
    }
    }
}
int32_t gpko_distance_rowwise(const gpk_geoarrow_desc* a, const gpk_geoarrow_desc* b,
                              const uint32_t* b_rows, double* out, int32_t n_threads) {
    const gpk_geoarrow_desc *pt = a, *other = b;
    int swapped = 0;
    if (a->geom_type != GPK_GEOM_POINT) {
        if (b->geom_type != GPK_GEOM_POINT || b_rows) return GPK_ERR_MISMATCHED_GEOMETRY;
        pt = b;
        other = a;
        swapped = 1;
    }
    (void)swapped;
    if (!b_rows && a->n_geoms != b->n_geoms) return GPK_ERR_INVALID_ARGUMENT;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads > 0 ? n_threads : 1)
    for (int64_t i = 0; i < pt->n_geoms; ++i) {
        const int64_t j = b_rows ? b_rows[i] : i;
        if (!is_valid_row(pt, i) || !is_valid_row(other, j) || point_is_empty(pt, i)) {
            out[i] = NAN;
            continue;
        }
        out[i] = point_geom_distance(pt->xy[2 * i], pt->xy[2 * i + 1], other, j);
    }
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A.6 Convex hull — geo 0.27 algorithm/convex_hull (quickhull).  Output: closed CCW exterior, no
 * collinear vertices.  Restated as an exact-orientation monotone chain (the hull is unique; only the
 * start vertex differs between hull algorithms, so parity tests canonicalise the ring start).
 * Degenerate inputs follow upstream: 0 pts -> empty ring; 1 pt -> [p, p]... kept simple: the
 * distinct hull vertices followed by the first again.  geoseries.rs:23-26.
 * ---------------------------------------------------------------------------------------------- */
static int cmp_xy(const void* pa, const void* pb) {
    const double* a = (const double*)pa;
    const double* b = (const double*)pb;
    if (a[0] < b[0]) return -1;
    if (a[0] > b[0]) return 1;
    if (a[1] < b[1]) return -1;
    if (a[1] > b[1]) return 1;
    return 0;
}
static int64_t hull_of(const double* xy, int64_t n, double* out) {
    if (n == 0) return 0;
    double* p = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* h = (double*)malloc(sizeof(double) * 2 * (size_t)(2 * n + 2)); /* chain stack */
    memcpy(p, xy, sizeof(double) * 2 * (size_t)n);
    qsort(p, (size_t)n, 2 * sizeof(double), cmp_xy);
    int64_t m = 0; /* unique */
    for (int64_t i = 0; i < n; ++i)
        if (m == 0 || p[2 * i] != p[2 * (m - 1)] || p[2 * i + 1] != p[2 * (m - 1) + 1]) {
            p[2 * m] = p[2 * i];
            p[2 * m + 1] = p[2 * i + 1];
            ++m;
        }
    int64_t k = 0;
    if (m < 3) {
        for (int64_t i = 0; i < m; ++i) {
            h[2 * k] = p[2 * i];
            h[2 * k + 1] = p[2 * i + 1];
            ++k;
        }
    } else {
        for (int64_t i = 0; i < m; ++i) { /* lower hull */
            while (k >= 2 && gpko_orient2d(h[2 * (k - 2)], h[2 * (k - 2) + 1], h[2 * (k - 1)],
                                           h[2 * (k - 1) + 1], p[2 * i], p[2 * i + 1]) <= 0)
                --k;
            h[2 * k] = p[2 * i];
            h[2 * k + 1] = p[2 * i + 1];
            ++k;
        }
        const int64_t lo = k + 1;
        for (int64_t i = m - 2; i >= 0; --i) { /* upper hull */
            while (k >= lo && gpko_orient2d(h[2 * (k - 2)], h[2 * (k - 2) + 1], h[2 * (k - 1)],
                                            h[2 * (k - 1) + 1], p[2 * i], p[2 * i + 1]) <= 0)
                --k;
            h[2 * k] = p[2 * i];
            h[2 * k + 1] = p[2 * i + 1];
            ++k;
        }
        --k; /* last == first */
    }
    if (m >= 3 && k < 3) { /* all collinear: the chain collapses to its two extremes */
        k = 2;
        h[0] = p[0]; h[1] = p[1];
        h[2] = p[2 * (m - 1)]; h[3] = p[2 * (m - 1) + 1];
    }
    memcpy(out, h, sizeof(double) * 2 * (size_t)k);
    out[2 * k] = out[0]; /* close */
    out[2 * k + 1] = out[1];
    free(p);
    free(h);
    return k + 1;
}
static void geom_coord_range(const gpk_geoarrow_desc* a, int64_t g, int64_t* c0, int64_t* c1) {
    switch (a->geom_type) {
    case GPK_GEOM_POINT:
        *c0 = g;
        *c1 = point_is_empty(a, g) ? g : g + 1;
        break;
    case GPK_GEOM_LINESTRING:
    case GPK_GEOM_MULTIPOINT:
        *c0 = a->geom_offsets[g];
        *c1 = a->geom_offsets[g + 1];
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTILINESTRING:
        *c0 = a->ring_offsets[a->geom_offsets[g]];
        *c1 = a->ring_offsets[a->geom_offsets[g + 1]];
        break;
    default:
        *c0 = a->ring_offsets[a->part_offsets[a->geom_offsets[g]]];
        *c1 = a->ring_offsets[a->part_offsets[a->geom_offsets[g + 1]]];
    }
}
int32_t gpko_convex_hull(const gpk_geoarrow_desc* a, double* out_xy, int32_t* out_ring_offsets) {
    int64_t w = 0;
    out_ring_offsets[0] = 0;
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        int64_t c0, c1;
        geom_coord_range(a, g, &c0, &c1);
        if (is_valid_row(a, g) && c1 > c0) w += hull_of(a->xy + 2 * c0, c1 - c0, out_xy + 2 * w);
        out_ring_offsets[g + 1] = (int32_t)w;
    }
    return GPK_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Spatial join — spatial_index.rs:37-143.  Candidates = pairs whose AABBs intersect with CLOSED
 * interval semantics (rstar AABB; pinned by KA-2, spatial_index.rs:361-395), then the exact refine
 * of spatial_index.rs:89-137.  Pairs are emitted sorted by (l, r).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double x0, y0, inv_w, inv_h;
    int32_t gx, gy;
    int64_t* cell_off; /* gx*gy+1 */
    int32_t* items;
} bbox_grid;

static inline int32_t cell_of(double v, double v0, double inv, int32_t g) {
    double f = floor((v - v0) * inv);
    if (!(f >= 0.0)) return 0;
    if (f >= (double)g) return g - 1;
    return (int32_t)f;
}
static void grid_build(bbox_grid* G, const double* bb, const uint8_t* have, int64_t m) {
    double minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
    int64_t cnt = 0;
    for (int64_t i = 0; i < m; ++i)
        if (have[i]) {
            minx = fmin(minx, bb[4 * i]);
            miny = fmin(miny, bb[4 * i + 1]);
            maxx = fmax(maxx, bb[4 * i + 2]);
            maxy = fmax(maxy, bb[4 * i + 3]);
            ++cnt;
        }
    int32_t g = (int32_t)ceil(sqrt((double)(cnt > 0 ? cnt : 1)));
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    G->gx = G->gy = g;
    G->x0 = cnt ? minx : 0.0;
    G->y0 = cnt ? miny : 0.0;
    const double w = cnt ? (maxx - minx) : 1.0, h = cnt ? (maxy - miny) : 1.0;
    G->inv_w = w > 0 ? g / w : 0.0;
    G->inv_h = h > 0 ? g / h : 0.0;
    const int64_t nc = (int64_t)g * g;
    G->cell_off = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            int64_t s = 0;
            for (int64_t c = 0; c < nc; ++c) {
                const int64_t t = G->cell_off[c];
                G->cell_off[c] = s;
                s += t;
            }
            G->cell_off[nc] = s;
            G->items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(s > 0 ? s : 1));
        }
        for (int64_t i = 0; i < m; ++i) {
            if (!have[i]) continue;
            const int32_t cx0 = cell_of(bb[4 * i], G->x0, G->inv_w, g);
            const int32_t cx1 = cell_of(bb[4 * i + 2], G->x0, G->inv_w, g);
            const int32_t cy0 = cell_of(bb[4 * i + 1], G->y0, G->inv_h, g);
            const int32_t cy1 = cell_of(bb[4 * i + 3], G->y0, G->inv_h, g);
            for (int32_t cy = cy0; cy <= cy1; ++cy)
                for (int32_t cx = cx0; cx <= cx1; ++cx) {
                    const int64_t c = (int64_t)cy * g + cx;
                    if (pass == 0)
                        G->cell_off[c]++;
                    else
                        G->items[G->cell_off[c]++] = (int32_t)i;
                }
        }
        if (pass == 1) { /* restore offsets shifted by the fill */
            for (int64_t c = nc; c > 0; --c) G->cell_off[c] = G->cell_off[c - 1];
            G->cell_off[0] = 0;
        }
    }
}
static int cmp_i32(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : (x > y);
}

int32_t gpko_spatial_join(const gpk_geoarrow_desc* left, const gpk_geoarrow_desc* right,
                          int32_t predicate, int32_t mode, int32_t n_threads, uint32_t* out_counts,
                          uint32_t* out_pairs, int64_t pair_capacity, int64_t* n_pairs,
                          int32_t* used_threads) {
    const int64_t nl = left->n_geoms, nr = right->n_geoms;
    double* rbb = (double*)malloc(sizeof(double) * 4 * (size_t)(nr > 0 ? nr : 1));
    uint8_t* rhave = (uint8_t*)malloc((size_t)(nr > 0 ? nr : 1));
    for (int64_t j = 0; j < nr; ++j) rhave[j] = is_valid_row(right, j) && geom_bbox(right, j, rbb + 4 * j);
    bbox_grid G;
    memset(&G, 0, sizeof G);
    if (mode == 1) grid_build(&G, rbb, rhave, nr);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    if (used_threads) *used_threads = n_threads;
    /* per-thread hit lists over contiguous row blocks keep the output sorted by l */
    const int64_t nblk = (nl + 4095) / 4096;
    uint32_t** blk_pairs = (uint32_t**)calloc((size_t)(nblk > 0 ? nblk : 1), sizeof(uint32_t*));
    int64_t* blk_n = (int64_t*)calloc((size_t)(nblk > 0 ? nblk : 1), sizeof(int64_t));
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t b = 0; b < nblk; ++b) {
        int64_t cap = 1024, cnt = 0;
        uint32_t* buf = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)cap);
        int32_t* cand = NULL;
        int64_t cand_cap = 0;
        const int64_t i1 = (b + 1) * 4096 < nl ? (b + 1) * 4096 : nl;
        for (int64_t i = b * 4096; i < i1; ++i) {
            double lb[4];
            uint32_t hits = 0;
            if (is_valid_row(left, i) && geom_bbox(left, i, lb)) {
                int64_t nc = 0;
                if (mode == 1) {
                    const int32_t cx0 = cell_of(lb[0], G.x0, G.inv_w, G.gx), cx1 = cell_of(lb[2], G.x0, G.inv_w, G.gx);
                    const int32_t cy0 = cell_of(lb[1], G.y0, G.inv_h, G.gy), cy1 = cell_of(lb[3], G.y0, G.inv_h, G.gy);
                    for (int32_t cy = cy0; cy <= cy1; ++cy)
                        for (int32_t cx = cx0; cx <= cx1; ++cx) {
                            const int64_t c = (int64_t)cy * G.gx + cx;
                            for (int64_t k = G.cell_off[c]; k < G.cell_off[c + 1]; ++k) {
                                if (nc == cand_cap) {
                                    cand_cap = cand_cap ? 2 * cand_cap : 64;
                                    cand = (int32_t*)realloc(cand, sizeof(int32_t) * (size_t)cand_cap);
                                }
                                cand[nc++] = G.items[k];
                            }
                        }
                    if (nc > 1) {
                        qsort(cand, (size_t)nc, sizeof(int32_t), cmp_i32);
                        int64_t u = 0;
                        for (int64_t k = 0; k < nc; ++k)
                            if (u == 0 || cand[k] != cand[u - 1]) cand[u++] = cand[k];
                        nc = u;
                    }
                }
                const int64_t total = mode == 1 ? nc : nr;
                for (int64_t k = 0; k < total; ++k) {
                    const int64_t j = mode == 1 ? cand[k] : k;
                    if (!rhave[j] || bbox_disjoint(lb, rbb + 4 * j)) continue;
                    const int r = gpko_predicate_pair(left, i, right, j, predicate);
                    if (r < 0) {
                        err = 1;
                        continue;
                    }
                    if (r) {
                        if (cnt == cap) {
                            cap *= 2;
                            buf = (uint32_t*)realloc(buf, sizeof(uint32_t) * 2 * (size_t)cap);
                        }
                        buf[2 * cnt] = (uint32_t)i;
                        buf[2 * cnt + 1] = (uint32_t)j;
                        ++cnt;
                        ++hits;
                    }
                }
            }
            if (out_counts) out_counts[i] = hits;
        }
        free(cand);
        blk_pairs[b] = buf;
        blk_n[b] = cnt;
    }
    int64_t total = 0;
    for (int64_t b = 0; b < nblk; ++b) total += blk_n[b];
    *n_pairs = total;
    int32_t rc = err ? GPK_ERR_MISMATCHED_GEOMETRY : GPK_OK;
    if (out_pairs) {
        if (total > pair_capacity)
            rc = GPK_ERR_CAPACITY;
        else {
            int64_t w = 0;
            for (int64_t b = 0; b < nblk; ++b) {
                memcpy(out_pairs + 2 * w, blk_pairs[b], sizeof(uint32_t) * 2 * (size_t)blk_n[b]);
                w += blk_n[b];
            }
        }
    }
    for (int64_t b = 0; b < nblk; ++b) free(blk_pairs[b]);
    free(blk_pairs);
    free(blk_n);
    free(rbb);
    free(rhave);
    free(G.cell_off);
    free(G.items);
    return rc;
}

/* ================================================================================================
 * geodesic_length (geoseries.rs:52-58) — geo 0.27 haversine_distance.rs / vincenty_distance.rs / geodesic_distance.rs, summed over
 * the segments of every linestring of a row (metres; coordinates are lon, lat degrees).  Pinned by geo's documented examples
 * for New York -> London (haversine 5_570_230 m, vincenty 5_585_234 m, geodesic 5_585_234 m, rounded; GeodesicLength's New York
 * -> London -> Osaka 15_109_158 m), by Vincenty's published Flinders Peak -> Buninyong line (54 972.271 m) and, for "geodesic"
 * (Karney's algorithm, which geo reaches through geographiclib-rs), by GeographicLib's published lines: Wellington ->
 * Salamanca 19 959 679.267 m, JFK -> LHR 5 551 759.400 m, the GeodTest line 35.60777 -139.44815 -> -11.17491 -69.95921
 * (8 935 244.5604818 m), the nearly antipodal 0 0 -> 0.5 179.5 of Karney 2013 (19 936 288.579 m):
 * tests/test_oracle_lineal_ops.py. */
#define KQ static
#define K_A 6378137.0
#define K_F (1.0 / 298.257223563)
#define K_F1 (1.0 - K_F)
#define K_E2 (K_F * (2.0 - K_F))
#define K_EP2 (K_E2 / (K_F1 * K_F1))
#define K_N (K_F / (2.0 - K_F))
#define K_B (K_A * K_F1)
#define K_PI 3.14159265358979323846
#define K_DEGREE (K_PI / 180.0)
#define K_TINY 1.4916681462400413e-154 /* sqrt(DBL_MIN) */
#define K_TOL0 2.220446049250313e-16   /* DBL_EPSILON */
#define K_TOL1 (200.0 * K_TOL0)
#define K_TOL2 1.4901161193847656e-08  /* sqrt(DBL_EPSILON) */
#define K_TOLB (K_TOL0 * K_TOL2)
#define K_XTHRESH (1000.0 * K_TOL2)
#define K_ETOL2 (0.1 * K_TOL2 / sqrt(fmax(0.001, fabs(K_F)) * fmin(1.0, 1.0 - K_F / 2.0) / 2.0))
#define K_MAXIT1 20
#define K_MAXIT2 83
/* ---- Karney's inverse geodesic on WGS84 (C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87, 2013; the order-6
 * series and the Newton / bisection scheme of GeographicLib's Geodesic::Inverse, which geo 0.27 reaches through the
 * geographiclib-rs crate: GeodesicDistance / GeodesicLength).  Only the distance s12 is produced. ------------------------------ */
KQ double k_sq(double x) { return x * x; }
KQ void k_norm2(double* s, double* c) {
    const double r = hypot(*s, *c);
    *s /= r;
    *c /= r;
}
KQ double k_sumx(double u, double v, double* t) {
    const double s = u + v;
    double up = s - v, vpp = s - up;
    up -= u;
    vpp -= v;
    *t = s != 0.0 ? 0.0 - (up + vpp) : s;
    return s;
}
KQ double k_ang_round(double x) {
    const double z = 1.0 / 16.0;
    double y = fabs(x);
    y = y < z ? z - (z - y) : y;
    return copysign(y, x);
}
KQ double k_ang_diff(double x, double y, double* e) {
    double t;
    double d = k_sumx(remainder(-x, 360.0), remainder(y, 360.0), &t);
    d = k_sumx(remainder(d, 360.0), t, &t);
    if (d == 0.0 || fabs(d) == 180.0) d = copysign(d, t == 0.0 ? y - x : -t);
    *e = t;
    return d;
}
KQ void k_sincosd(double x, double* sinx, double* cosx) {
    int q = 0;
    double r = remquo(x, 90.0, &q);
    r *= K_DEGREE;
    const double s = sin(r), c = cos(r);
    double sx, cx;
    switch ((unsigned)q & 3u) {
        case 0u: sx = s; cx = c; break;
        case 1u: sx = c; cx = -s; break;
        case 2u: sx = -s; cx = -c; break;
        default: sx = -c; cx = s; break;
    }
    cx += 0.0;
    if (sx == 0.0) sx = copysign(sx, x);
    *sinx = sx;
    *cosx = cx;
}
/* sum of c[k] sin(2 k x), k = 1 .. n (Clenshaw) */
KQ double k_sin_series(double sinx, double cosx, const double* c, int n) {
    const double ar = 2.0 * (cosx - sinx) * (cosx + sinx);
    c += n + 1;
    double y0 = (n & 1) ? *--c : 0.0, y1 = 0.0;
    n /= 2;
    while (n--) {
        y1 = ar * y0 - y1 + *--c;
        y0 = ar * y1 - y0 + *--c;
    }
    return 2.0 * sinx * cosx * y0;
}
KQ double k_A1m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (e2 + 4.0) + 64.0) / 256.0;
    return (t + eps) / (1.0 - eps);
}
KQ void k_C1f(double eps, double* c) {
    const double e2 = eps * eps;
    double d = eps;
    c[1] = d * ((6.0 - e2) * e2 - 16.0) / 32.0;
    d *= eps;
    c[2] = d * ((64.0 - 9.0 * e2) * e2 - 128.0) / 2048.0;
    d *= eps;
    c[3] = d * (9.0 * e2 - 16.0) / 768.0;
    d *= eps;
    c[4] = d * (3.0 * e2 - 5.0) / 512.0;
    d *= eps;
    c[5] = -7.0 * d / 1280.0;
    d *= eps;
    c[6] = -7.0 * d / 2048.0;
}
KQ double k_A2m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (-11.0 * e2 - 28.0) - 192.0) / 256.0;
    return (t - eps) / (1.0 + eps);
}
KQ void k_C2f(double eps, double* c) {
    const double e2 = eps * eps;
    double d = eps;
    c[1] = d * (e2 * (e2 + 2.0) + 16.0) / 32.0;
    d *= eps;
    c[2] = d * (e2 * (35.0 * e2 + 64.0) + 384.0) / 2048.0;
    d *= eps;
    c[3] = d * (15.0 * e2 + 80.0) / 768.0;
    d *= eps;
    c[4] = d * (7.0 * e2 + 35.0) / 512.0;
    d *= eps;
    c[5] = 63.0 * d / 1280.0;
    d *= eps;
    c[6] = 77.0 * d / 2048.0;
}
KQ double k_A3f(double eps) {
    const double n = K_N;
    double y = -3.0 / 128.0;
    y = y * eps + (-2.0 * n - 3.0) / 64.0;
    y = y * eps + ((-n - 3.0) * n - 1.0) / 16.0;
    y = y * eps + ((3.0 * n - 1.0) * n - 2.0) / 8.0;
    y = y * eps + (n - 1.0) / 2.0;
    y = y * eps + 1.0;
    return y;
}
KQ void k_C3f(double eps, double* c) {
    const double n = K_N;
    double m = eps;  /* eps^l */
    c[1] = m * ((((3.0 / 128.0) * eps + (2.0 * n + 5.0) / 128.0) * eps + ((-n + 3.0) * n + 3.0) / 64.0) * eps + ((-n) * n + 1.0) / 8.0) * eps
           + m * ((-n + 1.0) / 4.0);
    m *= eps;
    c[2] = m * ((((5.0 / 256.0) * eps + (n + 3.0) / 128.0) * eps + ((-3.0 * n - 2.0) * n + 3.0) / 64.0) * eps + ((n - 3.0) * n + 2.0) / 32.0);
    m *= eps;
    c[3] = m * (((7.0 / 512.0) * eps + (-10.0 * n + 9.0) / 384.0) * eps + ((5.0 * n - 9.0) * n + 5.0) / 192.0);
    m *= eps;
    c[4] = m * ((7.0 / 512.0) * eps + (-14.0 * n + 7.0) / 512.0);
    m *= eps;
    c[5] = m * (21.0 / 2560.0);
}
/* s12b (distance / b) and m12b (reduced length / b) of the arc sig1 .. sig2; *m0 = A1 - A2 */
KQ void k_lengths(double eps, double sig12, double ssig1, double csig1, double dn1, double ssig2, double csig2, double dn2, double* s12b, double* m12b,
               double* m0) {
    double ca[7], cb[7];
    k_C1f(eps, ca);
    k_C2f(eps, cb);
    const double A1 = k_A1m1f(eps), A2 = k_A2m1f(eps);
    const double B1 = k_sin_series(ssig2, csig2, ca, 6) - k_sin_series(ssig1, csig1, ca, 6);
    const double B2 = k_sin_series(ssig2, csig2, cb, 6) - k_sin_series(ssig1, csig1, cb, 6);
    *s12b = (1.0 + A1) * (sig12 + B1);
    *m0 = A1 - A2;
    const double J12 = (A1 - A2) * sig12 + ((1.0 + A1) * B1 - (1.0 + A2) * B2);
    *m12b = dn2 * (csig1 * ssig2) - dn1 * (ssig1 * csig2) - csig1 * csig2 * J12;
}
KQ double k_astroid(double x, double y) {
    const double p = x * x, q = y * y, r = (p + q - 1.0) / 6.0;
    if (q == 0.0 && r <= 0.0) return 0.0;
    const double S = p * q / 4.0, r2 = r * r, r3 = r * r2, disc = S * (S + 2.0 * r3);
    double u = r;
    if (disc >= 0.0) {
        double T3 = S + r3;
        T3 += T3 < 0.0 ? -sqrt(disc) : sqrt(disc);
        const double T = cbrt(T3);
        u += T + (T != 0.0 ? r2 / T : 0.0);
    } else {
        const double ang = atan2(sqrt(-disc), -(S + r3));
        u += 2.0 * r * cos(ang / 3.0);
    }
    const double v = sqrt(u * u + q), uv = u < 0.0 ? q / (v - u) : u + v, w = (uv - q) / (2.0 * v);
    return uv / (sqrt(uv + w * w) + w);
}
/* starting point of Newton's method; returns sig12 >= 0 when the line is short enough to be settled at once */
KQ double k_inverse_start(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2, double lam12, double slam12, double clam12,
                       double* psalp1, double* pcalp1, double* psalp2, double* pcalp2, double* pdnm) {
    double salp1 = 0.0, calp1 = 0.0, salp2 = 0.0, calp2 = 0.0, dnm = 0.0;
    double sig12 = -1.0;
    const double sbet12 = sbet2 * cbet1 - cbet2 * sbet1, cbet12 = cbet2 * cbet1 + sbet2 * sbet1, sbet12a = sbet2 * cbet1 + cbet2 * sbet1;
    const int shortline = cbet12 >= 0.0 && sbet12 < 0.5 && cbet2 * lam12 < 0.5;
    double somg12, comg12;
    if (shortline) {
        double sbetm2 = k_sq(sbet1 + sbet2);
        sbetm2 /= sbetm2 + k_sq(cbet1 + cbet2);
        dnm = sqrt(1.0 + K_EP2 * sbetm2);
        const double omg12 = lam12 / (K_F1 * dnm);
        somg12 = sin(omg12);
        comg12 = cos(omg12);
    } else {
        somg12 = slam12;
        comg12 = clam12;
    }
    salp1 = cbet2 * somg12;
    calp1 = comg12 >= 0.0 ? sbet12 + cbet2 * sbet1 * k_sq(somg12) / (1.0 + comg12) : sbet12a - cbet2 * sbet1 * k_sq(somg12) / (1.0 - comg12);
    const double ssig12 = hypot(salp1, calp1), csig12 = sbet1 * sbet2 + cbet1 * cbet2 * comg12;
    if (shortline && ssig12 < K_ETOL2) {
        salp2 = cbet1 * somg12;
        calp2 = sbet12 - cbet1 * sbet2 * (comg12 >= 0.0 ? k_sq(somg12) / (1.0 + comg12) : 1.0 - comg12);
        k_norm2(&salp2, &calp2);
        sig12 = atan2(ssig12, csig12);
    } else if (fabs(K_N) > 0.1 || csig12 >= 0.0 || ssig12 >= 6.0 * fabs(K_N) * K_PI * k_sq(cbet1)) {
        /* nothing to do: zeroth-order spherical approximation is fine */
    } else {
        /* nearly antipodal points: scale lam12 and bet2 to x, y and solve the astroid problem */
        const double lam12x = atan2(-slam12, -clam12);
        const double k2 = k_sq(sbet1) * K_EP2, eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
        const double lamscale = K_F * cbet1 * k_A3f(eps) * K_PI, betscale = lamscale * cbet1;
        const double x = lam12x / lamscale, y = sbet12a / betscale;
        if (y > -K_TOL1 && x > -1.0 - K_XTHRESH) {
            salp1 = fmin(1.0, -x);
            calp1 = -sqrt(1.0 - k_sq(salp1));
        } else {
            const double k = k_astroid(x, y);
            const double omg12a = lamscale * (-x * k / (1.0 + k));
            somg12 = sin(omg12a);
            comg12 = -cos(omg12a);
            salp1 = cbet2 * somg12;
            calp1 = sbet12a - cbet2 * sbet1 * k_sq(somg12) / (1.0 - comg12);
        }
    }
    if (!(salp1 <= 0.0)) {
        k_norm2(&salp1, &calp1);
    } else {
        salp1 = 1.0;
        calp1 = 0.0;
    }
    *psalp1 = salp1;
    *pcalp1 = calp1;
    *psalp2 = salp2;
    *pcalp2 = calp2;
    *pdnm = dnm;
    return sig12;
}
/* lambda12(alp1) - lam12 is what Newton's method drives to zero */
KQ double k_lambda12(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2, double salp1, double calp1, double slam120,
                  double clam120, double* psalp2, double* pcalp2, double* psig12, double* pssig1, double* pcsig1, double* pssig2, double* pcsig2,
                  double* peps, int diffp, double* pdlam12) {
    if (sbet1 == 0.0 && calp1 == 0.0) calp1 = -K_TINY;
    const double salp0 = salp1 * cbet1, calp0 = hypot(calp1, salp1 * sbet1);
    double ssig1 = sbet1, csig1 = calp1 * cbet1;
    const double somg1 = salp0 * sbet1, comg1 = csig1;
    k_norm2(&ssig1, &csig1);
    const double salp2 = cbet2 != cbet1 ? salp0 / cbet2 : salp1;
    const double calp2 = (cbet2 != cbet1 || fabs(sbet2) != -sbet1)
                             ? sqrt(k_sq(calp1 * cbet1) + (cbet1 < -sbet1 ? (cbet2 - cbet1) * (cbet1 + cbet2) : (sbet1 - sbet2) * (sbet1 + sbet2))) / cbet2
                             : fabs(calp1);
    double ssig2 = sbet2, csig2 = calp2 * cbet2;
    const double somg2 = salp0 * sbet2, comg2 = csig2;
    k_norm2(&ssig2, &csig2);
    const double sig12 = atan2(fmax(0.0, csig1 * ssig2 - ssig1 * csig2), csig1 * csig2 + ssig1 * ssig2);
    const double somg12 = fmax(0.0, comg1 * somg2 - somg1 * comg2), comg12 = comg1 * comg2 + somg1 * somg2;
    const double eta = atan2(somg12 * clam120 - comg12 * slam120, comg12 * clam120 + somg12 * slam120);
    const double k2 = k_sq(calp0) * K_EP2, eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
    double c3[6];
    k_C3f(eps, c3);
    const double B312 = k_sin_series(ssig2, csig2, c3, 5) - k_sin_series(ssig1, csig1, c3, 5);
    const double domg12 = -K_F * k_A3f(eps) * salp0 * (sig12 + B312);
    const double lam12 = eta + domg12;
    if (diffp) {
        if (calp2 == 0.0) {
            *pdlam12 = -2.0 * K_F1 * dn1 / sbet1;
        } else {
            double s12b, m12b, m0;
            k_lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12b, &m12b, &m0);
            *pdlam12 = m12b * K_F1 / (calp2 * cbet2);
        }
    }
    *psalp2 = salp2;
    *pcalp2 = calp2;
    *psig12 = sig12;
    *pssig1 = ssig1;
    *pcsig1 = csig1;
    *pssig2 = ssig2;
    *pcsig2 = csig2;
    *peps = eps;
    return lam12;
}
/* metres between (lon1, lat1) and (lon2, lat2), degrees */
KQ double k_geodesic_m(double lon1, double lat1, double lon2, double lat2) {
    double lon12s;
    double lon12 = k_ang_diff(lon1, lon2, &lon12s);
    int lonsign = signbit(lon12) ? -1 : 1;
    lon12 *= lonsign;
    lon12s *= lonsign;
    const double lam12 = lon12 * K_DEGREE;
    double slam12, clam12;
    k_sincosd(lon12, &slam12, &clam12);
    lon12s = (180.0 - lon12) - lon12s;  /* the supplementary longitude difference */
    lat1 = k_ang_round(fabs(lat1) > 90.0 ? NAN : lat1);
    lat2 = k_ang_round(fabs(lat2) > 90.0 ? NAN : lat2);
    const int swapp = fabs(lat1) < fabs(lat2) || lat2 != lat2 ? -1 : 1;
    if (swapp < 0) {
        lonsign *= -1;
        const double t = lat1;
        lat1 = lat2;
        lat2 = t;
    }
    const int latsign = signbit(lat1) ? 1 : -1;
    lat1 *= latsign;
    lat2 *= latsign;
    double sbet1, cbet1, sbet2, cbet2;
    k_sincosd(lat1, &sbet1, &cbet1);
    sbet1 *= K_F1;
    k_norm2(&sbet1, &cbet1);
    cbet1 = fmax(K_TINY, cbet1);
    k_sincosd(lat2, &sbet2, &cbet2);
    sbet2 *= K_F1;
    k_norm2(&sbet2, &cbet2);
    cbet2 = fmax(K_TINY, cbet2);
    if (cbet1 < -sbet1) {
        if (cbet2 == cbet1) sbet2 = copysign(sbet1, sbet2);
    } else {
        if (fabs(sbet2) == -sbet1) cbet2 = cbet1;
    }
    const double dn1 = sqrt(1.0 + K_EP2 * k_sq(sbet1)), dn2 = sqrt(1.0 + K_EP2 * k_sq(sbet2));
    double s12x = 0.0, m12x = 0.0, sig12 = 0.0;
    int meridian = lat1 == -90.0 || slam12 == 0.0;
    if (meridian) {
        const double calp1 = clam12, calp2 = 1.0;
        const double ssig1 = sbet1, csig1 = calp1 * cbet1, ssig2 = sbet2, csig2 = calp2 * cbet2;
        sig12 = atan2(fmax(0.0, csig1 * ssig2 - ssig1 * csig2), csig1 * csig2 + ssig1 * ssig2);
        double m0;
        k_lengths(K_N, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12x, &m12x, &m0);
        if (sig12 < 1.0 || m12x >= 0.0) {
            if (sig12 < 3.0 * K_TINY || (sig12 < K_TOL0 && (s12x < 0.0 || m12x < 0.0))) sig12 = m12x = s12x = 0.0;
            s12x *= K_B;
        } else {
            meridian = 0;  /* m12 < 0: the prolate case, or the geodesic runs over a pole the long way */
        }
    }
    if (!meridian && sbet1 == 0.0 && lon12s >= K_F * 180.0) {
        s12x = K_A * lam12;  /* along the equator */
    } else if (!meridian) {
        double salp1, calp1, salp2, calp2, dnm;
        sig12 = k_inverse_start(sbet1, cbet1, dn1, sbet2, cbet2, dn2, lam12, slam12, clam12, &salp1, &calp1, &salp2, &calp2, &dnm);
        if (sig12 >= 0.0) {
            s12x = sig12 * K_B * dnm;  /* short line */
        } else {
            double ssig1 = 0.0, csig1 = 0.0, ssig2 = 0.0, csig2 = 0.0, eps = 0.0;
            double salp1a = K_TINY, calp1a = 1.0, salp1b = K_TINY, calp1b = -1.0;
            int tripn = 0, tripb = 0;
            for (int numit = 0; numit < K_MAXIT2; ++numit) {
                double dv = 0.0;
                const double v = k_lambda12(sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam12, clam12, &salp2, &calp2, &sig12, &ssig1, &csig1,
                                            &ssig2, &csig2, &eps, numit < K_MAXIT1, &dv);
                if (tripb || !(fabs(v) >= (tripn ? 8.0 : 1.0) * K_TOL0) || numit == K_MAXIT2 - 1) break;
                if (v > 0.0 && (numit > K_MAXIT1 || calp1 / salp1 > calp1b / salp1b)) {
                    salp1b = salp1;
                    calp1b = calp1;
                } else if (v < 0.0 && (numit > K_MAXIT1 || calp1 / salp1 < calp1a / salp1a)) {
                    salp1a = salp1;
                    calp1a = calp1;
                }
                if (numit < K_MAXIT1 && dv > 0.0) {
                    const double dalp1 = -v / dv;
                    if (fabs(dalp1) < K_PI) {
                        const double sdalp1 = sin(dalp1), cdalp1 = cos(dalp1), nsalp1 = salp1 * cdalp1 + calp1 * sdalp1;
                        if (nsalp1 > 0.0) {
                            calp1 = calp1 * cdalp1 - salp1 * sdalp1;
                            salp1 = nsalp1;
                            k_norm2(&salp1, &calp1);
                            tripn = fabs(v) <= 16.0 * K_TOL0;
                            continue;
                        }
                    }
                }
                salp1 = (salp1a + salp1b) / 2.0;
                calp1 = (calp1a + calp1b) / 2.0;
                k_norm2(&salp1, &calp1);
                tripn = 0;
                tripb = (fabs(salp1a - salp1) + (calp1a - calp1) < K_TOLB || fabs(salp1 - salp1b) + (calp1 - calp1b) < K_TOLB);
            }
            double m0;
            k_lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12x, &m12x, &m0);
            s12x *= K_B;
        }
    }
    return 0.0 + s12x;
}
#undef KQ

static double o_haversine(double lon1, double lat1, double lon2, double lat2) {
    const double rad = 0.017453292519943295;
    const double t1 = lat1 * rad, t2 = lat2 * rad, dt = (lat2 - lat1) * rad, dl = (lon2 - lon1) * rad;
    const double sh = sin(dt / 2.0), sl = sin(dl / 2.0);
    const double a = sh * sh + cos(t1) * cos(t2) * (sl * sl);
    return 6371008.8 * (2.0 * asin(sqrt(a)));
}
static double o_vincenty(double lon1, double lat1, double lon2, double lat2) {
    const double rad = 0.017453292519943295, a = 6378137.0, b = 6356752.314245, f = 1.0 / 298.257223563;
    const double L = (lon2 - lon1) * rad;
    const double U1 = atan((1.0 - f) * tan(lat1 * rad)), U2 = atan((1.0 - f) * tan(lat2 * rad));
    const double sU1 = sin(U1), cU1 = cos(U1), sU2 = sin(U2), cU2 = cos(U2);
    double lam = L, lam_p, sS = 0, cS = 0, sig = 0, c2A = 0, c2SM = 0;
    int it = 100;
    for (;;) {
        const double sl = sin(lam), cl = cos(lam);
        const double t0 = cU2 * sl, t1 = cU1 * sU2 - sU1 * cU2 * cl;
        sS = sqrt(t0 * t0 + t1 * t1);
        if (sS == 0.0) return (lon1 == lon2 && lat1 == lat2) ? 0.0 : NAN;
        cS = sU1 * sU2 + cU1 * cU2 * cl;
        sig = atan2(sS, cS);
        const double sA = cU1 * cU2 * sl / sS;
        c2A = 1.0 - sA * sA;
        c2SM = c2A == 0.0 ? 0.0 : cS - 2.0 * sU1 * sU2 / c2A;
        const double C = f / 16.0 * c2A * (4.0 + f * (4.0 - 3.0 * c2A));
        lam_p = lam;
        lam = L + (1.0 - C) * f * sA * (sig + C * sS * (c2SM + C * cS * (-1.0 + 2.0 * c2SM * c2SM)));
        if (fabs(lam - lam_p) <= 1e-12) break;
        if (--it == 0) return NAN;
    }
    const double uSq = c2A * (a * a - b * b) / (b * b);
    const double A = 1.0 + uSq / 16384.0 * (4096.0 + uSq * (-768.0 + uSq * (320.0 - 175.0 * uSq)));
    const double B = uSq / 1024.0 * (256.0 + uSq * (-128.0 + uSq * (74.0 - 47.0 * uSq)));
    const double dS = B * sS * (c2SM + B / 4.0 * (cS * (-1.0 + 2.0 * c2SM * c2SM) - B / 6.0 * c2SM * (-3.0 + 4.0 * sS * sS) * (-3.0 + 4.0 * c2SM * c2SM)));
    return b * A * (sig - dS);
}
static double o_geodesic_seq(const double* xy, int64_t n, int32_t method) {
    double v = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i)
        v += method == GPK_GEODESIC_HAVERSINE ? o_haversine(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3])
             : method == GPK_GEODESIC_VINCENTY ? o_vincenty(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3])
                                               : k_geodesic_m(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3]);
    return v;
}
int32_t gpko_geodesic_length(const gpk_geoarrow_desc* a, int32_t method, double* out) {
    if (method != GPK_GEODESIC_HAVERSINE && method != GPK_GEODESIC_VINCENTY && method != GPK_GEODESIC_KARNEY) return GPK_ERR_INVALID_ARGUMENT;
    for (int64_t g = 0; g < a->n_geoms; ++g) {
        double v = 0.0;
        if (!is_valid_row(a, g)) {
            out[g] = NAN;
            continue;
        }
        switch (a->geom_type) {
        case GPK_GEOM_LINESTRING:
            v = o_geodesic_seq(a->xy + 2 * (int64_t)a->geom_offsets[g], a->geom_offsets[g + 1] - a->geom_offsets[g], method);
            break;
        case GPK_GEOM_MULTILINESTRING:
            for (int64_t l = a->geom_offsets[g]; l < a->geom_offsets[g + 1]; ++l) {
                int64_t n;
                const double* xy = ring_xy(a, l, &n);
                v += o_geodesic_seq(xy, n, method);
            }
            break;
        case GPK_GEOM_POLYGON:
        case GPK_GEOM_MULTIPOLYGON: {
            int64_t p0, p1;
            geom_parts(a, g, &p0, &p1);
            for (int64_t p = p0; p < p1; ++p) {
                ring_span s = part_rings(a, p);
                if (s.r1 > s.r0) {
                    int64_t n;
                    const double* xy = ring_xy(a, s.r0, &n);
                    v += o_geodesic_seq(xy, n, method);
                }
            }
            break;
        }
        default:
            v = 0.0;
        }
        out[g] = v;
    }
    return GPK_OK;
}

/* ================================================================================================
 * simplify (geoseries.rs:108-116) — geo 0.27 algorithm/simplify.rs, compute_rdp restated RECURSIVELY, as upstream
 * writes it: the farthest interior point from the chord (fold with `>=`: the last one among equals), split when it is
 * farther than epsilon (left part first; `simplified_len` is shared), otherwise cull the interior unless the sequence
 * would drop below INITIAL_MIN (2 linestrings, 4 polygon rings).  The distance is line_segment_distance with a plain
 * sqrt (upstream: hypot; at most an ulp apart).  The HIP kernel walks the same recursion with an explicit stack.
 * ================================================================================================ */
static double o_seg_dist(const double* p, const double* s, const double* e) {
    const double dx = e[0] - s[0], dy = e[1] - s[1];
    if (s[0] == e[0] && s[1] == e[1]) return sqrt((p[0] - s[0]) * (p[0] - s[0]) + (p[1] - s[1]) * (p[1] - s[1]));
    const double d2 = dx * dx + dy * dy;
    const double r = ((p[0] - s[0]) * dx + (p[1] - s[1]) * dy) / d2;
    if (r <= 0.0) return sqrt((p[0] - s[0]) * (p[0] - s[0]) + (p[1] - s[1]) * (p[1] - s[1]));
    if (r >= 1.0) return sqrt((p[0] - e[0]) * (p[0] - e[0]) + (p[1] - e[1]) * (p[1] - e[1]));
    const double t = ((s[1] - p[1]) * dx - (s[0] - p[0]) * dy) / d2;
    return fabs(t) * sqrt(d2);
}
static void o_rdp(const double* xy, int64_t i, int64_t j, double eps, int64_t min_pts, int64_t* len, uint8_t* keep) {
    if (j - i < 2) return;
    int64_t at = 0;
    double best = 0.0;
    for (int64_t k = i + 1; k < j; ++k) {
        const double d = o_seg_dist(xy + 2 * k, xy + 2 * i, xy + 2 * j);
        if (d >= best) {
            best = d;
            at = k;
        }
    }
    if (best > eps) {
        o_rdp(xy, i, at, eps, min_pts, len, keep);
        o_rdp(xy, at, j, eps, min_pts, len, keep);
        return;
    }
    const int64_t culled = j - i - 1;
    if (*len - culled < min_pts) return;
    *len -= culled;
    for (int64_t k = i + 1; k < j; ++k) keep[k] = 0;
}
int32_t gpko_simplify(const gpk_geoarrow_desc* a, double eps, double* out_xy, int32_t* out_seq_offsets, int64_t* n_out) {
    const int32_t* off;
    int64_t n_seq;
    int64_t min_pts = 2;
    switch (a->geom_type) {
    case GPK_GEOM_LINESTRING:
        off = a->geom_offsets;
        n_seq = a->n_geoms;
        break;
    case GPK_GEOM_MULTILINESTRING:
        off = a->ring_offsets;
        n_seq = a->n_rings;
        break;
    case GPK_GEOM_POLYGON:
    case GPK_GEOM_MULTIPOLYGON:
        off = a->ring_offsets;
        n_seq = a->n_rings;
        min_pts = 4;
        break;
    default:
        return GPK_ERR_MISMATCHED_GEOMETRY;
    }
    int64_t w = 0;
    out_seq_offsets[0] = 0;
    for (int64_t s = 0; s < n_seq; ++s) {
        const int64_t c0 = off[s], n = off[s + 1] - c0;
        const double* xy = a->xy + 2 * c0;
        uint8_t* keep = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
        memset(keep, 1, (size_t)(n > 0 ? n : 1));
        int64_t len = n;
        if (n >= 3 && eps > 0.0) o_rdp(xy, 0, n - 1, eps, min_pts, &len, keep);
        for (int64_t k = 0; k < n; ++k)
            if (keep[k]) {
                out_xy[2 * w] = xy[2 * k];
                out_xy[2 * w + 1] = xy[2 * k + 1];
                ++w;
            }
        free(keep);
        out_seq_offsets[s + 1] = (int32_t)w;
    }
    *n_out = w;
    return GPK_OK;
}
