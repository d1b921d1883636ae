"""Exact-arithmetic pins for the oracle's operators that nothing in the reference pins (area, centroid, convex hull,
intersects(polygon, polygon), line.contains(point), point-segment distance): random INTEGER geometries, every expected
value computed with Python integers / fractions.Fraction by an independent brute-force restatement.  CPU only."""
import math
from fractions import Fraction

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from geopolars_amd import _abi
from geopolars_amd.geoarrow import GeoArrowArray

from .lattice import concentric_pair, nudged, random_pair, star, star_with_hole

F = Fraction
coord = st.integers(-50, 50)
point = st.tuples(coord, coord)


def orient(a, b, c) -> int:
    d = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    return (d > 0) - (d < 0)


def on_segment(a, b, p) -> bool:
    return orient(a, b, p) == 0 and min(a[0], b[0]) <= p[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= p[1] <= max(a[1], b[1])


def seg_intersect(a, b, c, d) -> bool:
    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    if o1 != o2 and o3 != o4:
        return True
    return on_segment(a, b, c) or on_segment(a, b, d) or on_segment(c, d, a) or on_segment(c, d, b)


stars = st.builds(star, coord, coord, st.lists(st.integers(1, 12), min_size=8, max_size=8))


def shoelace2(ring):
    return sum(ring[i][0] * ring[(i + 1) % len(ring)][1] - ring[(i + 1) % len(ring)][0] * ring[i][1] for i in range(len(ring)))


def in_or_on(ring, p) -> int:
    """-1 outside, 0 on the boundary, 1 inside (even-odd, integer arithmetic)"""
    n = len(ring)
    inside = False
    for i in range(n):
        a, b = ring[i], ring[(i + 1) % n]
        if on_segment(a, b, p):
            return 0
        if (a[1] > p[1]) != (b[1] > p[1]):
            # x of the crossing compared with p.x without division
            t = (b[0] - a[0]) * (p[1] - a[1]) - (p[0] - a[0]) * (b[1] - a[1])
            if (t > 0) == (b[1] > a[1]):
                inside = not inside
    return 1 if inside else -1


@settings(max_examples=150, deadline=None)
@given(st.lists(stars, min_size=1, max_size=6))
def test_area_and_centroid_match_rational_shoelace(oracle, rings):
    a = GeoArrowArray.from_polygons([[r] for r in rings])
    area = oracle.area(a)
    signed = oracle.area(a, signed=True)
    c, valid = oracle.centroid(a)
    for i, r in enumerate(rings):
        s2 = shoelace2(r)
        assert signed[i] == s2 / 2 and area[i] == abs(s2) / 2  # halves of small integers: exact in binary64
        assert s2 != 0
        cx = F(sum((r[j][0] + r[(j + 1) % 8][0]) * (r[j][0] * r[(j + 1) % 8][1] - r[(j + 1) % 8][0] * r[j][1]) for j in range(8)), 3 * s2)
        cy = F(sum((r[j][1] + r[(j + 1) % 8][1]) * (r[j][0] * r[(j + 1) % 8][1] - r[(j + 1) % 8][0] * r[j][1]) for j in range(8)), 3 * s2)
        assert abs(c[i, 0] - float(cx)) <= 1e-12 * max(1.0, abs(float(cx))) and abs(c[i, 1] - float(cy)) <= 1e-12 * max(1.0, abs(float(cy)))


@settings(max_examples=150, deadline=None)
@given(st.lists(point, min_size=1, max_size=30))
def test_convex_hull_matches_bruteforce(oracle, pts):
    a = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.array(pts, dtype=np.float64), geom_offsets=np.array([0, len(pts)], np.int32))
    xy, off = oracle.convex_hull(a)
    hull = [tuple(int(v) for v in p) for p in xy[off[0] : off[1]]]
    distinct = sorted(set(pts))
    assert hull[0] == hull[-1] == distinct[0]  # closed, starts at the lexicographic minimum
    hv = hull[:-1]
    if len(distinct) == 1:
        assert hv == distinct
        return
    # brute force: a point is a hull vertex iff it is not on a segment between two others and not strictly inside
    def is_vertex(p):
        others = [q for q in distinct if q != p]
        for i, q in enumerate(others):
            for r in others[i + 1 :]:
                if on_segment(q, r, p):
                    return False
        # strictly inside some triangle of other points -> not a vertex
        for i, q in enumerate(others):
            for j, r in enumerate(others[i + 1 :], i + 1):
                for t in others[j + 1 :]:
                    o = (orient(q, r, p), orient(r, t, p), orient(t, q, p))
                    if all(x > 0 for x in o) or all(x < 0 for x in o):
                        return False
        return True

    expected = {p for p in distinct if is_vertex(p)}
    collinear = all(orient(distinct[0], distinct[-1], p) == 0 for p in distinct)
    if collinear:
        assert hv == [distinct[0], distinct[-1]]
        return
    assert set(hv) == expected and len(hv) == len(expected)
    assert all(orient(hv[i], hv[(i + 1) % len(hv)], hv[(i + 2) % len(hv)]) > 0 for i in range(len(hv)))  # strictly convex, counter-clockwise


@settings(max_examples=200, deadline=None)
@given(stars, stars)
def test_polygon_intersects_polygon_matches_bruteforce(oracle, ra, rb):
    a, b = GeoArrowArray.from_polygons([[ra]]), GeoArrowArray.from_polygons([[rb]])
    got = bool(oracle.predicate_rowwise(a, b, "intersects")[0])
    crossing = any(seg_intersect(ra[i], ra[(i + 1) % 8], rb[j], rb[(j + 1) % 8]) for i in range(8) for j in range(8))
    contained = any(in_or_on(ra, p) >= 0 for p in rb) or any(in_or_on(rb, p) >= 0 for p in ra)
    assert got == (crossing or contained)


@settings(max_examples=200, deadline=None)
@given(st.lists(point, min_size=1, max_size=8), point)
def test_linestring_contains_point_matches_integer_arithmetic(oracle, line, p):
    ls = GeoArrowArray.from_linestrings([line])
    pt = GeoArrowArray.from_points([p])
    pairs, counts, _ = oracle.spatial_join(pt, ls, "intersects", mode=0)
    got = bool(counts[0])
    closed = line[0] == line[-1]
    if p == line[0] or p == line[-1]:
        exp = closed  # an end point belongs to the boundary unless the linestring is closed (geo 0.27 contains/line_string.rs)
    else:
        exp = any(on_segment(line[i], line[i + 1], p) for i in range(len(line) - 1))
    assert got == exp
    back, bc, _ = oracle.spatial_join(ls, pt, "contains", mode=0)  # the same arm with the line on the left
    assert bool(bc[0]) == exp


@settings(max_examples=200, deadline=None)
@given(point, point, point)
def test_point_segment_distance_matches_rational(oracle, p, a, b):
    ls = GeoArrowArray.from_linestrings([[a, b]])
    got = float(oracle.distance_rowwise(GeoArrowArray.from_points([p]), ls)[0])
    ab = (b[0] - a[0], b[1] - a[1])
    ap = (p[0] - a[0], p[1] - a[1])
    d2 = ab[0] ** 2 + ab[1] ** 2
    if d2 == 0:
        exp2 = F(ap[0] ** 2 + ap[1] ** 2)
    else:
        t = F(ap[0] * ab[0] + ap[1] * ab[1], d2)
        t = min(max(t, F(0)), F(1))
        exp2 = (F(ap[0]) - t * ab[0]) ** 2 + (F(ap[1]) - t * ab[1]) ** 2
    exp = math.sqrt(float(exp2))
    assert (exp2 == 0) == (got == 0.0)  # zero / non-zero is exact
    assert abs(got - exp) <= 1e-12 * max(exp, 1e-300)


# ---- contains(polygon, polygon): the set statement "B is not empty and B is a subset of A" -----------------------
def _closed(ring):
    return list(ring) + [ring[0]]


def _edges(ring):
    r = _closed(ring)
    return [(r[i], r[i + 1]) for i in range(len(ring)) if r[i] != r[i + 1]]


def _cross(u, v):
    return u[0] * v[1] - u[1] * v[0]


def _split_params(p, q, rings):
    """parameters t in [0, 1] at which segment pq meets an edge of one of the rings (ends of collinear overlaps included)"""
    d = (q[0] - p[0], q[1] - p[1])
    ts = {F(0), F(1)}
    for ring in rings:
        for a, b in _edges(ring):
            e = (b[0] - a[0], b[1] - a[1])
            ap = (a[0] - p[0], a[1] - p[1])
            den = _cross(d, e)
            if den != 0:
                t, u = F(_cross(ap, e), den), F(_cross(ap, d), den)
                if 0 <= t <= 1 and 0 <= u <= 1:
                    ts.add(t)
            elif _cross(ap, d) == 0:
                dd = d[0] * d[0] + d[1] * d[1]
                for c in (a, b):
                    t = F((c[0] - p[0]) * d[0] + (c[1] - p[1]) * d[1], dd)
                    if 0 <= t <= 1:
                        ts.add(t)
    return sorted(ts)


def _pieces(ring, others):
    """mid points of the pieces into which the rings `others` cut the edges of `ring` (rational coordinates)"""
    out = []
    for p, q in _edges(ring):
        ts = _split_params(p, q, others)
        for t0, t1 in zip(ts, ts[1:]):
            t = (t0 + t1) / 2
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def _poly_pos(poly, p) -> int:
    """-1 outside, 0 boundary, 1 inside for polygon = [exterior, hole, ...]"""
    e = in_or_on(poly[0], p)
    if e <= 0:
        return e
    for h in poly[1:]:
        k = in_or_on(h, p)
        if k == 0:
            return 0
        if k > 0:
            return -1
    return 1


def intersects_bruteforce(pa, pb) -> bool:
    """the closed sets meet: two boundary segments meet, or a vertex of one polygon is not outside the other (with holes)"""
    ea = [e for ring in pa for e in _edges(ring)]
    eb = [e for ring in pb for e in _edges(ring)]
    if any(seg_intersect(a0, a1, b0, b1) for a0, a1 in ea for b0, b1 in eb):
        return True
    return any(_poly_pos(pa, v) >= 0 for ring in pb for v in ring) or any(_poly_pos(pb, v) >= 0 for ring in pa for v in ring)


def contains_bruteforce(pa, pb) -> bool:
    # (1) every vertex and every piece of the boundary of B lies in A (closed)
    for ring in pb:
        if any(_poly_pos(pa, v) < 0 for v in ring):
            return False
        if any(_poly_pos(pa, m) < 0 for m in _pieces(ring, pa)):
            return False
    # (2) no hole of A lies in B: look at the pieces of the hole's boundary
    for hole in pa[1:]:
        mids = _pieces(hole, pb)
        pos = [_poly_pos(pb, m) for m in mids]
        if any(x > 0 for x in pos):
            return False  # a piece of the hole boundary in the interior of B: B covers points of the hole
        if all(x == 0 for x in pos) and all(in_or_on(pb[0], m) == 0 for m in mids):
            return False  # the hole IS the exterior ring of B: B fills the hole
    return True


small = st.integers(-6, 6)
polys_with_holes = st.builds(
    star_with_hole,
    small,
    small,
    st.lists(st.integers(1, 10), min_size=8, max_size=8),
    st.one_of(st.none(), st.lists(st.integers(1, 9), min_size=8, max_size=8)),
    st.one_of(st.none(), st.integers(0, 7)),
)


@settings(max_examples=600, deadline=None)
@given(polys_with_holes, polys_with_holes)
def test_polygon_contains_polygon_matches_rational_set_statement(oracle, pa, pb):
    a, b = GeoArrowArray.from_polygons([pa]), GeoArrowArray.from_polygons([pb])
    got = bool(oracle.predicate_rowwise(a, b, "contains")[0])
    assert got == contains_bruteforce(pa, pb)
    assert bool(oracle.predicate_rowwise(b, a, "within")[0]) == got
    if got:  # a subset with an interior intersects
        assert bool(oracle.predicate_rowwise(a, b, "intersects")[0])


def test_polygon_contains_polygon_known_cases(oracle):
    sq = lambda x0, y0, x1, y1: [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    cw = lambda r: list(reversed(r))
    big, inner, hole = sq(0, 0, 10, 10), sq(2, 2, 4, 4), sq(3, 3, 7, 7)
    u_shape = [(0, 0), (10, 0), (10, 10), (7, 10), (7, 3), (3, 3), (3, 10), (0, 10)]
    cases = [
        ([big], [inner], True),
        ([big], [big], True),  # equal polygons: relate says contains
        ([cw(big)], [inner], True),  # ring orientation of either operand is irrelevant
        ([big], [cw(inner)], True),
        ([inner], [big], False),
        ([big], [sq(0, 0, 5, 5)], True),  # shares two boundary edges
        ([big], [sq(5, 5, 12, 8)], False),  # pokes out
        ([big], [sq(10, 0, 12, 5)], False),  # outside, shares an edge
        ([big, hole], [inner], False),  # overlaps the hole
        ([big, hole], [sq(3, 3, 7, 7)], False),  # fills the hole exactly
        ([big, hole], [sq(0, 0, 3, 3)], True),  # touches the hole at one corner
        ([big, hole], [sq(1, 1, 9, 9)], False),  # swallows the hole
        ([big, hole], [sq(1, 1, 9, 9), sq(2, 2, 8, 8)], True),  # ... unless its own hole covers it
        ([big, hole], [sq(1, 1, 9, 9), hole], True),  # ... exactly
        ([big, hole], [sq(1, 1, 9, 9), sq(4, 4, 6, 6)], False),  # own hole too small
        ([u_shape], [[(3, 10), (7, 10), (5, 12)]], False),
        ([u_shape], [[(0, 10), (3, 10), (3, 3), (7, 3), (7, 10), (10, 10), (10, 0), (0, 0)]], True),
        ([u_shape], [[(1, 9), (9, 9), (9, 1), (1, 1)]], False),  # edge crosses the notch between two touch points
        ([u_shape], [[(3, 10), (7, 10), (7, 3), (3, 3)]], False),  # fills the notch: boundary in A, interior outside
        ([u_shape], [[(0, 10), (3, 10), (10, 10), (10, 0), (0, 0)]], False),  # closes the notch with a collinear edge
    ]
    for pa, pb, exp in cases:
        a, b = GeoArrowArray.from_polygons([pa]), GeoArrowArray.from_polygons([pb])
        assert contains_bruteforce(pa, pb) == exp, (pa, pb)
        assert bool(oracle.predicate_rowwise(a, b, "contains")[0]) == exp, (pa, pb)
    # multipolygon operands: a member must hold the whole right polygon; a multipolygon right side needs every member held
    two = GeoArrowArray.from_multipolygons([[[sq(0, 0, 4, 4)], [sq(6, 0, 10, 4)]]])
    assert bool(oracle.predicate_rowwise(two, GeoArrowArray.from_polygons([[sq(7, 1, 9, 3)]]), "contains")[0])
    assert not bool(oracle.predicate_rowwise(two, GeoArrowArray.from_polygons([[sq(3, 1, 7, 3)]]), "contains")[0])
    assert bool(oracle.predicate_rowwise(GeoArrowArray.from_polygons([[big]]), two, "contains")[0])
    assert not bool(oracle.predicate_rowwise(GeoArrowArray.from_polygons([[sq(0, 0, 5, 5)]]), two, "contains")[0])
    # empty and invalid operands are never contained / never contain
    empty = GeoArrowArray.from_polygons([[]])
    assert not bool(oracle.predicate_rowwise(GeoArrowArray.from_polygons([[big]]), empty, "contains")[0])
    assert not bool(oracle.predicate_rowwise(empty, GeoArrowArray.from_polygons([[big]]), "contains")[0])


def test_polygon_contains_polygon_concentric_rings(oracle):
    import random

    rng = random.Random(20241008)
    pairs = [concentric_pair(rng) for _ in range(1500)]
    a = GeoArrowArray.from_polygons([p for p, _ in pairs])
    b = GeoArrowArray.from_polygons([q for _, q in pairs])
    got = oracle.predicate_rowwise(a, b, "contains").astype(bool)
    exp = np.array([contains_bruteforce(p, q) for p, q in pairs])
    assert 300 < exp.sum() < 1200  # both answers are well represented
    assert np.array_equal(got, exp)


def test_polygon_contains_polygon_random_neighbours(oracle):
    import random

    rng = random.Random(5)
    pairs = [random_pair(rng) for _ in range(3000)]
    a = GeoArrowArray.from_polygons([p for p, _ in pairs])
    b = GeoArrowArray.from_polygons([q for _, q in pairs])
    got = oracle.predicate_rowwise(a, b, "contains").astype(bool)
    exp = np.array([contains_bruteforce(p, q) for p, q in pairs])
    assert 50 < exp.sum() < 1500
    assert np.array_equal(got, exp)


# ---- more unpinned operators against exact / high-precision restatements -------------------------------------------------
def _ring_moments(ring):
    """(2*signed area, 6*area*cx, 6*area*cy) of a ring in exact integers"""
    n = len(ring)
    a2 = mx = my = 0
    for i in range(n):
        (x0, y0), (x1, y1) = ring[i], ring[(i + 1) % n]
        w = x0 * y1 - x1 * y0
        a2 += w
        mx += (x0 + x1) * w
        my += (y0 + y1) * w
    return a2, mx, my


@settings(max_examples=200, deadline=None)
@given(st.lists(polys_with_holes, min_size=1, max_size=4))
def test_area_and_centroid_of_polygons_with_holes_and_multipolygons(oracle, polys):
    """area = |exterior| - sum |holes|; centroid = area-weighted mean of the rings' centroids with holes negative
    (geo 0.27 area.rs / centroid.rs), rational arithmetic; the multipolygon of the same members adds the members up"""

    def poly_stats(p):
        a2 = mx = my = F(0)
        for k, ring in enumerate(p):
            r2, rx, ry = _ring_moments(ring)
            s = 1 if r2 > 0 else -1  # orientation-free: work with |ring area| and its centroid
            sign = 1 if k == 0 else -1
            a2 += sign * s * r2
            mx += sign * s * rx
            my += sign * s * ry
        return a2, mx, my

    a = GeoArrowArray.from_polygons(polys)
    area = oracle.area(a)
    c, valid = oracle.centroid(a)
    tot = [F(0), F(0), F(0)]
    for i, p in enumerate(polys):
        a2, mx, my = poly_stats(p)
        assert a2 > 0 and valid[i]
        assert area[i] == float(a2 / 2)
        assert abs(c[i, 0] - float(mx / (3 * a2))) <= 1e-12 * max(1.0, abs(float(mx / (3 * a2))))
        assert abs(c[i, 1] - float(my / (3 * a2))) <= 1e-12 * max(1.0, abs(float(my / (3 * a2))))
        tot = [tot[0] + a2, tot[1] + mx, tot[2] + my]
    m = GeoArrowArray.from_multipolygons([[p for p in polys]])
    assert oracle.area(m)[0] == float(tot[0] / 2)
    cm, vm = oracle.centroid(m)
    assert vm[0]
    assert abs(cm[0, 0] - float(tot[1] / (3 * tot[0]))) <= 1e-12 * max(1.0, abs(float(tot[1] / (3 * tot[0]))))
    assert abs(cm[0, 1] - float(tot[2] / (3 * tot[0]))) <= 1e-12 * max(1.0, abs(float(tot[2] / (3 * tot[0]))))


@settings(max_examples=200, deadline=None)
@given(polys_with_holes, st.lists(point, min_size=1, max_size=12))
def test_point_predicates_and_distance_against_polygon_with_hole(oracle, poly, pts):
    """contains = strictly inside; intersects = inside or on the boundary; within(point, polygon) = contains;
    distance = 0 unless outside (or in the hole), then the nearest boundary segment (Appendix A.4)"""
    n = len(pts)
    a = GeoArrowArray.from_polygons([poly] * n)
    p = GeoArrowArray.from_points(pts)
    pos = [_poly_pos(poly, q) for q in pts]
    assert oracle.predicate_rowwise(a, p, "contains").tolist() == [x > 0 for x in pos]
    assert oracle.predicate_rowwise(p, a, "within").tolist() == [x > 0 for x in pos]
    assert oracle.predicate_rowwise(a, p, "intersects").tolist() == [x >= 0 for x in pos]
    assert oracle.predicate_rowwise(p, a, "intersects").tolist() == [x >= 0 for x in pos]
    d = oracle.distance_rowwise(p, a)
    for i, q in enumerate(pts):
        if pos[i] >= 0:
            assert d[i] == 0.0
            continue
        best = None
        for ring in poly:
            for s, e in _edges(ring):
                ab = (e[0] - s[0], e[1] - s[1])
                ap = (q[0] - s[0], q[1] - s[1])
                t = min(max(F(ap[0] * ab[0] + ap[1] * ab[1], ab[0] ** 2 + ab[1] ** 2), F(0)), F(1))
                d2 = (F(ap[0]) - t * ab[0]) ** 2 + (F(ap[1]) - t * ab[1]) ** 2
                best = d2 if best is None or d2 < best else best
        exp = math.sqrt(float(best))
        assert d[i] > 0.0 and abs(d[i] - exp) <= 1e-12 * exp


@settings(max_examples=200, deadline=None)
@given(st.lists(st.lists(point, min_size=0, max_size=9), min_size=1, max_size=5))
def test_length_centroid_and_bounds_of_linestrings_and_multipoints(oracle, lines):
    ls = GeoArrowArray.from_linestrings(lines)
    length = oracle.euclidean_length(ls)
    c, valid = oracle.centroid(ls)
    b = oracle.bounds(ls)
    for i, l in enumerate(lines):
        segs = [(l[k], l[k + 1]) for k in range(len(l) - 1)]
        lens = [math.hypot(e[0] - s[0], e[1] - s[1]) for s, e in segs]
        tot = math.fsum(lens)
        assert abs(length[i] - tot) <= 1e-12 * max(tot, 1e-300)
        if not l:
            assert not valid[i] and np.isnan(b[i]).all()
            continue
        assert b[i].tolist() == [min(x for x, _ in l), min(y for _, y in l), max(x for x, _ in l), max(y for _, y in l)]
        assert valid[i]
        if tot > 0:  # length-weighted mean of the segment mid points
            ex = math.fsum(w * (s[0] + e[0]) / 2 for w, (s, e) in zip(lens, segs)) / tot
            ey = math.fsum(w * (s[1] + e[1]) / 2 for w, (s, e) in zip(lens, segs)) / tot
        else:  # all coordinates equal (or one coordinate): the mean of the points
            ex, ey = float(F(sum(x for x, _ in l), len(l))), float(F(sum(y for _, y in l), len(l)))
        assert abs(c[i, 0] - ex) <= 1e-12 * max(1.0, abs(ex)) and abs(c[i, 1] - ey) <= 1e-12 * max(1.0, abs(ey))
    flat = [q for l in lines for q in l]
    if flat:
        mp = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.array(flat, dtype=np.float64), geom_offsets=np.array([0, len(flat)], np.int32))
        cm, vm = oracle.centroid(mp)
        assert vm[0]
        assert abs(cm[0, 0] - float(F(sum(x for x, _ in flat), len(flat)))) <= 1e-12 * 50
        assert abs(cm[0, 1] - float(F(sum(y for _, y in flat), len(flat)))) <= 1e-12 * 50


@settings(max_examples=100, deadline=None)
@given(stars, st.lists(st.integers(-8, 8), min_size=6, max_size=6))
def test_affine_transform_with_integer_matrices_is_exact(oracle, ring, m):
    """[a, b, xoff, d, e, yoff] (upstream order, geo 0.27 AffineTransform::new): small integers leave no rounding"""
    a = GeoArrowArray.from_polygons([[ring]])
    out = oracle.affine_transform(a, m)
    exp = [(m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]) for x, y in _closed(ring)]
    assert out.tolist() == [[float(x), float(y)] for x, y in exp]


def test_polygon_contains_polygon_on_inexact_floats(oracle):
    """the same pairs scaled by 0.1 (no longer exactly representable) and nudged by a few ulps: relations that were exact
    touches become hair-thin gaps or overlaps, which only exact orientations (adaptive expansion path) resolve; the brute
    force works on the doubles' exact rational values"""
    import random

    rng = random.Random(11)

    pairs = []
    for _ in range(700):
        pa, pb = concentric_pair(rng) if rng.random() < 0.6 else random_pair(rng)
        pairs.append((nudged(pa, rng), nudged(pb, rng)))
    a = GeoArrowArray.from_polygons([p for p, _ in pairs])
    b = GeoArrowArray.from_polygons([q for _, q in pairs])
    got = oracle.predicate_rowwise(a, b, "contains").astype(bool)
    exact = lambda poly: [[(F(x), F(y)) for x, y in ring] for ring in poly]
    exp = np.array([contains_bruteforce(exact(p), exact(q)) for p, q in pairs])
    assert 40 < exp.sum() < 600
    assert np.array_equal(got, exp)


def test_oracle_reproduces_the_contains_golden(oracle):
    """the committed brute-force answers (tests/golden/contains_lattice.npz) — the same file the GPU test reads"""
    from .lattice import load_contains_golden

    a, b, exp = load_contains_golden()
    assert len(exp) == 4000 and 300 < exp.sum() < 1000
    assert np.array_equal(oracle.predicate_rowwise(a, b, "contains").astype(bool), exp)
    assert np.array_equal(oracle.predicate_rowwise(b, a, "within").astype(bool), exp)
    exp_i = load_contains_golden(key="intersects")[2]
    assert exp_i.sum() > exp.sum() and not exp_i.all()
    assert np.array_equal(oracle.predicate_rowwise(a, b, "intersects").astype(bool), exp_i)
