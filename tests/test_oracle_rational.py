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


def star(cx, cy, radii):
    """simple polygon with integer vertices: one vertex per direction of a fixed fan of 8 integer directions"""
    dirs = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]
    return [(cx + r * dx, cy + r * dy) for r, (dx, dy) in zip(radii, dirs)]


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
