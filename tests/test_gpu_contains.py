"""GPU parity: contains(polygon, polygon) — the Polygon x Polygon / MultiPolygon x Polygon `contains` arms of the join
dispatch (spatial_index.rs:99-101,107-111) and the row-wise contains / within predicates — through the C ABI against
the CPU oracle, bit-exact.  The oracle's decisions are pinned to exact rational arithmetic in test_oracle_rational.py."""
import random

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import join_pairs

from .lattice import concentric_pair, nudged, random_pair

pytestmark = pytest.mark.gpu


def _lattice_columns(n_each: int, seed: int):
    rng = random.Random(seed)
    pairs = [concentric_pair(rng) for _ in range(n_each)] + [random_pair(rng) for _ in range(n_each)]
    return GeoArrowArray.from_polygons([p for p, _ in pairs]), GeoArrowArray.from_polygons([q for _, q in pairs])


def test_rowwise_contains_and_within_on_lattice_polygons(gpk, oracle):
    """touching, collinear and coincident boundaries, holes filled / covered / swallowed: every rule of gpk_contains.h"""
    a, b = _lattice_columns(4000, 11)
    exp = oracle.predicate_rowwise(a, b, "contains")
    assert 1000 < int(exp.sum()) < 5000
    sa, sb = GeoSeries(a), GeoSeries(b)
    assert np.array_equal(sa.contains(sb), exp.astype(bool))
    assert np.array_equal(sb.within(sa), exp.astype(bool))
    # the reversed question has its own answers (equal polygons contain each other)
    rev = oracle.predicate_rowwise(b, a, "contains")
    assert np.array_equal(sb.contains(sa), rev.astype(bool))
    # ring orientation of either side is irrelevant: reverse every ring of a
    flipped = a.xy.copy()
    for r in range(len(a.ring_offsets) - 1):
        c0, c1 = a.ring_offsets[r], a.ring_offsets[r + 1]
        flipped[c0:c1] = a.xy[c0:c1][::-1]
    a2 = GeoArrowArray(a.geom_type, flipped, geom_offsets=a.geom_offsets, ring_offsets=a.ring_offsets)
    assert np.array_equal(GeoSeries(a2).contains(sb), exp.astype(bool))


def test_rowwise_contains_is_invariant_under_exact_transforms(gpk, oracle):
    a, b = _lattice_columns(1500, 12)
    exp = oracle.predicate_rowwise(a, b, "contains").astype(bool)
    for scale, dx, dy in ((0.125, 3.0, -7.0), (1024.0, -1.0e6, 2.0e6)):  # powers of two and integers: exact in binary64
        ta = GeoArrowArray(a.geom_type, a.xy * scale + np.array([dx, dy]), geom_offsets=a.geom_offsets, ring_offsets=a.ring_offsets)
        tb = GeoArrowArray(b.geom_type, b.xy * scale + np.array([dx, dy]), geom_offsets=b.geom_offsets, ring_offsets=b.ring_offsets)
        assert np.array_equal(GeoSeries(ta).contains(GeoSeries(tb)), exp)


def test_contains_join_small_polygons_in_big_ones(gpk, oracle):
    """big 64-vertex cells on the left, 20k small polygons on the right: inside, straddling and outside"""
    big = synth.star_polygons(100, 64)
    small = synth.clustered_polygons(20_000, seed=41, mean_neighbours=0.5)
    exp_pairs, exp_counts, _ = oracle.spatial_join(big, small, "contains", mode=1)
    assert 2000 < len(exp_pairs) < 20_000
    got_pairs, got_counts = join_pairs(GeoSeries(big), GeoSeries(small), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
    # contained implies intersecting
    ipairs, _ = join_pairs(GeoSeries(big), GeoSeries(small), "intersects")
    assert set(map(tuple, got_pairs.tolist())) <= set(map(tuple, ipairs.tolist()))


def test_contains_join_multipolygons_with_holes_on_the_left(gpk, oracle):
    """MultiPolygon x Polygon (spatial_index.rs:107-111): power-law members, some with a hole"""
    left = synth.powerlaw_multipolygons(3000, seed=51)
    right = synth.clustered_polygons(30_000, seed=52, mean_neighbours=0.05, min_verts=4, max_verts=12)
    exp_pairs, exp_counts, _ = oracle.spatial_join(left, right, "contains", mode=1)
    assert len(exp_pairs) > 200
    got_pairs, got_counts = join_pairs(GeoSeries(left), GeoSeries(right), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)


def test_contains_join_on_lattice_polygons_all_pairs(gpk, oracle):
    """every polygon of one lattice column against every polygon of the other: the join and the brute-force oracle"""
    a, b = _lattice_columns(150, 13)
    exp_pairs, exp_counts, _ = oracle.spatial_join(a, b, "contains", mode=0)
    assert len(exp_pairs) > 1000
    got_pairs, got_counts = join_pairs(GeoSeries(a), GeoSeries(b), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)


def test_contains_arms_without_an_upstream_match_are_empty(gpk, oracle):
    """`_ => false` (spatial_index.rs:136): contains with a MultiPolygon on the right, within for polygonal pairs"""
    polys = synth.star_polygons(30, 16)
    multi = synth.powerlaw_multipolygons(40, seed=3)
    for l, r, pred in ((polys, multi, "contains"), (multi, multi, "contains"), (polys, polys, "within"), (multi, polys, "within")):
        exp_pairs, exp_counts, _ = oracle.spatial_join(l, r, pred, mode=0)
        assert len(exp_pairs) == 0
        pairs, counts = join_pairs(GeoSeries(l), GeoSeries(r), pred)
        assert pairs.shape == (0, 2) and not counts.any()
    # the arm that exists: a polygon contains itself
    pairs, counts = join_pairs(GeoSeries(polys), GeoSeries(polys), "contains")
    assert pairs.tolist() == [[i, i] for i in range(30)]


def test_rowwise_contains_with_multipolygon_operands_nulls_and_empties(gpk, oracle):
    sq = lambda x0, y0, x1, y1: [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    two = [[sq(0, 0, 4, 4)], [sq(6, 0, 10, 4), sq(7, 1, 8, 2)]]
    left = GeoArrowArray.from_multipolygons([two, two, two, two, [[sq(0, 0, 10, 10)]], [], two])
    right = GeoArrowArray.from_multipolygons(
        [[[sq(1, 1, 3, 3)]], [[sq(3, 1, 7, 3)]], [[sq(6.5, 0.5, 9, 3)]], [[sq(8.5, 0.5, 9.5, 3.5)], [sq(1, 1, 2, 2)]], two, [[sq(0, 0, 1, 1)]], []]
    )
    exp = oracle.predicate_rowwise(left, right, "contains").astype(bool)
    assert exp.tolist() == [True, False, False, True, True, False, False]
    assert np.array_equal(GeoSeries(left).contains(GeoSeries(right)), exp)
    assert np.array_equal(GeoSeries(right).within(GeoSeries(left)), exp)
    # null rows never match
    validity = np.array([0b1111110], dtype=np.uint8)
    ln = GeoArrowArray(left.geom_type, left.xy, geom_offsets=left.geom_offsets, part_offsets=left.part_offsets, ring_offsets=left.ring_offsets, validity=validity)
    expn = oracle.predicate_rowwise(ln, right, "contains").astype(bool)
    assert exp[0] and not expn[0] and np.array_equal(GeoSeries(ln).contains(GeoSeries(right)), expn)


def test_rowwise_linestring_contains_point(gpk, oracle):
    """contains(linestring, point) / within(point, linestring), row-wise: Contains<Coord> of geo 0.27"""
    lines = synth.random_linestrings(5000, seed=61)
    rng = np.random.default_rng(62)
    pts = np.empty((5000, 2))
    for i in range(5000):  # a vertex, an edge midpoint-ish lattice point, an end point or a random point
        c0, c1 = lines.geom_offsets[i], lines.geom_offsets[i + 1]
        k = rng.integers(0, 4)
        if k == 0:
            pts[i] = lines.xy[rng.integers(c0, c1)]
        elif k == 1:
            pts[i] = lines.xy[c0]
        elif k == 2:
            j = rng.integers(c0, c1 - 1)
            pts[i] = (lines.xy[j] + lines.xy[j + 1]) / 2  # on the segment only when the halving is exact
        else:
            pts[i] = rng.uniform(0, 1000, 2)
    p = GeoArrowArray.from_points(pts)
    exp = oracle.predicate_rowwise(lines, p, "contains").astype(bool)
    assert exp.any() and not exp.all()
    assert np.array_equal(GeoSeries(lines).contains(GeoSeries(p)), exp)
    assert np.array_equal(GeoSeries(p).within(GeoSeries(lines)), exp)


def test_rowwise_contains_on_inexact_floats(gpk, oracle):
    """lattice pairs scaled by 0.1 and nudged by a few ulps: touches turn into hair-thin gaps and overlaps, the orientation
    filter gives up and the expansion path decides — on the GPU exactly as in the oracle (which is pinned to the doubles'
    rational values in test_oracle_rational.py)"""
    rng = random.Random(21)
    pairs = []
    for _ in range(6000):
        pa, pb = concentric_pair(rng) if rng.random() < 0.6 else random_pair(rng)
        pairs.append((nudged(pa, rng), nudged(pb, rng)))
    a = GeoArrowArray.from_polygons([p for p, _ in pairs])
    b = GeoArrowArray.from_polygons([q for _, q in pairs])
    exp = oracle.predicate_rowwise(a, b, "contains").astype(bool)
    assert 300 < exp.sum() < 5000
    assert np.array_equal(GeoSeries(a).contains(GeoSeries(b)), exp)
    assert np.array_equal(GeoSeries(b).within(GeoSeries(a)), exp)


def test_rowwise_contains_matches_the_rational_golden(gpk):
    """answers computed with Fraction arithmetic by tests/golden/make_contains_golden.py: no oracle between the GPU and
    the exact result"""
    from .lattice import load_contains_golden

    a, b, exp = load_contains_golden()
    assert np.array_equal(GeoSeries(a).contains(GeoSeries(b)), exp)
    assert np.array_equal(GeoSeries(b).within(GeoSeries(a)), exp)
    exp_i = load_contains_golden(key="intersects")[2]
    assert np.array_equal(GeoSeries(a).intersects(GeoSeries(b)), exp_i)
    assert np.array_equal(GeoSeries(b).intersects(GeoSeries(a)), exp_i)
