"""GPU parity for the streaming / row-wise operators through the C ABI vs the
CPU oracle.  Tolerances: f64 results within 1e-9 relative (north star); affine and bounds are
bit-exact (no reduction-order freedom); booleans bit-exact."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _close(got, exp, rtol=RTOL):
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape
    nan_g, nan_e = np.isnan(got), np.isnan(exp)
    assert np.array_equal(nan_g, nan_e)
    g, e = got[~nan_g], exp[~nan_e]
    scale = np.maximum(np.abs(e), 1e-300)
    bad = np.abs(g - e) > rtol * scale
    assert not bad.any(), f"max rel err {np.max(np.abs(g - e) / scale):.3e}"


def _ring(n, r, cx, cy, phase=0.0):
    t = phase + 2 * np.pi * np.arange(n) / n
    rr = r * (1.0 + 0.3 * np.sin(7 * t))
    return [(cx + rr[i] * np.cos(t[i]), cy + rr[i] * np.sin(t[i])) for i in range(n)]


def _ragged_polygons():
    """every size class of the streaming reductions in one array: rings of 3..16, 17..128, 129..512 vertices, one-chunk
    long rings, rings spanning several 8192-coordinate chunks (one with a long hole), a degenerate zero-area giant"""
    sizes = [3, 4, 15, 16, 17, 100, 128, 129, 400, 512, 513, 5000, 8191, 8192, 8193, 30000, 70001]
    polys = [[_ring(n, 10.0 + k, 100.0 * k, 50.0)] for k, n in enumerate(sizes)]
    polys.append([_ring(40000, 50.0, -500.0, -500.0), _ring(20000, 10.0, -500.0, -500.0, 0.5)[::-1]])
    line = [(float(i), 2.0 * i) for i in range(9000)]
    polys.append([line + line[-2::-1]])  # 17999-coordinate ring of zero area: length-weighted centroid, chunked
    return GeoArrowArray.from_polygons(polys)


def _giant_linestrings():
    rng = np.random.default_rng(12)
    return GeoArrowArray.from_linestrings([np.cumsum(rng.normal(size=(n, 2)), axis=0).tolist() for n in (2, 9, 600, 8192, 8193, 50000)])


def _arrays():
    return {
        "stars": synth.star_polygons(300, 64),
        "tri": synth.star_polygons(1000, 3),
        "clustered": synth.clustered_polygons(2000),
        "multipoly": synth.powerlaw_multipolygons(1500),
        "lines": synth.random_linestrings(800),
        "ragged": _ragged_polygons(),
        "giant_lines": _giant_linestrings(),
        "points": synth.uniform_points(5000),
        "holes": GeoArrowArray.from_polygons(
            [
                [[(0, 0), (10, 0), (10, 10), (0, 10)], [(2, 2), (2, 8), (8, 8), (8, 2)]],
                [[(0, 0), (0, 5), (5, 5), (5, 0)]],  # clockwise exterior
                [[(1, 1), (2, 2), (3, 3), (1, 1)]],  # zero-area ring -> linestring centroid
                [[(4, 4), (4, 4), (4, 4), (4, 4)]],  # all-identical ring -> point centroid
                [],  # empty polygon
            ]
        ),
    }


@pytest.mark.parametrize("name", list(_arrays()))
def test_area_length_bounds_centroid(gpk, oracle, name):
    a = _arrays()[name]
    s = GeoSeries(a)
    _close(s.area(), oracle.area(a))
    _close(s.signed_area(), oracle.area(a, signed=True))
    _close(s.euclidean_length(), oracle.euclidean_length(a))
    got_b, exp_b = s.bounds(), oracle.bounds(a)
    assert np.array_equal(got_b, exp_b, equal_nan=True)  # min/max: bit exact
    exp_c, exp_v = oracle.centroid(a)
    _close(s.centroid().array.xy, exp_c)


@pytest.mark.parametrize("name", ["stars", "multipoly", "lines", "points"])
def test_affine_bit_exact(gpk, oracle, name):
    a = _arrays()[name]
    m = [0.8660254037844387, -0.5, 12.25, 0.5, 0.8660254037844387, -3.125]
    got = GeoSeries(a).affine_transform(m).array.xy
    assert np.array_equal(got, oracle.affine_transform(a, m))
    t = GeoSeries(a).translate(10.0, 10.0).array.xy  # benches/affine.rs:25
    assert np.array_equal(t, oracle.affine_transform(a, [1, 0, 10.0, 0, 1, 10.0]))


def _canon(ring):
    ring = ring[:-1] if len(ring) > 1 and np.array_equal(ring[0], ring[-1]) else ring
    if len(ring) == 0:
        return ring
    k = np.lexsort((ring[:, 1], ring[:, 0]))[0]
    return np.roll(ring, -k, axis=0)


@pytest.mark.parametrize("name", ["stars", "multipoly", "lines", "holes"])
def test_convex_hull(gpk, oracle, name):
    a = _arrays()[name]
    exp_xy, exp_off = oracle.convex_hull(a)
    h = GeoSeries(a).convex_hull().array
    assert np.array_equal(h.ring_offsets, exp_off)
    for g in range(len(a)):
        got = _canon(h.xy[h.ring_offsets[g] : h.ring_offsets[g + 1]])
        exp = _canon(exp_xy[exp_off[g] : exp_off[g + 1]])
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("other", ["lines", "stars", "multipoly", "holes", "points"])
def test_distance_rowwise(gpk, oracle, other):
    b = _arrays()[other]
    pts = synth.uniform_points(20_000)
    rows = (np.arange(len(pts)) % len(b)).astype(np.uint32)
    exp = oracle.distance_rowwise(pts, b, rows)
    got = GeoSeries(pts).distance(GeoSeries(b), rows)
    _close(got, exp)
    shuffled = np.random.default_rng(5).permutation(rows)
    _close(GeoSeries(pts).distance(GeoSeries(b), shuffled), oracle.distance_rowwise(pts, b, shuffled))


def test_distance_points_on_lines(gpk, oracle):
    """vertices and (numerically) on-segment points: the f64::EPSILON short-circuit of
    line_string_contains_point must fire identically."""
    ls = synth.random_linestrings(200)
    off = ls.geom_offsets
    rows, pts = [], []
    for i in range(len(ls)):
        v = ls.xy[off[i] : off[i + 1]]
        pts += [v[0], v[-1], (v[0] + v[1]) / 2.0, v[1] + (v[2] - v[1]) * 0.25]
        rows += [i] * 4
    p = GeoArrowArray.from_points(np.array(pts))
    rows = np.array(rows, dtype=np.uint32)
    exp = oracle.distance_rowwise(p, ls, rows)
    got = GeoSeries(p).distance(GeoSeries(ls), rows)
    assert np.array_equal(got == 0.0, exp == 0.0)
    _close(got, exp)


def test_distance_swapped_and_identity(gpk, oracle):
    ls = synth.random_linestrings(3000)
    pts = synth.uniform_points(3000)
    exp = oracle.distance_rowwise(pts, ls)
    _close(GeoSeries(pts).distance(GeoSeries(ls)), exp)
    _close(GeoSeries(ls).distance(GeoSeries(pts)), exp)


@pytest.mark.parametrize("pred", ["contains", "within", "intersects"])
def test_predicates_rowwise_point_polygon(gpk, oracle, pred):
    polys = _arrays()["holes"]
    gx, gy = np.meshgrid(np.arange(-1, 12, 0.5), np.arange(-1, 12, 0.5))
    pts = GeoArrowArray.from_points(np.stack([gx.ravel(), gy.ravel()], axis=1))
    rows = (np.arange(len(pts)) % len(polys)).astype(np.uint32)
    if pred == "contains":  # contains(polygon_row, point): polygons on the left need equal lengths
        polys_rep = synth.star_polygons(len(pts), 9)
        exp = oracle.predicate_rowwise(polys_rep, pts, pred)
        got = GeoSeries(polys_rep).contains(GeoSeries(pts))
    else:
        exp = oracle.predicate_rowwise(pts, polys, pred, rows)
        got = getattr(GeoSeries(pts), pred)(GeoSeries(polys), rows)
    assert np.array_equal(got, exp)


def test_predicate_within_large(gpk, oracle):
    mp = synth.powerlaw_multipolygons(3000)
    c, _ = oracle.centroid(mp)
    # half the points sit at a member centroid's neighbourhood, half are random
    pts = synth.uniform_points(len(mp)).xy
    pts[::2] = c[::2]
    p = GeoArrowArray.from_points(pts)
    for pred in ("within", "intersects"):
        exp = oracle.predicate_rowwise(p, mp, pred)
        got = getattr(GeoSeries(p), pred)(GeoSeries(mp))
        assert exp.any()
        assert np.array_equal(got, exp)


def test_intersects_polygon_polygon(gpk, oracle):
    a = synth.clustered_polygons(4000, seed=11, mean_neighbours=40.0)
    b = synth.clustered_polygons(4000, seed=12, mean_neighbours=40.0)
    rows = np.random.default_rng(3).integers(0, len(b), len(a)).astype(np.uint32)
    # re-pair every other row with a bbox-overlapping partner so that both outcomes are well covered
    ba, bb = oracle.bounds(a), oracle.bounds(b)
    order = np.argsort(bb[:, 0])
    pos = np.searchsorted(bb[order, 0], ba[:, 0])
    near = order[np.clip(pos, 0, len(b) - 1)]
    rows[::2] = near[::2]
    exp = oracle.predicate_rowwise(a, b, "intersects", rows)
    assert 0.02 < exp.mean() < 0.98
    got = GeoSeries(a).intersects(GeoSeries(b), rows)
    assert np.array_equal(got, exp)
    same = GeoSeries(a).intersects(GeoSeries(a))
    assert same.all()
    nested = GeoArrowArray.from_polygons([[[(0, 0), (10, 0), (10, 10), (0, 10)]], [[(0, 0), (10, 0), (10, 10), (0, 10)]], [[(0, 0), (1, 0), (1, 1), (0, 1)]]])
    inner = GeoArrowArray.from_polygons([[[(4, 4), (5, 4), (5, 5), (4, 5)]], [[(10, 10), (11, 10), (11, 11), (10, 11)]], [[(2, 2), (3, 2), (3, 3), (2, 3)]]])
    assert GeoSeries(nested).intersects(GeoSeries(inner)).tolist() == [True, True, False]
    assert oracle.predicate_rowwise(nested, inner, "intersects").tolist() == [True, True, False]


def test_centroid_polygon_fully_covered_by_hole(gpk, oracle):
    """exterior area == hole area: the polygon degenerates to its exterior linestring (centroid.rs add_polygon)."""
    sq = [(0, 0), (4, 0), (4, 4), (0, 4)]
    a = GeoArrowArray.from_polygons([[sq, sq[::-1]], [sq]])
    exp, _ = oracle.centroid(a)
    _close(GeoSeries(a).centroid().array.xy, exp)
    assert exp.tolist() == [[2.0, 2.0], [2.0, 2.0]]


def _affine_rows_reference(a: GeoArrowArray, mats: np.ndarray, oracle) -> np.ndarray:
    """row-by-row application of the oracle's affine (bit-exact reference for per-geometry matrices)"""
    from geopolars_amd.dist import slice_rows

    out = np.empty_like(a.xy)
    w = 0
    for g in range(len(a)):
        row = slice_rows(a, g, g + 1)
        out[w : w + row.n_coords] = oracle.affine_transform(row, mats[g])
        w += row.n_coords
    return out


@pytest.mark.parametrize("name", ["holes", "multipoly", "lines"])
@pytest.mark.parametrize("origin", ["center", "centroid", (3.0, -2.0)])
def test_rotate_scale_skew_per_geometry_origin(gpk, oracle, name, origin):
    """geoseries.rs:85-139: rotate / scale / skew about centroid | bbox center | point == one affine matrix per row."""
    import math

    a = _arrays()[name]
    if name == "multipoly":
        from geopolars_amd.dist import slice_rows

        a = slice_rows(a, 0, 200)
    if name == "holes":  # drop the empty polygon: its origin is undefined (NaN) in both implementations
        a = GeoArrowArray.from_polygons([[[(0, 0), (10, 0), (10, 10), (0, 10)], [(2, 2), (2, 8), (8, 8), (8, 2)]], [[(0, 0), (0, 5), (5, 5), (5, 0)]]])
    s = GeoSeries(a)
    if origin == "center":
        b = oracle.bounds(a)
        o = np.stack([(b[:, 0] + b[:, 2]) / 2.0, (b[:, 1] + b[:, 3]) / 2.0], axis=1)
    elif origin == "centroid":
        o = s.centroid().array.xy  # the GPU centroid itself is checked elsewhere; matrices must match bit for bit
    else:
        o = np.tile(np.array([origin], dtype=np.float64), (len(a), 1))
    z = np.zeros(len(a))
    t = math.radians(30.0)
    c, sn = math.cos(t), math.sin(t)
    rot = np.stack([z + c, z - sn, o[:, 0] - c * o[:, 0] + sn * o[:, 1], z + sn, z + c, o[:, 1] - sn * o[:, 0] - c * o[:, 1]], axis=1)
    assert np.array_equal(s.rotate(30.0, origin).array.xy, _affine_rows_reference(a, rot, oracle))
    sc = np.stack([z + 2.0, z, o[:, 0] * (1 - 2.0), z, z + 0.5, o[:, 1] * (1 - 0.5)], axis=1)
    assert np.array_equal(s.scale(2.0, 0.5, origin).array.xy, _affine_rows_reference(a, sc, oracle))
    tx, ty = math.tan(math.radians(10.0)), math.tan(math.radians(-5.0))
    sk = np.stack([z + 1.0, z + tx, -o[:, 1] * tx, z + ty, z + 1.0, -o[:, 0] * ty], axis=1)
    assert np.array_equal(s.skew(10.0, -5.0, origin).array.xy, _affine_rows_reference(a, sk, oracle))


def test_one_to_one_columns_skip_the_combine_pass_and_keep_nulls_and_empties(gpk, oracle):
    """single-ring polygon, linestring and multipoint columns finish in stage 1 (no per-geometry combine): null rows
    still give NaN, empty rows NaN bounds / zero measures, clockwise rings a negative signed area"""
    polys = synth.star_polygons(5000, 7)
    xy = polys.xy.copy()
    ro = polys.ring_offsets
    for r in range(0, 5000, 3):  # every third ring clockwise
        xy[ro[r] : ro[r + 1]] = polys.xy[ro[r] : ro[r + 1]][::-1]
    keep = np.ones(5000, np.uint8)
    keep[::11] = 0
    a = GeoArrowArray(polys.geom_type, xy, polys.geom_offsets, ring_offsets=polys.ring_offsets, validity=np.packbits(keep, bitorder="little"))
    s = GeoSeries(a)
    _close(s.area(), oracle.area(a))
    _close(s.signed_area(), oracle.area(a, signed=True))
    assert (s.signed_area()[keep == 1] < 0).any()
    _close(s.euclidean_length(), oracle.euclidean_length(a))
    assert np.array_equal(s.bounds(), oracle.bounds(a), equal_nan=True)
    assert np.isnan(s.area()[0]) and np.isnan(s.bounds()[0]).all()
    lines = GeoArrowArray.from_linestrings([[(0, 0), (3, 4)], [], [(1, 1)], [(0, 0), (1, 0), (1, 1), (0, 1), (0, 0)]] * 300)
    sl = GeoSeries(lines)
    _close(sl.euclidean_length(), oracle.euclidean_length(lines))
    assert np.array_equal(sl.bounds(), oracle.bounds(lines), equal_nan=True)
    mp = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.arange(40.0).reshape(20, 2), geom_offsets=np.array([0, 3, 3, 10, 20], np.int32))
    assert np.array_equal(GeoSeries(mp).bounds(), oracle.bounds(mp), equal_nan=True)
    # centroids of the same one-to-one columns, degenerate single-ring polygons included
    for arr in (a, lines, mp):
        exp_c, exp_v = oracle.centroid(arr)
        got = GeoSeries(arr).centroid()
        _close(got.array.xy, exp_c)
    degenerate = GeoArrowArray.from_polygons(
        [
            [[(0, 0), (4, 0), (4, 3), (0, 3)]],
            [[(1, 1), (2, 2), (3, 3), (1, 1)]],  # zero-area ring -> linestring centroid
            [[(4, 4), (4, 4), (4, 4), (4, 4)]],  # all-identical ring -> point centroid
            [[(0, 0), (0, 5), (5, 5), (5, 0)]],  # clockwise
        ]
        * 500
        + [[_ring(20000, 3.0, 7.0, 7.0)], [[(float(i), 2.0 * i) for i in range(9000)] + [(float(i), 2.0 * i) for i in range(8998, -1, -1)]]]
    )
    exp_c, exp_v = oracle.centroid(degenerate)
    got = GeoSeries(degenerate).centroid()
    _close(got.array.xy, exp_c)


def test_convex_hull_adversarial_point_sets(gpk, oracle):
    """the cooperative hull (sort + parallel left-turn filtering, 64- and 128-point instantiations, the chain beyond): point
    sets built to stress the filtering — everything collinear, everything identical, many duplicates, points ON the chord
    between the extremes, a convex polygon (nothing leaves), a spiral (the filter needs many rounds), lattice blocks full of
    collinear triples — at every size around the instantiation limits"""
    rng = np.random.default_rng(5)
    sets = []
    for n in (1, 2, 3, 4, 5, 16, 17, 63, 64, 65, 66, 127, 128, 129, 130, 300):
        t = np.linspace(0.0, 1.0, n)
        sets.append(np.stack([t * 7.0 - 3.0, t * 14.0 + 1.0], axis=1))  # collinear
        sets.append(np.tile([[2.5, -1.25]], (n, 1)))  # one point, n times
        k = np.arange(n)
        ang = 2 * np.pi * k / max(n, 1)
        sets.append(np.stack([np.cos(ang), np.sin(ang)], axis=1) * 10.0)  # convex position
        sets.append(np.stack([np.cos(6 * ang), np.sin(6 * ang)], axis=1) * (1.0 + k[:, None] / max(n, 1)))  # spiral
        sets.append(rng.integers(0, 4, (n, 2)).astype(np.float64))  # tiny lattice: duplicates and collinear triples everywhere
        half = rng.integers(-5, 6, (n, 2)).astype(np.float64)
        half[::2, 1] = half[::2, 0] * 2.0  # every other point on the line y = 2x (through the likely extremes)
        sets.append(half)
        sets.append(rng.permutation(np.concatenate([rng.normal(size=(n, 2)), rng.normal(size=(n, 2))])[: max(n, 1)]))
    a = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.concatenate(sets), geom_offsets=np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int32))
    exp_xy, exp_off = oracle.convex_hull(a)
    h = GeoSeries(a).convex_hull().array
    assert np.array_equal(h.ring_offsets, exp_off)
    for g in range(len(a)):
        got = _canon(h.xy[h.ring_offsets[g] : h.ring_offsets[g + 1]])
        exp = _canon(exp_xy[exp_off[g] : exp_off[g + 1]])
        assert np.array_equal(got, exp), (g, len(sets[g]))
    # without canonicalisation: the ring starts at the lexicographically smallest vertex and is closed
    for g in range(len(a)):
        ring = h.xy[h.ring_offsets[g] : h.ring_offsets[g + 1]]
        if len(ring) >= 2:
            assert np.array_equal(ring[0], ring[-1])
            assert tuple(ring[0]) == min(map(tuple, ring))
