"""GPU join / predicate parity on inputs built to stress the accelerated routes of gpk_pip_tile: raster and
sub-cell borders, polygons smaller than a sub-cell, holes, overlapping and nested polygons (multi-hit rows),
degenerate extents (raster off), large coordinate offsets (raster guard), degenerate rings, null rows."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import join_pairs

pytestmark = pytest.mark.gpu


def check_join(oracle, pts: GeoArrowArray, polys: GeoArrowArray, pred="intersects"):
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, pred, mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), pred)
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)
    return exp_pairs, exp_counts


def lattice(x0, x1, y0, y1, step):
    gx, gy = np.meshgrid(np.arange(x0, x1 + step / 2, step), np.arange(y0, y1 + step / 2, step))
    return GeoArrowArray.from_points(np.stack([gx.ravel(), gy.ravel()], axis=1))


def test_points_on_raster_lines_and_vertices(gpk, oracle):
    """extent 0..1021 with R = 64..512 puts raster lines on integers / simple fractions; lattice points land
    exactly on cell and sub-cell borders, on polygon vertices and on axis-aligned edges."""
    polys = GeoArrowArray.from_polygons(
        [[[(x, y), (x + 96, y), (x + 96, y + 64), (x + 48, y + 96), (x, y + 64)]] for x in range(0, 1000, 128) for y in range(0, 1000, 128)]
        + [[[(0, 0), (1021, 0), (1021, 1021), (0, 1021)], [(10, 10), (10, 1011), (1011, 1011), (1011, 10)]]]  # frame with a big hole
    )
    exp_pairs, exp_counts = check_join(oracle, lattice(-8, 1032, -8, 1032, 8.0), polys)
    assert exp_counts.max() >= 1 and len(exp_pairs) > 1000
    check_join(oracle, lattice(0, 1021, 0, 1021, 1021 / 509.0), polys)  # exactly the raster pitch


def test_polygons_smaller_than_a_subcell(gpk, oracle):
    rng = np.random.default_rng(3)
    big = [[[(0, 0), (1000, 0), (1000, 1000), (0, 1000)]]]
    c = rng.uniform(10, 990, (400, 2))
    tiny = [[[(x, y), (x + 0.01, y), (x + 0.01, y + 0.01), (x, y + 0.01)]] for x, y in c]
    polys = GeoArrowArray.from_polygons(big + tiny)
    pts = np.concatenate([c + 0.005, c, c + [0.01, 0.0], rng.uniform(0, 1000, (5000, 2))])
    exp_pairs, exp_counts = check_join(oracle, GeoArrowArray.from_points(pts), polys)
    assert exp_counts.max() == 2  # inside the big square and inside a tiny one


def test_overlapping_and_nested_polygons_multi_hit(gpk, oracle):
    rings = []
    for k in range(12):  # 12 concentric squares: a central point is in all of them
        r = 40 + 30 * k
        rings.append([[(500 - r, 500 - r), (500 + r, 500 - r), (500 + r, 500 + r), (500 - r, 500 + r)]])
    polys = GeoArrowArray.from_polygons(rings + [[[(0, 0), (300, 0), (0, 300)]], [[(100, 100), (400, 100), (100, 400)]]])
    exp_pairs, exp_counts = check_join(oracle, lattice(0, 1000, 0, 1000, 10.0), polys)
    assert exp_counts.max() == 12
    # sorted by (l, r)
    assert np.all(np.diff(exp_pairs[:, 0].astype(np.int64)) >= 0)


def test_sparse_three_to_six_hit_rows_use_the_overflow_list(gpk, oracle):
    """power-law multipolygons overlap irregularly: most rows have 0-2 hits, a few have 3-6 (the rows that go through
    the tile's overflow list and the multi-hit pool), with multipolygon parts mapped back to geometry ids"""
    mp = synth.powerlaw_multipolygons(4000, seed=77, domain=300.0)
    pts = synth.uniform_points(60_000, seed=78, domain=300.0)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, mp, "within", mode=1)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(mp), "within")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
    hist = np.bincount(exp_counts)
    assert len(hist) > 3 and hist[2] > 0 and hist[3:].sum() > 0, hist


def test_multipolygon_with_overlapping_parts_counts_geometry_once(gpk, oracle):
    mp = GeoArrowArray.from_multipolygons(
        [
            [[[(0, 0), (10, 0), (10, 10), (0, 10)]], [[(5, 5), (15, 5), (15, 15), (5, 15)]]],  # overlapping parts (invalid, still defined)
            [[[(20, 0), (30, 0), (30, 10), (20, 10)], [(22, 2), (22, 8), (28, 8), (28, 2)]], [[(24, 4), (26, 4), (26, 6), (24, 6)]]],  # island in a hole
        ]
    )
    exp_pairs, exp_counts = check_join(oracle, lattice(-1, 31, -1, 16, 0.5), mp, "within")
    assert exp_counts.max() == 1


def test_degenerate_extent_disables_raster(gpk, oracle):
    """all polygon vertices on one horizontal line: zero-height extent -> generic exact path"""
    flat = GeoArrowArray.from_polygons([[[(0, 5), (10, 5), (20, 5), (0, 5)]], [[(30, 5), (40, 5), (30, 5)]]])
    check_join(oracle, lattice(-5, 45, 0, 10, 2.5), flat)
    one = GeoArrowArray.from_polygons([[[(3, 3), (3, 3), (3, 3)]]])  # a single repeated coordinate
    check_join(oracle, GeoArrowArray.from_points([[3, 3], [3, 4], [2, 3]]), one)


def test_large_coordinate_offset_raster_guard(gpk, oracle):
    """UTM-like magnitudes with a metre-scale extent, then an offset so large that a raster cell is only a few
    ulps wide: the accelerator must switch itself off rather than mislabel cells."""
    base = synth.star_polygons(50, 16)
    pts = synth.uniform_points(20_000)
    for off, scale in ((5.0e6, 1.0), (2.0 ** 40, 2.0 ** -10), (2.0 ** 50, 2.0 ** -20)):
        polys = GeoArrowArray(base.geom_type, base.xy * scale + off, base.geom_offsets, ring_offsets=base.ring_offsets)
        p = GeoArrowArray.from_points(pts.xy * scale + off)
        exp_pairs, _ = check_join(oracle, p, polys)  # parity is the point; the last case collapses most coordinates
        assert len(exp_pairs) > 100 or off >= 2.0 ** 50


def test_degenerate_rings_and_empty_polygons(gpk, oracle):
    polys = GeoArrowArray.from_polygons(
        [
            [],  # empty polygon
            [[(0, 0), (4, 0), (4, 4), (0, 4)]],
            [[(10, 10), (12, 10)]],  # two-coordinate "ring" (closed on the way in -> 3 coords, zero area)
            [[(6, 0), (8, 0), (8, 2), (6, 2), (6, 0), (8, 0), (8, 2), (6, 2)]],  # ring walked twice: winding number 2
            [[(0, 6), (4, 6), (0, 10), (4, 10)]],  # bow-tie (self-intersecting)
        ],
    )
    check_join(oracle, lattice(-1, 13, -1, 11, 0.5), polys)


def test_null_rows_never_match(gpk, oracle):
    polys = synth.star_polygons(40, 12)
    keep = np.ones(len(polys), np.uint8)
    keep[::3] = 0
    polys_n = GeoArrowArray(polys.geom_type, polys.xy, polys.geom_offsets, ring_offsets=polys.ring_offsets, validity=np.packbits(keep, bitorder="little"))
    pts = synth.uniform_points(30_000)
    pk = np.ones(len(pts), np.uint8)
    pk[::5] = 0
    pts_n = GeoArrowArray.from_points(pts.xy, validity=np.packbits(pk, bitorder="little"))
    exp_pairs, exp_counts = check_join(oracle, pts_n, polys_n)
    assert not np.isin(exp_pairs[:, 1], np.arange(0, len(polys), 3)).any()
    assert exp_counts[::5].sum() == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_small_worlds(gpk, oracle, seed):
    """random mixtures: a few stars, boxes sharing edges (tessellation: shared-border cells), lattice + random
    points; extents and counts vary so that different raster sizes (R = 64..512) are exercised."""
    rng = np.random.default_rng(100 + seed)
    n_star = int(rng.integers(1, 200))
    stars = synth.star_polygons(n_star, int(rng.integers(3, 40)), seed=seed)
    k = int(rng.integers(2, 9))
    w = 1000.0 / k
    boxes = GeoArrowArray.from_polygons([[[(i * w, j * w), ((i + 1) * w, j * w), ((i + 1) * w, (j + 1) * w), (i * w, (j + 1) * w)]] for i in range(k) for j in range(k)])
    polys = GeoArrowArray(
        stars.geom_type,
        np.concatenate([stars.xy, boxes.xy]),
        geom_offsets=np.concatenate([stars.geom_offsets, boxes.geom_offsets[1:] + stars.geom_offsets[-1]]),
        ring_offsets=np.concatenate([stars.ring_offsets, boxes.ring_offsets[1:] + stars.ring_offsets[-1]]),
    )
    pts = np.concatenate([rng.uniform(-20, 1020, (20_000, 2)), lattice(0, 1000, 0, 1000, w / 4).xy, polys.xy[:: max(1, len(polys.xy) // 500)]])
    exp_pairs, exp_counts = check_join(oracle, GeoArrowArray.from_points(pts), polys)
    assert exp_counts.max() >= 1


def test_rowwise_predicates_on_degenerate_inputs(gpk, oracle):
    polys = GeoArrowArray.from_polygons(
        [[], [[(0, 0), (4, 0), (4, 4), (0, 4)], [(1, 1), (1, 3), (3, 3), (3, 1)]], [[(6, 0), (8, 0), (8, 2), (6, 2), (6, 0), (8, 0), (8, 2), (6, 2)]]]
    )
    pts = lattice(-1, 9, -1, 5, 0.5)
    rows = (np.arange(len(pts)) % len(polys)).astype(np.uint32)
    for pred in ("within", "intersects"):
        exp = oracle.predicate_rowwise(pts, polys, pred, rows)
        got = getattr(GeoSeries(pts), pred)(GeoSeries(polys), rows)
        assert np.array_equal(got, exp)
    d_exp = oracle.distance_rowwise(pts, polys, rows)
    d_got = GeoSeries(pts).distance(GeoSeries(polys), rows)
    assert np.array_equal(d_got == 0, d_exp == 0)
    m = np.isfinite(d_exp) & (d_exp < 1e300)
    assert np.allclose(d_got[m], d_exp[m], rtol=1e-9, atol=0)
    assert np.array_equal(d_got[~m], d_exp[~m], equal_nan=True)


def test_tessellation_shared_borders_two_part_cells(gpk, oracle):
    """polygons that share every border: all boundary cells hold two crossing parts (two-part level-2 records).  Points on
    shared edges / vertices intersect both neighbours (or four at a corner) and are contained in none."""
    tess = synth.tessellation(12, 8, seed=5)
    rng = np.random.default_rng(6)
    ro = tess.ring_offsets
    verts = tess.xy[np.unique(rng.integers(0, tess.n_coords, 3000))]
    mids = (tess.xy[:-1] + tess.xy[1:])[np.setdiff1d(np.arange(tess.n_coords - 1), ro[1:-1] - 1)][::5] / 2.0  # edge midpoints (exact: halves)
    pts = GeoArrowArray.from_points(np.concatenate([synth.uniform_points(150_000, seed=7).xy, verts, mids, verts + 1e-9]))
    for pred in ("intersects", "within"):
        exp_pairs, exp_counts, _ = oracle.spatial_join(pts, tess, pred, mode=1)
        got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(tess), pred)
        assert np.array_equal(got_counts, exp_counts), pred
        assert np.array_equal(got_pairs, exp_pairs), pred
    hist = np.bincount(exp_counts)
    assert hist[0] > 1000 and hist[1] > 100_000  # border points are contained in no polygon, interior points in exactly one


def test_small_polygons_inside_big_ones(gpk, oracle):
    """cells covered by one part and crossed by another (entry lists): rows with two hits"""
    big = [[[(x, y), (x + 250.0, y), (x + 250.0, y + 250.0), (x, y + 250.0)]] for x in (0.0, 250.0, 500.0, 750.0) for y in (0.0, 250.0, 500.0, 750.0)]
    stars = synth.star_polygons(900, 16, seed=11)
    small = [[stars.xy[stars.ring_offsets[r] : stars.ring_offsets[r + 1] - 1].tolist()] for r in range(len(stars))]
    polys = GeoArrowArray.from_polygons(big + small)
    pts = GeoArrowArray.from_points(np.concatenate([synth.uniform_points(200_000, seed=12).xy, synth.adversarial_points(stars, seed=13).xy]))
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
    assert np.bincount(exp_counts)[2] > 10_000


def test_few_vertex_rings_spanning_the_whole_raster(gpk, oracle):
    """Rings with 8 vertices that each cover most of the extent: far more slab rows than coordinates (the scan scratch
    of the index build is sized by the longest scanned array, which here is the slab table), hundreds of overlapping
    parts per cell, and points on the integer lattice (vertices, edges) and off it."""
    import random

    from .lattice import concentric_pair, random_pair

    rng = random.Random(77)
    polys = [p for pair in (concentric_pair(rng) for _ in range(100)) for p in pair] + [p for pair in (random_pair(rng) for _ in range(100)) for p in pair]
    right = GeoArrowArray.from_polygons(polys)
    g = np.random.default_rng(78)
    pts = np.concatenate([g.integers(-14, 15, (6000, 2)).astype(np.float64), g.integers(-28, 29, (6000, 2)) / 2.0, g.uniform(-14, 14, (6000, 2))])
    left = GeoArrowArray.from_points(pts)
    exp_pairs, exp_counts, _ = oracle.spatial_join(left, right, "intersects", mode=1)
    assert exp_counts.max() > 50
    got_pairs, got_counts = join_pairs(GeoSeries(left), GeoSeries(right), "intersects")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)


def _wiggly_ring(n, cx, cy, r_lo, r_hi, seed, reverse=False):
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * (np.arange(n) + rng.uniform(0, 0.9, n)) / n
    rad = rng.uniform(r_lo, r_hi, n)
    ring = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
    ring = np.concatenate([ring, ring[:1]])
    return ring[::-1] if reverse else ring


def test_refined_slab_rows_of_dense_rings(gpk, oracle):
    """A 30k-vertex exterior and a 6k-vertex hole put ~30 edges into a base slab row, so both rings get refined rows
    (shift 2): level-2 records of the part are SUB_INDIRECT (slab through PartInfo), the hole walk uses the hole's own
    row shift, and the small polygon on top turns some cells into entry lists."""
    ext = _wiggly_ring(30_000, 500, 500, 380, 450, 1)
    hole = _wiggly_ring(6_000, 520, 480, 90, 110, 2, reverse=True)
    small = _wiggly_ring(40, 150, 500, 20, 30, 3)
    xy = np.concatenate([ext, hole, small])
    polys = GeoArrowArray(
        _abi.GEOM_POLYGON, xy, geom_offsets=np.array([0, 2, 3], np.int32), ring_offsets=np.array([0, len(ext), len(ext) + len(hole), len(xy)], np.int32)
    )
    rng = np.random.default_rng(4)
    ang = rng.uniform(0, 2 * np.pi, 30_000)
    near_ext = np.stack([500 + rng.uniform(375, 455, 30_000) * np.cos(ang), 500 + rng.uniform(375, 455, 30_000) * np.sin(ang)], 1)
    near_hole = np.stack([520 + rng.uniform(85, 115, 30_000) * np.cos(ang), 480 + rng.uniform(85, 115, 30_000) * np.sin(ang)], 1)
    pts = np.concatenate([rng.uniform(0, 1000, (30_000, 2)), near_ext, near_hole, ext[::40], hole[::20], (ext[:-1:50] + ext[1::50]) / 2])
    exp_pairs, exp_counts = check_join(oracle, GeoArrowArray.from_points(pts), polys)
    assert 20_000 < len(exp_pairs) < len(pts) and exp_counts.max() == 2


def test_refined_slab_rows_in_two_part_cells(gpk, oracle):
    """a 12 x 12 tessellation whose shared borders have 3000 segments per side: two-part level-2 records (SubCell2) whose
    parts both have refined rows"""
    t = synth.tessellation(12, 3000)
    rng = np.random.default_rng(5)
    k = rng.integers(0, len(t.xy), 40_000)
    on_border = t.xy[k]
    near_border = on_border + rng.uniform(-0.02, 0.02, on_border.shape)
    pts = np.concatenate([rng.uniform(0, 1000, (60_000, 2)), on_border, near_border])
    exp_pairs, exp_counts = check_join(oracle, GeoArrowArray.from_points(pts), t)
    # a border vertex is on the boundary of both neighbours: contained in neither; everything else is in exactly one cell
    assert exp_counts.max() == 1 and 35_000 < (exp_counts == 0).sum() < 45_000
