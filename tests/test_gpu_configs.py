"""GPU parity at configuration scale (BASELINE.json configs C2-C5, >= 1M rows on the sharded side): the HIP path through
the C ABI against the CPU oracle on a RANDOM sample of rows (not a prefix) — the same sampled check bench.py gates its
numbers with.  Sizes are the smallest that exercise what the full configurations exercise (grouped-distance schedule,
segmented candidate lists, multi-hit pools, the lean and the general tile kernel) while the oracle still finishes in
seconds on a sample."""
import numpy as np
import pytest

from geopolars_amd import synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs

pytestmark = pytest.mark.gpu


def _sample(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(n, k), replace=False)).astype(np.int64)


def _pairs_of_rows(pairs, rows):
    lo = np.searchsorted(pairs[:, 0], rows, side="left")
    hi = np.searchsorted(pairs[:, 0], rows, side="right")
    cnt = hi - lo
    idx = np.repeat(lo - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
    out = pairs[idx].copy()
    out[:, 0] = np.repeat(np.arange(len(rows), dtype=pairs.dtype), cnt)
    return out


def _check_join_sample(oracle, left, right, predicate, pairs, counts, k, seed):
    idx = _sample(len(left), k, seed)
    ep, ec, _ = oracle.spatial_join(left.take(idx), right, predicate, mode=1)
    assert np.array_equal(counts[idx], ec)
    assert np.array_equal(_pairs_of_rows(pairs, idx.astype(np.uint32)), ep)
    assert int(counts.sum()) == len(pairs)
    assert np.all(np.diff(pairs[:, 0].astype(np.int64)) >= 0)  # sorted by l ...
    same = pairs[1:, 0] == pairs[:-1, 0]
    assert np.all(pairs[1:, 1][same] > pairs[:-1, 1][same])  # ... then strictly by r


def test_c2_scale_sampled(gpk, oracle):
    """C2 at 4M points x 1000 x 64 (lean tile kernel: disjoint polygons), 200k random rows against the oracle."""
    polys, pts = synth.star_polygons(1000, 64), synth.uniform_points(4_000_000, seed=101)
    ps, qs = GeoSeries(pts), GeoSeries(polys)
    pairs, counts = join_pairs(ps, qs, "intersects", r_index=SpatialIndex(qs))
    _check_join_sample(oracle, pts, polys, "intersects", pairs, counts, 200_000, 7)


def test_c2_scale_general_kernel_equals_lean(gpk, oracle):
    """the same join through the general tile kernel (GPK_NO_LEAN is read once per process: use a right side the lean
    kernel does not take — two overlapping copies of the polygons) equals the union of the single-copy answers"""
    polys = synth.star_polygons(400, 32)
    both = GeoArrowArray.concat([polys, polys])
    pts = synth.uniform_points(1_000_000, seed=102)
    p1, c1 = join_pairs(GeoSeries(pts), GeoSeries(polys), "contains")
    p2, c2 = join_pairs(GeoSeries(pts), GeoSeries(both), "contains")
    assert np.array_equal(c2, 2 * c1)
    exp = np.concatenate([p1, p1 + np.array([0, 400], dtype=np.uint32)])
    exp = exp[np.lexsort((exp[:, 1], exp[:, 0]))]
    assert np.array_equal(p2, exp)


@pytest.mark.parametrize("shuffled", [False, True])
def test_c3_scale_sampled(gpk, oracle, shuffled):
    """C3 at 2M points x 20k linestrings (100 rows per target: the grouped schedule; shuffled map: the radix grouping)."""
    n, L = 2_000_000, 20_000
    ls, pts = synth.random_linestrings(L), synth.uniform_points(n, seed=103)
    rows = (np.arange(n, dtype=np.uint32) % L).astype(np.uint32)
    if shuffled:
        rows = np.random.default_rng(3).permutation(rows)
    got = GeoSeries(pts).distance(GeoSeries(ls), other_rows=rows)
    idx = _sample(n, 200_000, 8)
    exp = oracle.distance_rowwise(pts.take(idx), ls, rows[idx])
    g = got[idx]
    rel = np.abs(g - exp) / np.maximum(np.abs(exp), 1e-300)
    assert np.all((rel <= 1e-9) | (g == exp))  # tolerance of the north star: 1e-9 relative
    assert np.array_equal(g == 0.0, exp == 0.0)


def test_c3_prepared_row_map_equals_one_shot(gpk, oracle):
    """gpk_rowmap_build + gpk_distance_rowmap (the map ordered once) answers exactly what gpk_distance_rowwise answers, for
    skewed maps too: many distinct targets per 64-row chunk, targets without rows, long linestrings (several LDS windows)."""
    from geopolars_amd.geoseries import RowMap

    rng = np.random.default_rng(5)
    ls = synth.random_linestrings(3000, min_log2=0.0, max_log2=8.0)
    pts = synth.uniform_points(400_000, seed=106)
    gs, ps = GeoSeries(ls), GeoSeries(pts)
    for rows in (
        (np.arange(400_000) % 3000).astype(np.uint32),
        rng.integers(0, 3000, 400_000).astype(np.uint32),
        np.minimum(rng.zipf(1.3, 400_000) - 1, 2999).astype(np.uint32),  # a few targets own most rows, most targets own none
        np.sort(rng.integers(0, 3000, 400_000)).astype(np.uint32),
    ):
        one = ps.distance(gs, other_rows=rows)
        rm = RowMap(gs, rows)
        two = ps.distance(gs, row_map=rm)
        assert np.array_equal(one, two)
        idx = _sample(400_000, 60_000, 11)
        exp = oracle.distance_rowwise(pts.take(idx), ls, rows[idx])
        g = two[idx]
        rel = np.abs(g - exp) / np.maximum(np.abs(exp), 1e-300)
        assert np.all((rel <= 1e-9) | (g == exp))
        assert np.array_equal(g == 0.0, exp == 0.0)


def test_c3_points_on_the_linestring_are_at_zero(gpk, oracle):
    """vertices, points on axis-aligned segments and (as representable) segment midpoints: the zero / non-zero outcome of
    upstream's line_string_contains_point short-circuit is replayed exactly by the grouped kernel"""
    ls = synth.random_linestrings(500, min_log2=1.0, max_log2=6.0)
    xy, off = ls.xy, ls.geom_offsets
    rng = np.random.default_rng(6)
    rows, pts = [], []
    for t in range(500):
        v = xy[off[t] : off[t + 1]]
        for _ in range(20):  # >= 8 rows per target: the grouped schedule
            k = rng.integers(0, len(v) - 1)
            kind = rng.integers(0, 4)
            a, b = v[k], v[k + 1]
            pts.append(a if kind == 0 else ((a + b) / 2 if kind == 1 else (a + (b - a) * 0.25 if kind == 2 else a + np.array([1e-3, -1e-3]))))  # (1e-3: a point 1e-9 off a segment of a 1e3-sized coordinate frame has no 1e-9-relative distance in f64, upstream's formula included)
            rows.append(t)
    pts = GeoArrowArray.from_points(np.array(pts))
    rows = np.array(rows, dtype=np.uint32)
    got = GeoSeries(pts).distance(GeoSeries(ls), other_rows=rows)
    exp = oracle.distance_rowwise(pts, ls, rows)
    assert np.array_equal(got == 0.0, exp == 0.0)
    assert (exp == 0.0).sum() > 2000
    rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)
    # "as representable" mid points that upstream does not call on the segment sit ~1e-13 off it: the value of such a
    # distance is rounding noise of the 1e3-sized coordinates in ANY evaluation order (upstream's included): both tiny
    tiny = (exp < 1e-9) & (got < 1e-9)
    assert np.all((rel <= 1e-9) | (got == exp) | tiny)
    assert np.all(rel[exp > 1e-6] <= 1e-9)


def test_c3_out_of_range_row_map_entries_are_null_rows(gpk):
    """b_rows entries >= n_geoms(b) behave like null rows (NaN), on the grouped and on the row-major schedule"""
    ls, pts = synth.random_linestrings(50), synth.uniform_points(5000, seed=104)
    for rows in ((np.arange(5000) % 50).astype(np.uint32), np.arange(5000, dtype=np.uint32) % 50):
        rows = rows.copy()
        bad = np.array([0, 17, 4999, 2500])
        rows[bad] = [50, 1 << 31, 0xFFFFFFFF, 51]
        got = GeoSeries(pts).distance(GeoSeries(ls), other_rows=rows)
        assert np.all(np.isnan(got[bad]))
        ok = np.ones(5000, dtype=bool)
        ok[bad] = False
        assert not np.any(np.isnan(got[ok]))
    few = GeoSeries(synth.uniform_points(100, seed=105))  # fewer than 8 rows per target: the row-major kernel
    rows = np.arange(100, dtype=np.uint32) % 50
    rows[3] = 50
    got = few.distance(GeoSeries(ls), other_rows=rows)
    assert np.isnan(got[3]) and not np.any(np.isnan(np.delete(got, 3)))


def test_c4_scale_sampled(gpk, oracle):
    """C4 at full size: 1M x 1M polygons, index without the point tables, 100k random left rows against the oracle."""
    left = synth.clustered_polygons(1_000_000, seed=41, mean_neighbours=4.0)
    right = synth.clustered_polygons(1_000_000, seed=42, mean_neighbours=4.0)
    ls, rs = GeoSeries(left), GeoSeries(right)
    pairs, counts = join_pairs(ls, rs, "intersects", r_index=SpatialIndex(rs, for_points=False))
    assert len(pairs) > 2_000_000
    _check_join_sample(oracle, left, right, "intersects", pairs, counts, 100_000, 9)


def test_c5_scale_sampled(gpk, oracle):
    """C5 shape: 2M points within 600k power-law multipolygons (overlaps, holes, multi-hit rows) + area of all of them."""
    mp = synth.powerlaw_multipolygons(600_000, seed=51)
    pts = synth.uniform_points(2_000_000, seed=52)
    ms, ps = GeoSeries(mp), GeoSeries(pts)
    pairs, counts = join_pairs(ps, ms, "within", r_index=SpatialIndex(ms))
    assert counts.max() >= 2
    _check_join_sample(oracle, pts, mp, "within", pairs, counts, 200_000, 10)
    area, exp = ms.area(), oracle.area(mp)
    assert np.all(np.abs(area - exp) <= 1e-9 * np.maximum(np.abs(exp), 1e-300))


# ---- BASELINE.json's FULL sizes, every row against the oracle (OpenMP over rows on the box's host cores) ------------
def test_c2_full_size_every_row(gpk, oracle):
    """C2 as quoted: 10M points x 1000 x 64-vertex polygons — counts and pairs of ALL rows equal the oracle's."""
    polys, pts = synth.star_polygons(1000, 64), synth.uniform_points(10_000_000, seed=12345)
    ps, qs = GeoSeries(pts), GeoSeries(polys)
    ix = SpatialIndex(qs)
    assert ix.describe()["route"] == 1  # the headline path: the routed tile kernel
    pairs, counts = join_pairs(ps, qs, "intersects", r_index=ix)
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    assert np.array_equal(counts, ec)
    assert np.array_equal(pairs, ep)
    # size-independent properties: counts sum to the pairs, rows sorted, the answer does not depend on the point order
    assert int(counts.sum()) == len(pairs) and np.all(np.diff(pairs[:, 0].astype(np.int64)) >= 0)
    perm = np.random.default_rng(4).permutation(len(pts))
    p2, c2 = join_pairs(GeoSeries(pts.take(perm)), qs, "intersects", r_index=ix)
    assert np.array_equal(c2, counts[perm])


def test_c3_full_size_every_row(gpk, oracle):
    """C3 as quoted: 10M points x 100k linestrings (row map i % 100k), all 10M distances within 1e-9 relative."""
    n, L = 10_000_000, 100_000
    ls, pts = synth.random_linestrings(L), synth.uniform_points(n, seed=777)
    rows = (np.arange(n, dtype=np.uint32) % L).astype(np.uint32)
    got = GeoSeries(pts).distance(GeoSeries(ls), other_rows=rows)
    exp = oracle.distance_rowwise(pts, ls, rows)
    rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)
    assert np.all((rel <= 1e-9) | (got == exp))  # tolerance of the north star: 1e-9 relative
    assert np.array_equal(got == 0.0, exp == 0.0)


def test_many_small_parts_index_variants(gpk, oracle):
    """more than two parts per cell of the 2048 raster (2.4M parts): the index takes a 4096 raster, plain entry lists and part
    boxes; GPK_INDEX_PIP_FULL adds the per-entry records — both answer like the oracle"""
    mp = synth.powerlaw_multipolygons(1_200_000, seed=61)
    pts = synth.uniform_points(1_000_000, seed=62)
    ms, ps = GeoSeries(mp), GeoSeries(pts)
    ix = SpatialIndex(ms)
    assert ix.describe()["R"] == 4096
    pairs, counts = join_pairs(ps, ms, "within", r_index=ix)
    assert counts.max() >= 2
    _check_join_sample(oracle, pts, mp, "within", pairs, counts, 100_000, 12)
    full = SpatialIndex(ms, full=True)
    p2, c2 = join_pairs(ps, ms, "within", r_index=full)
    assert np.array_equal(p2, pairs) and np.array_equal(c2, counts)
    assert full.nbytes() > ix.nbytes()


def test_c5_full_size_sampled(gpk, oracle):
    """C5 as quoted per GPU: 6.25M points within ALL 5M power-law multipolygons (8 chunks of 625k, as bench.py builds them) — 200k
    random rows against the oracle, the size-independent properties of the pair list on all of it, area() of all 5M rows."""
    mp = GeoArrowArray.concat([synth.powerlaw_multipolygons(625_000, seed=51 + k, size_n=5_000_000) for k in range(8)])
    pts = synth.uniform_points(6_250_000, seed=52)
    ms, ps = GeoSeries(mp), GeoSeries(pts)
    ix = SpatialIndex(ms)
    d = ix.describe()
    assert d["R"] == 4096 and d["list_heavy"] and not d["lean"]
    pairs, counts = join_pairs(ps, ms, "within", r_index=ix)
    assert counts.max() >= 2 and len(pairs) > 1_500_000
    _check_join_sample(oracle, pts, mp, "within", pairs, counts, 200_000, 13)
    area, exp = ms.area(), oracle.area(mp)
    assert np.all(np.abs(area - exp) <= 1e-9 * np.maximum(np.abs(exp), 1e-300))
