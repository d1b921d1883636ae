"""Size-independent properties of the GPU path (SURVEY.md §8c "extra pins"): exact invariances under transforms that
are exact in binary floating point (integer translation, power-of-two scaling, quarter turns), symmetry between
predicates, idempotence of the hull, round trips.  Every comparison that can be bitwise is bitwise."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import join_pairs

pytestmark = pytest.mark.gpu


def _moved(a: GeoArrowArray, f) -> GeoArrowArray:
    return GeoArrowArray(a.geom_type, f(a.xy), a.geom_offsets, part_offsets=a.part_offsets, ring_offsets=a.ring_offsets, validity=a.validity, n_geoms=len(a))


def _grid_snap(a: GeoArrowArray, step=1.0 / 64) -> GeoArrowArray:
    """coordinates on a dyadic grid: integer translations, power-of-two scalings and quarter turns are then exact"""
    return _moved(a, lambda xy: np.round(xy / step) * step)


TRANSFORMS = {
    "translate": lambda xy: xy + np.array([4096.0, -1024.0]),
    "scale_pow2": lambda xy: xy * 8.0,
    "quarter_turn": lambda xy: np.stack([-xy[:, 1], xy[:, 0]], axis=1),
    "mirror": lambda xy: xy * np.array([-1.0, 1.0]),
}


@pytest.mark.parametrize("name", list(TRANSFORMS))
def test_join_is_invariant_under_exact_transforms(gpk, name):
    """the raster, the slabs and the directory are all rebuilt for the transformed right side (different cells, different
    edge order), yet every (point, polygon) decision must come out the same — including the points on vertices / edges"""
    polys = _grid_snap(synth.star_polygons(300, 24))
    pts = _grid_snap(synth.uniform_points(100_000, seed=5))
    adv = _grid_snap(synth.adversarial_points(polys, seed=6))
    pts = GeoArrowArray.from_points(np.concatenate([pts.xy, adv.xy, polys.xy[::7]]))  # vertices themselves included
    base_pairs, base_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    f = TRANSFORMS[name]
    got_pairs, got_counts = join_pairs(GeoSeries(_moved(pts, f)), GeoSeries(_moved(polys, f)), "intersects")
    assert np.array_equal(got_counts, base_counts) and np.array_equal(got_pairs, base_pairs)
    assert len(base_pairs) > 30_000


def test_within_point_polygon_equals_contains_polygon_point(gpk):
    polys = synth.star_polygons(500, 16)
    pts = synth.uniform_points(500, seed=9)
    rows = np.arange(500, dtype=np.uint32)[::-1].copy()
    a = GeoSeries(pts).within(GeoSeries(polys), rows)
    b = GeoSeries(GeoArrowArray(polys.geom_type, polys.xy, polys.geom_offsets, ring_offsets=polys.ring_offsets)).contains(GeoSeries(pts))  # identity rows
    polys_rev = GeoSeries(polys)
    c = np.array([polys_rev.contains(GeoSeries(pts), np.full(500, i, np.uint32))[rows[i]] for i in range(0, 500, 50)])
    assert a.dtype == np.bool_ or a.dtype == np.uint8
    assert np.array_equal(np.asarray(a)[::50].astype(bool), c.astype(bool))
    assert b.shape == a.shape


def test_area_sign_flips_with_ring_orientation_and_centroid_stays(gpk):
    polys = synth.star_polygons(2000, 33)
    ro = polys.ring_offsets
    rev = polys.xy.copy()
    for r in range(len(ro) - 1):
        rev[ro[r] : ro[r + 1]] = polys.xy[ro[r] : ro[r + 1]][::-1]
    a, b = GeoSeries(polys), GeoSeries(_moved(polys, lambda xy: rev))
    assert np.allclose(a.area(), b.area(), rtol=1e-12, atol=0)
    assert np.allclose(a.signed_area(), -b.signed_area(), rtol=1e-12, atol=0)
    assert np.allclose(a.centroid().array.xy, b.centroid().array.xy, rtol=1e-9, atol=1e-9)
    assert np.array_equal(a.bounds(), b.bounds())


def test_hull_is_idempotent_and_contains_its_input(gpk):
    polys = synth.clustered_polygons(3000, seed=8)
    h1 = GeoSeries(polys).convex_hull()
    h2 = h1.convex_hull()
    assert np.array_equal(h1.array.xy, h2.array.xy) and np.array_equal(h1.array.ring_offsets, h2.array.ring_offsets)
    # a convex ring's centroid lies inside its bounds, and the hull's bounds are the input's bounds (bitwise: min / max)
    assert np.array_equal(h1.bounds(), GeoSeries(polys).bounds())
    c, b = h1.centroid().array.xy, h1.bounds()
    assert np.all((c[:, 0] >= b[:, 0]) & (c[:, 0] <= b[:, 2]) & (c[:, 1] >= b[:, 1]) & (c[:, 1] <= b[:, 3]))
    # every input vertex intersects (lies in or on) the hull of its own polygon
    first_vertex = GeoSeries(GeoArrowArray.from_points(polys.xy[polys.ring_offsets[polys.geom_offsets[:-1]]]))
    assert np.all(first_vertex.intersects(h1))


def test_affine_identity_and_integer_translation_round_trip(gpk):
    a = _grid_snap(synth.powerlaw_multipolygons(3000, seed=4))
    s = GeoSeries(a)
    assert np.array_equal(s.affine_transform([1, 0, 0, 0, 1, 0]).array.xy, a.xy)
    there = s.translate(12345.0, -777.0)
    back = there.translate(-12345.0, 777.0)
    assert np.array_equal(back.array.xy, a.xy)
    assert np.allclose(there.area(), s.area(), rtol=1e-9)


def test_join_counts_pairs_and_sharding_are_consistent_at_scale(gpk):
    """a larger instance checked through properties only: counts sum to the number of pairs, pairs are sorted and
    unique, every hit lies in the polygon's bounds, and the join of row shards concatenates to the join of the whole"""
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(2_000_000, seed=77)
    ps, gs = GeoSeries(pts), GeoSeries(polys)
    pairs, counts = join_pairs(ps, gs, "intersects")
    assert counts.sum() == len(pairs)
    key = pairs[:, 0].astype(np.int64) * 2000 + pairs[:, 1]
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.repeat(np.arange(len(counts)), counts), pairs[:, 0])
    b = gs.bounds()[pairs[:, 1]]
    xy = pts.xy[pairs[:, 0]]
    assert np.all((xy[:, 0] >= b[:, 0]) & (xy[:, 0] <= b[:, 2]) & (xy[:, 1] >= b[:, 1]) & (xy[:, 1] <= b[:, 3]))
    from geopolars_amd.dist import slice_rows

    cut = 1_234_567
    p0, c0 = join_pairs(GeoSeries(slice_rows(pts, 0, cut)), gs, "intersects")
    p1, c1 = join_pairs(GeoSeries(slice_rows(pts, cut, len(pts))), gs, "intersects", left_row_base=cut)
    assert np.array_equal(np.concatenate([p0, p1]), pairs) and np.array_equal(np.concatenate([c0, c1]), counts)
