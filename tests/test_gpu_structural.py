"""GPU: the structural operators of `trait GeoSeries` through the C ABI (gpk_structural.hip) — envelope, exterior, explode,
geom_type, is_empty, is_ring, x / y, rotate / scale / skew about a per-geometry origin — against plain numpy restatements of
what the reference's docs say (geoseries.rs:28-83,85-139,177-180) and against the two operators the reference benches
(benches/explode.rs:10-24, benches/affine.rs:23-26)."""
import math

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

pytestmark = pytest.mark.gpu


def _with_nulls(a: GeoArrowArray, every: int = 5) -> GeoArrowArray:
    keep = np.ones(len(a), dtype=bool)
    keep[::every] = False
    return GeoArrowArray(a.geom_type, a.xy, a.geom_offsets, a.part_offsets, a.ring_offsets, np.packbits(keep, bitorder="little"), n_geoms=len(a))


def test_geoseries_structural_accessors(gpk):
    polys = GeoArrowArray.from_polygons([[[(0, 0), (4, 0), (4, 4), (0, 4)], [(1, 1), (1, 2), (2, 2), (2, 1)]], [], [[(5, 5), (6, 5), (6, 6)]]])
    s = GeoSeries(polys)
    assert s.geom_type().tolist() == [3, 3, 3]
    assert s.is_empty().tolist() == [False, True, False]
    ext = s.exterior().array
    assert ext.geom_type == _abi.GEOM_LINESTRING and ext.geom_offsets.tolist() == [0, 5, 5, 9]
    assert np.array_equal(ext.xy[:5], polys.xy[:5]) and np.array_equal(ext.xy[5:], polys.xy[10:14])
    pts = GeoSeries(GeoArrowArray.from_points([[1.0, 2.0], [np.nan, np.nan]]))
    assert pts.x().tolist()[0] == 1.0 and pts.y().tolist()[0] == 2.0 and pts.is_empty().tolist() == [False, True]
    with pytest.raises(_abi.MismatchedGeometry):
        s.x()
    with pytest.raises(_abi.MismatchedGeometry):
        pts.exterior()
    with pytest.raises(_abi.MismatchedGeometry):
        s.is_ring()


def test_explode_and_is_ring(gpk):
    mp = GeoArrowArray.from_multipolygons([[[[(0, 0), (1, 0), (0, 1)]], [[(5, 5), (6, 5), (5, 6)]]], [[[(2, 2), (3, 2), (2, 3)]]]])
    ex, parents = GeoSeries(mp).explode(return_parents=True)
    exa = ex.array
    assert exa.geom_type == _abi.GEOM_POLYGON and len(exa) == 3 and exa.ring_offsets.tolist() == [0, 4, 8, 12] and exa.geom_offsets.tolist() == [0, 1, 2, 3]
    assert parents.tolist() == [0, 0, 1]
    assert np.array_equal(ex.area(), [0.5, 0.5, 0.5])  # the view is a working column: operators run on it
    pts = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.arange(8.0).reshape(4, 2), geom_offsets=np.array([0, 2, 4], np.int32))
    assert len(GeoSeries(pts).explode()) == 4  # benches/explode.rs: two-point MultiPoints -> points
    ls = GeoArrowArray.from_linestrings([[(0, 0), (1, 0), (1, 1), (0, 0)], [(0, 0), (1, 1)], [(2, 2)], []])
    # geo-types LineString::is_closed: first == last, and an EMPTY linestring counts as closed (JTS LinearRing rule)
    assert GeoSeries(ls).is_ring().tolist() == [True, False, True, True]


def test_explode_of_the_reference_bench_shape(gpk):
    """benches/explode.rs:10-24: 45,000 two-point MultiPoints -> 90,000 points sharing the coordinate buffer."""
    n = 45_000
    xy = np.random.default_rng(3).uniform(-180, 180, (2 * n, 2))
    mp = GeoArrowArray(_abi.GEOM_MULTIPOINT, xy, geom_offsets=np.arange(0, 2 * n + 1, 2, dtype=np.int32))
    ex, parents = GeoSeries(mp).explode(return_parents=True)
    assert len(ex) == 2 * n and ex.array.geom_type == _abi.GEOM_POINT and np.array_equal(ex.array.xy, xy)
    assert np.array_equal(parents, np.repeat(np.arange(n, dtype=np.int32), 2))


@pytest.mark.parametrize("kind", ["multipoly", "multiline", "multipoint"])
def test_explode_members_of_null_rows_are_null(gpk, kind):
    if kind == "multipoly":
        a = synth.powerlaw_multipolygons(400)
    elif kind == "multiline":
        ls = synth.random_linestrings(300)
        a = GeoArrowArray(_abi.GEOM_MULTILINESTRING, ls.xy, geom_offsets=np.arange(0, 301, 3, dtype=np.int32), ring_offsets=ls.geom_offsets)
    else:
        a = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.random.default_rng(1).uniform(0, 9, (500, 2)), geom_offsets=np.arange(0, 501, 5, dtype=np.int32))
    an = _with_nulls(a, 3)
    ex, parents = GeoSeries(an).explode(return_parents=True)
    exa = ex.array
    members = np.diff(a.geom_offsets)
    assert np.array_equal(parents, np.repeat(np.arange(len(a), dtype=np.int32), members))
    assert np.array_equal(exa.is_valid(), an.is_valid()[parents])
    assert np.array_equal(exa.xy, a.xy)
    assert np.array_equal(ex.geom_type(), np.where(exa.is_valid(), exa.geom_type, -1))


def test_envelope_rectangles_and_nulls(gpk, oracle):
    polys = _with_nulls(synth.clustered_polygons(3000, seed=2))
    env = GeoSeries(polys).envelope().array
    b = oracle.bounds(polys)
    assert env.geom_type == _abi.GEOM_POLYGON and env.n_coords == 5 * len(polys)
    r = env.xy.reshape(-1, 5, 2)
    ok = polys.is_valid()
    exp = np.stack([b[:, [0, 1]], b[:, [2, 1]], b[:, [2, 3]], b[:, [0, 3]], b[:, [0, 1]]], axis=1)
    assert np.array_equal(r[ok], exp[ok])
    assert np.array_equal(env.is_valid(), ok)
    with_empty = GeoArrowArray.from_polygons([[[(0, 0), (2, 0), (2, 1)]], [], [[(5, 5), (6, 5), (6, 7)]]])
    e2 = GeoSeries(with_empty).envelope().array
    assert e2.is_valid().tolist() == [True, False, True] and e2.xy[10:15].tolist() == [[5, 5], [6, 5], [6, 7], [5, 7], [5, 5]]
    pts = GeoSeries(synth.uniform_points(50))
    assert np.array_equal(pts.envelope().array.xy, pts.array.xy)  # the envelope of a point is the point


def test_centroid_hull_exterior_keep_nulls(gpk):
    polys = _with_nulls(synth.star_polygons(200, 12), 4)
    s = GeoSeries(polys)
    ok = polys.is_valid()
    assert np.array_equal(s.centroid().array.is_valid(), ok)
    assert np.array_equal(s.convex_hull().array.is_valid(), ok)
    ext = s.exterior().array
    assert np.array_equal(ext.is_valid(), ok) and np.all(np.diff(ext.geom_offsets)[~ok] == 0) and np.all(np.diff(ext.geom_offsets)[ok] == 13)
    assert np.array_equal(s.geom_type(), np.where(ok, 3, -1))
    assert not s.is_empty()[~ok].any()


@pytest.mark.parametrize("origin", ["centroid", "center", (3.0, -2.0)])
def test_rotate_scale_skew_matrices_are_built_like_the_reference_formulas(gpk, oracle, origin):
    """gpk_affine_about_origin == affine_transform with the matrix of geoseries.rs:85-139 written out on the host, per row,
    bit for bit (same expression order, no contraction)."""
    a = synth.clustered_polygons(500, seed=9)
    s = GeoSeries(a)
    if origin == "centroid":
        o = s.centroid().array.xy
    elif origin == "center":
        b = s.bounds()
        o = np.stack([(b[:, 0] + b[:, 2]) / 2.0, (b[:, 1] + b[:, 3]) / 2.0], axis=1)
    else:
        o = np.tile(np.array([origin]), (len(a), 1))
    t = math.radians(33.0)
    c, sn = math.cos(t), math.sin(t)
    tx, ty = math.tan(math.radians(12.0)), math.tan(math.radians(-7.0))
    z = np.zeros(len(o))
    mats = {
        "rotate": np.stack([z + c, z - sn, o[:, 0] - c * o[:, 0] + sn * o[:, 1], z + sn, z + c, o[:, 1] - sn * o[:, 0] - c * o[:, 1]], axis=1),
        "scale": np.stack([z + 1.5, z, o[:, 0] * (1 - 1.5), z, z + 0.25, o[:, 1] * (1 - 0.25)], axis=1),
        "skew": np.stack([z + 1.0, z + tx, -o[:, 1] * tx, z + ty, z + 1.0, -o[:, 0] * ty], axis=1),
    }
    got = {"rotate": s.rotate(33.0, origin), "scale": s.scale(1.5, 0.25, origin), "skew": s.skew(12.0, -7.0, origin)}
    rows = np.repeat(np.arange(len(a)), np.diff(a.ring_offsets)[a.geom_offsets[:-1]])  # one ring per polygon here
    for name, m in mats.items():
        mm = m[rows]
        exp = np.stack([mm[:, 0] * a.xy[:, 0] + mm[:, 1] * a.xy[:, 1] + mm[:, 2], mm[:, 3] * a.xy[:, 0] + mm[:, 4] * a.xy[:, 1] + mm[:, 5]], axis=1)
        assert np.array_equal(got[name].array.xy, exp), name
    with pytest.raises(ValueError):
        s.rotate(1.0, "middle")


def test_translate_of_the_reference_bench(gpk):
    """benches/affine.rs:23-26: translate(10, 10)"""
    pts = synth.uniform_points(202)
    assert np.array_equal(GeoSeries(pts).translate(10.0, 10.0).array.xy, pts.xy + 10.0)


def test_zero_row_columns_and_columns_of_empty_geometries(gpk):
    """a zero-length series returns zero-length results (the C entry points accept a NULL output when there is nothing to write);
    rotate / scale / skew over rows that hold no coordinate leave them as they are"""
    from geopolars_amd.geoseries import RowMap

    for empty in (GeoArrowArray.from_polygons([]), GeoArrowArray.from_linestrings([])):
        s = GeoSeries(empty)
        assert len(s.geom_type()) == 0 and len(s.is_empty()) == 0 and len(s.envelope()) == 0
        assert len(s.rotate(30.0)) == 0 and len(s.scale(2.0, 3.0)) == 0 and len(s.skew(10.0, 5.0, origin="centroid")) == 0
    ls = GeoSeries(GeoArrowArray.from_linestrings([]))
    assert len(ls.is_ring()) == 0 and len(ls.geodesic_length("haversine")) == 0 and len(ls.euclidean_length()) == 0
    hollow = GeoSeries(GeoArrowArray.from_polygons([[], [], []]))  # three rows, no ring, no coordinate
    assert hollow.is_empty().tolist() == [True, True, True]
    for t in (hollow.rotate(45.0), hollow.scale(2.0, 2.0, origin="centroid"), hollow.skew(1.0, 2.0, origin=(0.0, 0.0))):
        assert len(t) == 3 and t.array.n_coords == 0
    lines = GeoSeries(synth.random_linestrings(50))
    m = RowMap(lines, np.zeros(0, dtype=np.uint32))  # an empty batch builds an empty map
    pts = GeoSeries(GeoArrowArray.from_points(np.zeros((0, 2))))
    assert len(pts.distance(lines, row_map=m)) == 0
