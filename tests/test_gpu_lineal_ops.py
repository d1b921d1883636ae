"""GPU parity: geodesic_length (haversine / vincenty) and simplify through the C ABI against the CPU oracle — lengths within 1e-9
relative, simplified coordinates and offsets bit-exact (the kernel walks geo's recursion with the oracle's own distance
expression)."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

pytestmark = pytest.mark.gpu


def _lonlat_lines(n, seed, lo=2, hi=200):
    rng = np.random.default_rng(seed)
    nv = rng.integers(lo, hi, n)
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum(nv)
    start = np.stack([rng.uniform(-170, 170, n), rng.uniform(-80, 80, n)], axis=1)
    step = rng.normal(0, 0.05, (int(off[-1]), 2))
    step[off[:-1]] = start
    xy = np.cumsum(step, axis=0)
    xy -= np.repeat(xy[off[:-1]] - start, nv, axis=0)
    xy[:, 1] = np.clip(xy[:, 1], -89.0, 89.0)
    return GeoArrowArray(_abi.GEOM_LINESTRING, xy, geom_offsets=off)


@pytest.mark.parametrize("method", ["geodesic", "haversine", "vincenty"])
def test_geodesic_length_parity(gpk, oracle, method):
    lines = _lonlat_lines(20_000, 3)
    got, exp = GeoSeries(lines).geodesic_length(method), oracle.geodesic_length(lines, method)
    assert np.all(np.abs(got - exp) <= 1e-9 * np.abs(exp))  # the north star's tolerance for f64 measures
    polys = synth.clustered_polygons(5000, seed=4, domain=60.0)  # (lon, lat) inside +-60 degrees
    got, exp = GeoSeries(polys).geodesic_length(method), oracle.geodesic_length(polys, method)
    assert np.all(np.abs(got - exp) <= 1e-9 * np.abs(exp))
    mp = synth.powerlaw_multipolygons(3000, domain=80.0)
    keep = np.ones(len(mp), dtype=bool)
    keep[::9] = False
    mp.validity = np.packbits(keep, bitorder="little")
    got, exp = GeoSeries(mp).geodesic_length(method), oracle.geodesic_length(mp, method)
    assert np.array_equal(np.isnan(got), ~keep) and np.all(np.abs(got[keep] - exp[keep]) <= 1e-9 * np.abs(exp[keep]))
    assert np.array_equal(GeoSeries(synth.uniform_points(10)).geodesic_length(method), np.zeros(10))


def _pairs_as_lines(l1, p1, l2, p2):
    n = len(l1)
    xy = np.empty((2 * n, 2))
    xy[0::2, 0], xy[0::2, 1], xy[1::2, 0], xy[1::2, 1] = l1, p1, l2, p2
    return GeoArrowArray(_abi.GEOM_LINESTRING, xy, geom_offsets=np.arange(0, 2 * n + 1, 2, dtype=np.int32))


def test_karney_kernel_against_an_independent_solver(gpk):
    """the HIP kernel (csrc/gpk_karney.h) against oracle/geodesic_quadrature.py — Bessel's reduction evaluated by Gauss-Legendre
    quadrature, the azimuth found by bisection: no series, no Newton step, nothing shared with the kernel or with the C oracle
    (which is the kernel's algorithm written a second time).  Random pairs over the whole ellipsoid, nearly antipodal pairs at
    three scales, short lines, meridians, the equator, the poles."""
    from oracle import geodesic_quadrature as gq

    rng = np.random.default_rng(41)
    sets = []
    n = 12_000
    sets.append((rng.uniform(-180, 180, n), rng.uniform(-90, 90, n), rng.uniform(-180, 180, n), rng.uniform(-90, 90, n)))
    for spread in (0.5, 0.01, 1e-4):  # nearly antipodal: the regime Karney's paper is about (Vincenty's iteration fails here)
        m = 3_000
        p1, l1 = rng.uniform(-75, 75, m), rng.uniform(-180, 180, m)
        sets.append((l1, p1, l1 + 180.0 + rng.normal(0, spread, m), -p1 + rng.normal(0, spread, m)))
    m = 2_000
    p1, l1 = rng.uniform(-89, 89, m), rng.uniform(-180, 180, m)
    sets.append((l1, p1, l1 + rng.normal(0, 1e-3, m), p1 + rng.normal(0, 1e-3, m)))  # short lines
    sets.append((l1, p1, l1, rng.uniform(-90, 90, m)))  # meridians
    sets.append((l1, p1, l1 + 180.0, rng.uniform(-90, 90, m)))  # over a pole
    sets.append((l1, np.zeros(m), rng.uniform(-180, 180, m), np.zeros(m)))  # along (or, past (1 - f) * 180 degrees, off) the equator
    sets.append((l1, np.full(m, 90.0), rng.uniform(-180, 180, m), rng.uniform(-90, 90, m)))  # from the pole
    for l1, p1, l2, p2 in sets:
        p2 = np.clip(p2, -90.0, 90.0)
        got = GeoSeries(_pairs_as_lines(l1, p1, l2, p2)).geodesic_length("geodesic")
        exp = gq.inverse_distance(l1, p1, l2, p2)
        assert np.all(np.abs(got - exp) <= 1e-9 * exp + 2e-8)  # (1e-9 relative: the north star's tolerance; 20 nm absolute for millimetre lines)


def test_geodesic_published_values_and_methods(gpk):
    nyc_london = GeoSeries(GeoArrowArray.from_linestrings([[(-74.006, 40.7128), (-0.1278, 51.5074)]]))
    assert round(float(nyc_london.geodesic_length("haversine")[0])) == 5_570_230  # geo's HaversineLength doc example
    assert round(float(nyc_london.geodesic_length("vincenty")[0])) == 5_585_234  # geo's VincentyLength doc example
    assert round(float(nyc_london.geodesic_length("geodesic")[0])) == 5_585_234  # geo's GeodesicDistance doc example
    assert round(float(nyc_london.geodesic_length()[0])) == 5_585_234  # ("geodesic" is the default: georust/geoseries.py:128)
    three = GeoSeries(GeoArrowArray.from_linestrings([[(-74.006, 40.7128), (-0.1278, 51.5074), (135.5244559, 34.687455)]]))
    assert round(float(three.geodesic_length("geodesic")[0])) == 15_109_158  # geo's GeodesicLength doc example
    published = GeoSeries(GeoArrowArray.from_linestrings([[(174.81, -41.32), (-5.50, 40.96)], [(-73.8, 40.6), (-0.5, 51.6)], [(0.0, 0.0), (179.5, 0.5)], [(0.0, 0.0), (180.0, 0.0)]]))
    got = published.geodesic_length("geodesic")  # GeographicLib: Wellington -> Salamanca, JFK -> LHR; Karney 2013's antipodal pair
    assert np.allclose(got, [19959679.26735382, 5551759.4003186841, 19936288.578965, 20003931.458625], rtol=0, atol=2e-3)
    with pytest.raises(ValueError):
        nyc_london.geodesic_length("rhumb")


def _same(a: GeoArrowArray, xy, off):
    inner = a.ring_offsets if a.ring_offsets is not None else a.geom_offsets
    return np.array_equal(a.xy, xy) and np.array_equal(inner, off)


@pytest.mark.parametrize("eps", [0.0, 0.05, 0.8, 4.0, 50.0])
def test_simplify_parity(gpk, oracle, eps):
    for arr in (
        synth.random_linestrings(4000),  # 5..257 vertices: the 64-lane kernel
        synth.clustered_polygons(6000, seed=8),  # short rings: the 8-lane kernel, INITIAL_MIN = 4
        synth.powerlaw_multipolygons(2500),  # ragged rings up to thousands of vertices, holes
    ):
        got = GeoSeries(arr).simplify(eps).array
        xy, off = oracle.simplify(arr, eps)
        assert _same(got, xy, off)
        assert got.geom_type == arr.geom_type and np.array_equal(got.geom_offsets if arr.ring_offsets is not None else off, got.geom_offsets)
    if eps == 0.0:
        assert np.array_equal(got.xy, arr.xy)


def test_simplify_long_ring_and_degenerate_sequences(gpk, oracle):
    t = np.linspace(0, 2 * np.pi, 60_001)
    ring = np.stack([100 * np.cos(t) + 3 * np.cos(37 * t), 100 * np.sin(t) + 3 * np.sin(53 * t)], axis=1)
    ring[-1] = ring[0]
    poly = GeoArrowArray(_abi.GEOM_POLYGON, ring, geom_offsets=np.array([0, 1], np.int32), ring_offsets=np.array([0, len(ring)], np.int32))
    for eps in (0.01, 0.5, 10.0, 1000.0):
        got = GeoSeries(poly).simplify(eps).array
        xy, off = oracle.simplify(poly, eps)
        assert _same(got, xy, off) and off[-1] >= 4
    ls = GeoArrowArray.from_linestrings([[(0, 0), (1, 1)], [(5, 5)], [], [(0, 0), (2, 3), (4, 3), (6, 0)], [(0, 0), (1, 0), (2, 0), (3, 0)]])
    got = GeoSeries(ls).simplify(1.0).array
    xy, off = oracle.simplify(ls, 1.0)
    assert _same(got, xy, off)
    assert got.geom_offsets.tolist() == [0, 2, 3, 3, 7, 9]
    pts = GeoSeries(synth.uniform_points(5))
    assert np.array_equal(pts.simplify(1.0).array.xy, pts.array.xy)
