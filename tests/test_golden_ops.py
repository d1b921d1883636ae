"""tests/golden/ops_lattice.npz — answers computed with integer / rational arithmetic by tests/golden/make_ops_golden.py,
independent of the oracle and of the library: the oracle (CPU) and the HIP path (GPU) are both held against the file."""
import os

import numpy as np
import pytest

from geopolars_amd import _abi
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    z = np.load(os.path.join(HERE, "golden", "ops_lattice.npz"))
    polys = GeoArrowArray(_abi.GEOM_POLYGON, z["xy"], geom_offsets=z["geom_offsets"], ring_offsets=z["ring_offsets"])
    return z, polys, GeoArrowArray.from_points(z["points"])


def _canon(ring):
    ring = ring[:-1] if len(ring) > 1 and np.array_equal(ring[0], ring[-1]) else ring
    return np.roll(ring, -np.lexsort((ring[:, 1], ring[:, 0]))[0], axis=0)


def _check(z, area, centroid, hull_xy, hull_off, contains, intersects, within, distance, rtol):
    if rtol < 1e-10:
        assert np.array_equal(area, z["area"])  # halves of integers: exact whatever the summation order
    assert np.allclose(area, z["area"], rtol=rtol, atol=0)
    assert np.allclose(centroid, z["centroid"], rtol=rtol, atol=rtol)
    assert np.array_equal(hull_off, z["hull_offsets"])
    for g in range(len(area)):
        lo, hi = hull_off[g], hull_off[g + 1]
        assert np.array_equal(_canon(hull_xy[lo:hi]), _canon(z["hull_xy"][lo:hi])), g
    pos = z["position"]
    assert np.array_equal(contains, pos > 0) and np.array_equal(within, pos > 0) and np.array_equal(intersects, pos >= 0)
    assert np.array_equal(distance == 0, z["distance"] == 0)
    assert np.allclose(distance, z["distance"], rtol=rtol, atol=0)


def test_oracle_reproduces_the_ops_golden(oracle):
    z, polys, pts = _load()
    c, valid = oracle.centroid(polys)
    assert valid.all()
    hx, ho = oracle.convex_hull(polys)
    _check(
        z, oracle.area(polys), c, hx, ho,
        oracle.predicate_rowwise(polys, pts, "contains").astype(bool), oracle.predicate_rowwise(polys, pts, "intersects").astype(bool),
        oracle.predicate_rowwise(pts, polys, "within").astype(bool), oracle.distance_rowwise(pts, polys), rtol=1e-12,
    )


@pytest.mark.gpu
def test_hip_path_reproduces_the_ops_golden(gpk):
    z, polys, pts = _load()
    s, p = GeoSeries(polys), GeoSeries(pts)
    h = s.convex_hull().array
    _check(z, s.area(), s.centroid().array.xy, h.xy, h.ring_offsets, s.contains(p), s.intersects(p), p.within(s), p.distance(s), rtol=1e-9)


def _load_join():
    z = np.load(os.path.join(HERE, "golden", "join_lattice.npz"))
    polys = GeoArrowArray(_abi.GEOM_POLYGON, z["xy"], geom_offsets=z["geom_offsets"], ring_offsets=z["ring_offsets"])
    return polys, GeoArrowArray.from_points(z["points"]), z["pairs"]


def test_oracle_reproduces_the_join_golden(oracle):
    """tests/golden/join_lattice.npz: point-in-polygon pairs from an even-odd ring walk in Python integers (1251 point /
    polygon incidences lie exactly on a boundary and must be rejected, KA-1)"""
    polys, pts, exp = _load_join()
    for mode in (0, 1):  # brute force and grid directory
        pairs, counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=mode)
        assert np.array_equal(pairs, exp) and np.array_equal(counts, np.bincount(exp[:, 0], minlength=len(pts)))
    back, _, _ = oracle.spatial_join(polys, pts, "contains", mode=0)  # polygon on the left: the transposed pairs
    assert np.array_equal(back[np.lexsort((back[:, 0], back[:, 1]))][:, ::-1], exp)


@pytest.mark.gpu
def test_hip_join_reproduces_the_join_golden(gpk):
    from geopolars_amd.spatial_index import join_pairs

    polys, pts, exp = _load_join()
    pairs, counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    assert np.array_equal(pairs, exp) and np.array_equal(counts, np.bincount(exp[:, 0], minlength=len(pts)))
    back, _ = join_pairs(GeoSeries(polys), GeoSeries(pts), "contains")
    assert np.array_equal(back[np.lexsort((back[:, 0], back[:, 1]))][:, ::-1], exp)


def _load_lines():
    z = np.load(os.path.join(HERE, "golden", "lines_lattice.npz"))
    return z, GeoArrowArray(_abi.GEOM_LINESTRING, z["xy"], geom_offsets=z["geom_offsets"]), GeoArrowArray.from_points(z["points"])


def _check_lines(z, length, centroid, bounds, contains, within, distance, rtol):
    ok = z["centroid_valid"]
    assert np.allclose(length, z["length"], rtol=rtol, atol=0)
    assert np.isnan(centroid[~ok]).all() and np.allclose(centroid[ok], z["centroid"][ok], rtol=rtol, atol=rtol)
    assert np.array_equal(bounds, z["bounds"], equal_nan=True)
    assert np.array_equal(contains, z["contains"]) and np.array_equal(within, z["contains"])
    assert not np.isnan(distance).any() and np.array_equal(distance == 0, z["distance"] == 0)
    assert np.allclose(distance, z["distance"], rtol=rtol, atol=0)


def test_oracle_reproduces_the_lines_golden(oracle):
    z, lines, pts = _load_lines()
    c, valid = oracle.centroid(lines)
    assert np.array_equal(valid, z["centroid_valid"])
    c = np.where(valid[:, None], c, np.nan)
    _check_lines(z, oracle.euclidean_length(lines), c, oracle.bounds(lines), oracle.predicate_rowwise(lines, pts, "contains").astype(bool),
                 oracle.predicate_rowwise(pts, lines, "within").astype(bool), oracle.distance_rowwise(pts, lines), rtol=1e-12)


@pytest.mark.gpu
def test_hip_path_reproduces_the_lines_golden(gpk):
    z, lines, pts = _load_lines()
    s, p = GeoSeries(lines), GeoSeries(pts)
    _check_lines(z, s.euclidean_length(), s.centroid().array.xy, s.bounds(), s.contains(p), p.within(s), p.distance(s), rtol=1e-9)
