"""Streaming reductions on MIXED columns (sequences of every size class in one column) against the oracle — sequence
lengths around every boundary of the size classes and the chunking (0, 1, 2, 3 coordinates; 16 / 17; 63 / 64 / 65; 128 / 129;
512 / 513 = the long class; runs of more than 64 empty and of one-coordinate sequences; open rings; a long sequence first,
last, and between short ones), for every geometry family and operator, including the direct-result forms of columns whose
sequences are their geometries."""
import numpy as np
import pytest

from geopolars_amd import _abi
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

pytestmark = pytest.mark.gpu

LENGTHS = [0, 1, 2, 3, 4, 5, 7, 9, 15, 16, 17, 31, 33, 63, 64, 65, 66, 100, 127, 128, 129, 200, 511, 512, 513, 514, 700, 1500]


def _seq_lengths(rng, n):
    """a few thousand lengths: mostly short (power-law-ish), every boundary length sprinkled in, runs of empties"""
    base = np.minimum((rng.pareto(1.3, n) * 5).astype(np.int64) + 3, 3000)
    special = rng.choice(LENGTHS, size=n)
    pick = rng.uniform(size=n) < 0.3
    out = np.where(pick, special, base)
    out[:3] = [600, 0, 5]  # a long sequence first
    out[100:180] = 0  # more than 64 empty sequences in a row
    out[300:420] = 1  # more than 64 one-coordinate sequences in a row
    out[-1] = 900  # a long sequence last
    return out


def _ring(rng, n, centre, closed=True):
    if n == 0:
        return np.zeros((0, 2))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rng.uniform(0.5, 1.0, n)
    xy = centre + np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
    if closed and n >= 2:
        xy[-1] = xy[0]
    return xy


def _columns():
    rng = np.random.default_rng(99)
    cols = {}
    lens = _seq_lengths(rng, 4000)
    centres = rng.uniform(-50, 50, (len(lens), 2)) + [1000.0, -2000.0]  # far from the origin: the shift by the first vertex matters
    open_ring = rng.uniform(size=len(lens)) < 0.05
    rings = [_ring(rng, int(n), c, closed=not o) for n, c, o in zip(lens, centres, open_ring)]
    xy = np.concatenate(rings) if rings else np.zeros((0, 2))
    ring_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    # polygons: 1-4 rings each (holes), some polygons empty
    k = rng.integers(0, 5, 2000)
    geom_off = np.minimum(np.concatenate([[0], np.cumsum(k)]), len(lens)).astype(np.int32)
    geom_off[-1] = len(lens)
    cols["polygons"] = GeoArrowArray(_abi.GEOM_POLYGON, xy, geom_offsets=geom_off, ring_offsets=ring_off)
    cols["multilinestrings"] = GeoArrowArray(_abi.GEOM_MULTILINESTRING, xy, geom_offsets=geom_off, ring_offsets=ring_off)
    cols["linestrings"] = GeoArrowArray(_abi.GEOM_LINESTRING, xy, geom_offsets=ring_off)  # one sequence per row: the direct-result forms
    cols["multipoints"] = GeoArrowArray(_abi.GEOM_MULTIPOINT, xy, geom_offsets=ring_off)
    # multipolygons: parts of 1-3 rings, geometries of 0-4 parts
    pk = rng.integers(1, 4, 3000)
    part_off = np.minimum(np.concatenate([[0], np.cumsum(pk)]), len(lens)).astype(np.int32)
    part_off = np.unique(part_off)
    if part_off[-1] != len(lens):
        part_off = np.append(part_off, len(lens)).astype(np.int32)
    n_parts = len(part_off) - 1
    gk = rng.integers(0, 5, 1500)
    g_off = np.minimum(np.concatenate([[0], np.cumsum(gk)]), n_parts).astype(np.int32)
    g_off[-1] = n_parts
    cols["multipolygons"] = GeoArrowArray(_abi.GEOM_MULTIPOLYGON, xy, geom_offsets=g_off, part_offsets=part_off.astype(np.int32), ring_offsets=ring_off)
    # single-ring polygons of mixed sizes: the one-to-one direct-result forms on a mixed column
    cols["single_ring_polygons"] = GeoArrowArray(_abi.GEOM_POLYGON, xy, geom_offsets=np.arange(len(lens) + 1, dtype=np.int32), ring_offsets=ring_off)
    return cols


def _close(got, exp, tol=1e-9):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    m = ~np.isnan(exp)
    assert np.array_equal(got[m] == 0.0, exp[m] == 0.0)  # zero / non-zero exact
    scale = np.maximum(np.abs(exp[m]), 1e-300)
    assert np.all(np.abs(got[m] - exp[m]) <= tol * scale), float(np.max(np.abs(got[m] - exp[m]) / scale))


@pytest.mark.parametrize("name", ["polygons", "multilinestrings", "linestrings", "multipoints", "multipolygons", "single_ring_polygons"])
def test_mixed_column_reductions(gpk, oracle, name):
    a = _columns()[name]
    s = GeoSeries(a)
    _close(s.area(), oracle.area(a))
    _close(s.signed_area(), oracle.area(a, signed=True))
    _close(s.euclidean_length(), oracle.euclidean_length(a))
    assert np.array_equal(s.bounds(), oracle.bounds(a), equal_nan=True)  # min / max: bit exact
    exp_c, _ = oracle.centroid(a)
    # centroid = ratio of sums that nearly cancel for rings far from the origin: compare relative to the coordinates' magnitude
    got_c = s.centroid().array.xy
    assert np.array_equal(np.isnan(got_c), np.isnan(exp_c))
    inf = np.isinf(exp_c)  # weights that cancel exactly (a hole as large as its exterior): upstream divides by zero too
    assert np.array_equal(got_c[inf], exp_c[inf])
    m = ~np.isnan(exp_c) & ~inf
    assert np.all(np.abs(got_c[m] - exp_c[m]) <= 1e-9 * np.maximum(np.abs(exp_c[m]), 1.0))


def test_mixed_column_results_do_not_depend_on_the_neighbours(gpk):
    """bounds of a polygon are bit-identical whether it sits in the mixed column or alone"""
    a = _columns()["polygons"]
    whole = GeoSeries(a).bounds()
    ro, go = a.ring_offsets, a.geom_offsets
    for g in (0, 1, 57, 1999):
        r0, r1 = go[g], go[g + 1]
        if r1 == r0:
            assert np.all(np.isnan(whole[g]))
            continue
        xy = a.xy[ro[r0] : ro[r1]]
        one = GeoArrowArray(_abi.GEOM_POLYGON, xy, geom_offsets=np.array([0, r1 - r0], dtype=np.int32), ring_offsets=(ro[r0 : r1 + 1] - ro[r0]).astype(np.int32))
        assert np.array_equal(GeoSeries(one).bounds()[0], whole[g], equal_nan=True)
