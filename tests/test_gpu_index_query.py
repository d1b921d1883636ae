"""GPU parity of gpk_index_query_envelope — a query on the index by itself, which is what the reference's own index tests do:
`spatial_index.r_tree.locate_in_envelope(&AABB::from_corners([0.0, 0.0], [20.0, 20.0]))` (spatial_index.rs:383-393 over points,
:422-429 over polygons; rstar: every leaf CONTAINED in the query box, closed intervals) and `locate_in_envelope_intersecting` — through
the C ABI, against the reference's known answers (KA-2, KA-3, replayed verbatim) and against the brute-force restatement
(oracle/pyoracle.py: envelope_query) on random boxes, bit-exact on pairs and counts."""
import ctypes as C

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import MEM_DEVICE, MEM_HOST, SpatialIndex

pytestmark = pytest.mark.gpu

KA_POINTS = [(0.0, 10.0), (1.0, 1.0), (10.0, 0.0), (1.0, -1.0), (0.0, -10.0), (-1.0, -1.0), (-10.0, 0.0), (-1.0, 1.0), (0.0, 10.0)]


def test_ka2_spatial_index_points_replayed(gpk):
    """spatial_index.rs:361-395: nine points, locate_in_envelope([0, 0] - [20, 20]) holds exactly {0, 1, 2, 8} — (10, 0) and the two
    (0, 10) lie ON the query box's border: closed intervals"""
    index = SpatialIndex(GeoSeries(GeoArrowArray.from_points(KA_POINTS)), for_points=False)
    indexes = index.locate_in_envelope((0.0, 0.0), (20.0, 20.0)).tolist()
    assert 0 in indexes and 1 in indexes and 2 in indexes and 8 in indexes
    assert len(indexes) == 4


def test_ka3_spatial_index_polygons_replayed(gpk):
    """spatial_index.rs:397-430: two squares sharing the corner (0, 0); only polygon 0 lies inside [0, 0] - [20, 20]; both MEET it"""
    polys = GeoArrowArray.from_polygons(
        [[[(0.0, 0.0), (10.0, 0.0), (10.0, 10.0), (0.0, 10.0)]], [[(0.0, 0.0), (-10.0, 0.0), (-10.0, -10.0), (0.0, -10.0)]]]
    )
    for for_points in (True, False):
        index = SpatialIndex(GeoSeries(polys), for_points=for_points)
        indexes = index.locate_in_envelope((0.0, 0.0), (20.0, 20.0)).tolist()
        assert 0 in indexes and len(indexes) == 1
        assert index.locate_in_envelope_intersecting((0.0, 0.0), (20.0, 20.0)).tolist() == [0, 1]
        assert index.locate_in_envelope_intersecting((0.5, 0.5), (20.0, 20.0)).tolist() == [0]


def _boxes(rng, n, lo, hi):
    c = rng.uniform(lo, hi, (n, 2))
    half = rng.uniform(0.0, 0.2 * (hi - lo), (n, 2)) * rng.choice([0.02, 0.2, 1.0], (n, 1))
    return np.column_stack([c - half, c + half])


@pytest.mark.parametrize("shape", ["polygons", "points", "lines", "multipolygons_with_nulls"])
def test_random_boxes_against_the_brute_force_restatement(gpk, oracle, shape):
    rng = np.random.default_rng(11)
    if shape == "polygons":
        arr = synth.star_polygons(3000, 24)
    elif shape == "points":
        arr = synth.uniform_points(20_000, seed=3)
    elif shape == "lines":
        arr = synth.random_linestrings(4000, seed=5, max_log2=6.0)
    else:
        mps = []
        for i in range(1500):
            parts = []
            for _ in range(int(rng.integers(0, 4))):  # (some rows are EMPTY multipolygons: no leaf)
                x, y, w = rng.uniform(0, 990), rng.uniform(0, 990), rng.uniform(0.5, 9.0)
                parts.append([[(x, y), (x + w, y), (x + w, y + w), (x, y + w)]])
            mps.append(parts)
        arr = GeoArrowArray.from_multipolygons(mps)
        arr.validity = np.packbits(rng.uniform(size=len(arr)) > 0.1, bitorder="little")
    boxes = np.concatenate(
        [
            _boxes(rng, 400, 0.0, 1000.0),
            [[-1e9, -1e9, 1e9, 1e9]],  # everything (a row long enough for the segmented sort)
            [[2000.0, 2000.0, 3000.0, 3000.0]],  # outside the extent
            [[500.0, 500.0, 500.0, 500.0]],  # a degenerate box
            [[np.nan, np.nan, np.nan, np.nan]],  # an empty query
            [[600.0, 600.0, 400.0, 400.0]],  # corners the wrong way round: AABB::from_corners orders them
        ]
    )
    index = SpatialIndex(GeoSeries(arr), for_points=False)
    for mode in ("contained", "intersecting"):
        ep, ec = oracle.envelope_query(arr, boxes, mode)
        gp, gc = index.query_envelopes(boxes, mode)
        assert np.array_equal(gc, ec), mode
        assert np.array_equal(gp, ep), mode
    assert ec[400] == int((arr.is_valid() & ~np.isnan(oracle.bounds(arr)[:, 0])).sum())


def test_device_buffers_capacity_and_count_only(gpk, oracle):
    import torch

    lib = _abi.lib()
    arr = synth.star_polygons(2000, 16)
    index = SpatialIndex(GeoSeries(arr), for_points=False)
    boxes = _boxes(np.random.default_rng(2), 1000, 0.0, 1000.0)
    ep, ec = oracle.envelope_query(arr, boxes, "intersecting")
    dev = torch.device("cuda", 0)
    bd = torch.from_numpy(boxes).to(dev)
    counts = torch.zeros(len(boxes), dtype=torch.int32, device=dev)
    pairs = torch.zeros((len(ep) + 10, 2), dtype=torch.int32, device=dev)
    n = C.c_int64(0)
    st = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.gpk_index_query_envelope(index.handle, bd.data_ptr(), len(boxes), _abi.QUERY_INTERSECTING, counts.data_ptr(), pairs.data_ptr(), len(pairs), C.byref(n), MEM_DEVICE, st))
    torch.cuda.synchronize()
    assert n.value == len(ep)
    assert np.array_equal(counts.cpu().numpy().view(np.uint32), ec)
    assert np.array_equal(pairs[: len(ep)].cpu().numpy().view(np.uint32), ep)
    # count only; a buffer that is too small reports the total
    n.value = 0
    _abi.check(lib.gpk_index_query_envelope(index.handle, boxes.ctypes.data, len(boxes), _abi.QUERY_INTERSECTING, None, None, 0, C.byref(n), MEM_HOST, None))
    assert n.value == len(ep)
    small = np.zeros((5, 2), dtype=np.uint32)
    rc = lib.gpk_index_query_envelope(index.handle, boxes.ctypes.data, len(boxes), _abi.QUERY_INTERSECTING, None, small.ctypes.data, 5, C.byref(n), MEM_HOST, None)
    assert rc == _abi.GPK_ERR_CAPACITY and n.value == len(ep)
    assert lib.gpk_index_query_envelope(index.handle, boxes.ctypes.data, len(boxes), 7, None, None, 0, C.byref(n), MEM_HOST, None) == _abi.GPK_ERR_INVALID_ARGUMENT
    # no boxes
    _abi.check(lib.gpk_index_query_envelope(index.handle, None, 0, _abi.QUERY_CONTAINED, None, None, 0, C.byref(n), MEM_HOST, None))
    assert n.value == 0
