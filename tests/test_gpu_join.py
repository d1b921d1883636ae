"""GPU parity: point-in-polygon spatial join through the C ABI vs the CPU
oracle, bit-exact on counts and sorted (l, r) pairs."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs

pytestmark = pytest.mark.gpu

KA1_POINTS = [(0.0, 10.0), (1.0, 1.0), (10.0, 1.0), (1.0, -1.0), (0.0, -10.0), (-1.0, -1.0), (-10.0, 0.0), (-1.0, 1.0), (0.0, 10.0)]
KA1_SQUARE = [[[(0.0, 0.0), (20.0, 0.0), (20.0, 20.0), (0.0, 20.0)]]]


def test_ka1_boundary_is_not_contained(gpk):
    """spatial_join_test, spatial_index.rs:432-484: inner join has exactly 2 rows -> hits {1, 2};
    both (0, 10) points lie on the edge x = 0 and must be rejected."""
    pts = GeoSeries(GeoArrowArray.from_points(KA1_POINTS))
    poly = GeoSeries(GeoArrowArray.from_polygons(KA1_SQUARE))
    pairs, counts = join_pairs(pts, poly, "intersects")
    assert pairs.tolist() == [[1, 0], [2, 0]]
    assert counts.tolist() == [0, 1, 1, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("n_points,n_polys,n_verts", [(1, 1, 3), (257, 7, 5), (50_000, 1000, 64), (200_000, 300, 17)])
def test_c2_parity(gpk, oracle, n_points, n_polys, n_verts):
    polys = synth.star_polygons(n_polys, n_verts)
    pts = synth.uniform_points(n_points)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)


def test_adversarial_points(gpk, oracle):
    """vertices, edge midpoints, points level with vertices, bbox corners: every degenerate arm of
    coord_pos_relative_to_ring, including the exact-arithmetic fallback."""
    polys = synth.star_polygons(64, 64)
    pts = synth.adversarial_points(polys)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "contains", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "contains")
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)


def test_axis_aligned_and_collinear(gpk, oracle):
    """integer lattice points against rectangles / an L-shape with horizontal and vertical edges and
    a hole: on-edge, on-vertex, collinear-with-horizontal-edge cases are all exact."""
    polys = GeoArrowArray.from_polygons(
        [
            [[(0, 0), (4, 0), (4, 4), (0, 4)], [(1, 1), (1, 3), (3, 3), (3, 1)]],
            [[(5, 0), (9, 0), (9, 2), (7, 2), (7, 4), (5, 4)]],
            [[(2, 2), (6, 2), (6, 6), (2, 6)]],  # overlaps the other two: multi-hit rows
        ]
    )
    gx, gy = np.meshgrid(np.arange(-1, 11, 0.5), np.arange(-1, 8, 0.5))
    pts = GeoArrowArray.from_points(np.stack([gx.ravel(), gy.ravel()], axis=1))
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    assert exp_counts.max() >= 2
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)


def test_multipolygons_with_holes(gpk, oracle):
    mp = synth.powerlaw_multipolygons(500)
    pts = synth.uniform_points(100_000)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, mp, "within", mode=1)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(mp), "within")
    assert len(exp_pairs) > 0
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)


def test_prebuilt_index_and_row_base(gpk, oracle):
    """spatial_join_test_with_precomputed_indexes (spatial_index.rs:558-624) + the row base used by
    row-sharded runs."""
    polys = GeoSeries(synth.star_polygons(100, 16))
    pts = synth.uniform_points(10_000)
    idx = SpatialIndex(polys)
    a, ca = join_pairs(GeoSeries(pts), polys, "intersects", r_index=idx)
    b, cb = join_pairs(GeoSeries(pts), polys, "intersects")
    assert np.array_equal(a, b) and np.array_equal(ca, cb)
    c, _ = join_pairs(GeoSeries(pts), polys, "intersects", r_index=idx, left_row_base=1000)
    assert np.array_equal(c[:, 0], a[:, 0] + 1000) and np.array_equal(c[:, 1], a[:, 1])


def test_empty_and_nan_inputs(gpk):
    polys = GeoSeries(synth.star_polygons(10, 8))
    empty = GeoSeries(GeoArrowArray.from_points(np.zeros((0, 2))))
    pairs, counts = join_pairs(empty, polys)
    assert pairs.shape == (0, 2) and counts.shape == (0,)
    nanpts = GeoSeries(GeoArrowArray.from_points([[np.nan, np.nan], [np.nan, 1.0]]))
    pairs, counts = join_pairs(nanpts, polys)
    assert pairs.shape == (0, 2) and counts.tolist() == [0, 0]
    # the polygon-left arm with an empty side
    pairs, counts = join_pairs(polys, empty)
    assert pairs.shape == (0, 2) and counts.tolist() == [0] * 10
    pairs, counts = join_pairs(polys, nanpts)
    assert pairs.shape == (0, 2) and counts.tolist() == [0] * 10
    lines = GeoSeries(GeoArrowArray.from_linestrings([[(0, 0), (1, 1)], []]))
    pairs, counts = join_pairs(lines, empty)
    assert pairs.shape == (0, 2) and counts.tolist() == [0, 0]


@pytest.mark.parametrize("n,neigh", [(500, 4.0), (6000, 12.0)])
def test_c4_polygon_polygon_intersects_join(gpk, oracle, n, neigh):
    """configs[3] at test scale: polygon x polygon intersects join (spatial_index.rs:102-104), sorted pairs."""
    a = synth.clustered_polygons(n, seed=21, mean_neighbours=neigh)
    b = synth.clustered_polygons(n + 37, seed=22, mean_neighbours=neigh)
    exp_pairs, exp_counts, _ = oracle.spatial_join(a, b, "intersects", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(a), GeoSeries(b), "intersects")
    assert len(exp_pairs) > n // 4
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)


def test_multipolygon_polygon_intersects_join(gpk, oracle):
    a = synth.powerlaw_multipolygons(800, seed=31)
    b = synth.clustered_polygons(900, seed=32, mean_neighbours=20.0)
    for l, r in ((a, b), (b, a)):  # spatial_index.rs:107-123
        exp_pairs, exp_counts, _ = oracle.spatial_join(l, r, "intersects", mode=1)
        got_pairs, got_counts = join_pairs(GeoSeries(l), GeoSeries(r), "intersects")
        assert len(exp_pairs) > 0
        assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)


def test_count_only_and_capacity_contract(gpk, oracle):
    """C ABI contract of gpk_spatial_join: count-only calls (no pair buffer) and a too-small pair buffer
    (GPK_ERR_CAPACITY with the exact total reported, the first `capacity` pairs written)."""
    import ctypes as C

    from geopolars_amd import _abi
    from geopolars_amd._abi import MEM_HOST, PREDICATES

    polys = GeoSeries(synth.star_polygons(50, 16))
    pts = GeoSeries(synth.uniform_points(30_000))
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts.array, polys.array, "intersects", mode=1)
    lib = _abi.lib()
    n_pairs = C.c_int64(-1)
    counts = np.empty(len(pts), dtype=np.uint32)
    rc = lib.gpk_spatial_join(pts.device().handle, polys.device().handle, None, PREDICATES["intersects"], 0, counts.ctypes.data, None, 0, C.byref(n_pairs), MEM_HOST, None)
    assert rc == _abi.GPK_OK and n_pairs.value == len(exp_pairs) and np.array_equal(counts, exp_counts)
    rc = lib.gpk_spatial_join(pts.device().handle, polys.device().handle, None, PREDICATES["intersects"], 0, None, None, 0, C.byref(n_pairs), MEM_HOST, None)
    assert rc == _abi.GPK_OK and n_pairs.value == len(exp_pairs)
    cap = len(exp_pairs) // 2
    small = np.zeros((cap, 2), dtype=np.uint32)
    rc = lib.gpk_spatial_join(pts.device().handle, polys.device().handle, None, PREDICATES["intersects"], 0, None, small.ctypes.data, cap, C.byref(n_pairs), MEM_HOST, None)
    assert rc == _abi.GPK_ERR_CAPACITY and n_pairs.value == len(exp_pairs)
    assert "capacity" in _abi.last_error()
    assert np.array_equal(small, exp_pairs[:cap])
    rc = lib.gpk_spatial_join(pts.device().handle, polys.device().handle, None, 99, 0, None, None, 0, C.byref(n_pairs), MEM_HOST, None)
    assert rc == _abi.GPK_ERR_INVALID_ARGUMENT


def test_join_from_threads(gpk, oracle):
    """the ABI is re-entrant: per-thread scratch arenas, shared immutable handles (like Arc<SpatialIndex>)."""
    import threading

    polys = GeoSeries(synth.star_polygons(80, 20))
    idx = SpatialIndex(polys)
    sets = [synth.uniform_points(20_000, seed=100 + k) for k in range(4)]
    expected = [oracle.spatial_join(p, polys.array, "intersects", mode=1)[0] for p in sets]
    results = [None] * 4

    def work(k):
        s = GeoSeries(sets[k])
        for _ in range(3):
            results[k] = join_pairs(s, polys, "intersects", r_index=idx)[0]

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(4):
        assert np.array_equal(results[k], expected[k])


def test_stream_ordered_join_matches_blocking_join(gpk, oracle):
    """gpk_spatial_join_async: several joins queued on one stream without a host wait in between give the same
    counts / pairs / total as the blocking call and the oracle; pairs beyond the capacity are dropped, the total is not."""
    import torch

    from geopolars_amd.spatial_index import join_pairs_enqueue

    polys_h = synth.star_polygons(300, 24)
    pts_h = synth.uniform_points(200_000, seed=7)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts_h, polys_h, "intersects", mode=1)
    polys, pts = GeoSeries(polys_h), GeoSeries(pts_h)
    idx = SpatialIndex(polys)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for cap in (len(pts_h.xy), 1000, 0):
        counts = torch.full((len(pts_h.xy),), -1, dtype=torch.int32, device=dev)
        pairs = torch.full((max(cap, 1), 2), -1, dtype=torch.int32, device=dev)
        total = torch.full((1,), -1, dtype=torch.int64, device=dev)
        join_pairs_enqueue(pts.device(), polys.device(), idx, "intersects", counts, pairs if cap else None, total, stream=stream)
        outs.append((cap, counts, pairs, total))
    torch.cuda.synchronize()
    for cap, counts, pairs, total in outs:
        assert int(total.item()) == len(exp_pairs)
        assert np.array_equal(counts.cpu().numpy().view(np.uint32), exp_counts)
        k = min(cap, len(exp_pairs))
        assert np.array_equal(pairs.cpu().numpy().view(np.uint32)[:k], exp_pairs[:k])
        if cap:
            assert np.all(pairs.cpu().numpy()[k:] == -1)  # nothing written past the capacity / the total


def test_join_dispatch_arms_of_the_reference(gpk, oracle):
    """every arm of the match in spatial_index.rs:89-137: polygon LEFT x point RIGHT (the same `poly.contains(point)`),
    Line / LineString / MultiLineString <-> Point (`line.contains(point)`), and `_ => false` for everything else"""
    polys = synth.star_polygons(300, 16)
    pts = synth.uniform_points(40_000, seed=21)
    # polygon on the left: the transpose of the point-left join, sorted by (polygon, point)
    exp_pairs, exp_counts, _ = oracle.spatial_join(polys, pts, "intersects", mode=1)
    got_pairs, got_counts = join_pairs(GeoSeries(polys), GeoSeries(pts), "intersects")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs) and len(exp_pairs) > 5000
    fwd, _ = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects")
    assert np.array_equal(np.sort(fwd[:, 0].astype(np.int64) * 1000 + fwd[:, 1]), np.sort(got_pairs[:, 1].astype(np.int64) * 1000 + got_pairs[:, 0]))
    shifted, _ = join_pairs(GeoSeries(polys), GeoSeries(pts), "within", left_row_base=50)
    assert np.array_equal(shifted[:, 0], exp_pairs[:, 0] + 50)
    mp = synth.powerlaw_multipolygons(400, seed=3, domain=300.0)
    p2 = synth.uniform_points(20_000, seed=4, domain=300.0)
    exp_pairs, exp_counts, _ = oracle.spatial_join(mp, p2, "contains", mode=1)
    got_pairs, got_counts = join_pairs(GeoSeries(mp), GeoSeries(p2), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)

    # lines and points: points ON the lines (vertices, end points, segment midpoints) and off them
    rng = np.random.default_rng(8)
    lines = GeoArrowArray.from_linestrings(
        [np.cumsum(rng.integers(-4, 5, (int(n), 2)), axis=0).astype(float).tolist() for n in rng.integers(2, 12, 400)]
        + [[(0, 0), (4, 0), (4, 4), (0, 4), (0, 0)], [(7, 7), (7, 7)], []]
    )
    on = np.concatenate([lines.xy[::2], (lines.xy[:-1] + lines.xy[1:])[::3] / 2.0, [[0.0, 0.0], [7.0, 7.0], [2.0, 0.0]]])
    lp = GeoArrowArray.from_points(np.concatenate([on, rng.integers(-30, 30, (3000, 2)).astype(float), [[np.nan, np.nan]]]))
    for left, right in ((lp, lines), (lines, lp)):
        exp_pairs, exp_counts, _ = oracle.spatial_join(left, right, "intersects", mode=0)
        got_pairs, got_counts = join_pairs(GeoSeries(left), GeoSeries(right), "intersects")
        assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
        assert 50 < len(exp_pairs)
    ml = GeoArrowArray(_abi.GEOM_MULTILINESTRING, lines.xy, geom_offsets=np.array([0, 3, 3, 200, len(lines)], np.int32), ring_offsets=lines.geom_offsets)
    exp_pairs, exp_counts, _ = oracle.spatial_join(lp, ml, "contains", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(lp), GeoSeries(ml), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)

    # `_ => false`: an empty join, not an error
    for left, right in ((pts, pts), (lines, lines), (lines, polys), (polys, lines)):
        got_pairs, got_counts = join_pairs(GeoSeries(left), GeoSeries(right), "intersects")
        assert len(got_pairs) == 0 and not got_counts.any() and len(got_counts) == len(left)


def test_one_row_with_tens_of_thousands_of_candidates(gpk, oracle):
    """one polygon over the whole domain against a column of small ones (and the reverse): a candidate slice far beyond
    the in-place sort of the fill pass goes through the segmented radix sort; pairs still come out sorted by (l, r)."""
    rng = np.random.default_rng(8)
    huge = synth.star_polygons(1, 64)
    c = rng.uniform(0, 1000, (40_000, 2))
    rings = np.empty((len(c), 5, 2))
    for k, (sx, sy) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1), (-1, -1))):
        rings[:, k, 0] = c[:, 0] + sx * 0.5
        rings[:, k, 1] = c[:, 1] + sy * 0.5
    small = GeoArrowArray(_abi.GEOM_POLYGON, rings.reshape(-1, 2), geom_offsets=np.arange(len(c) + 1, dtype=np.int32), ring_offsets=np.arange(0, 5 * len(c) + 1, 5, dtype=np.int32))
    for pred in ("intersects", "contains"):
        for l, r in ((huge, small), (small, huge)):
            exp_pairs, exp_counts, _ = oracle.spatial_join(l, r, pred, mode=1)
            got_pairs, got_counts = join_pairs(GeoSeries(l), GeoSeries(r), pred)
            assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
    assert len(exp_pairs) == 0  # a small square never contains the big star


def test_self_intersecting_ring_follows_the_oracles_winding_rule(gpk, oracle):
    """the {5/2} star winds twice around its core: the HIP join answers what the oracle answers (winding number != 0), whichever
    kernel serves it (SURVEY Appendix A.1 [verify]; tests/test_oracle_exact.py pins the oracle's choice)"""
    import math

    star = [(50 + 40 * math.cos(2 * math.pi * (2 * k) / 5 + 0.3), 50 + 40 * math.sin(2 * math.pi * (2 * k) / 5 + 0.3)) for k in range(5)]
    polys = GeoArrowArray.from_polygons([[star]])
    pts = synth.uniform_points(20_000, seed=77, domain=100.0)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "contains", mode=0)
    got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "contains")
    assert np.array_equal(got_counts, exp_counts) and np.array_equal(got_pairs, exp_pairs)
    core = np.hypot(pts.xy[:, 0] - 50, pts.xy[:, 1] - 50) < 10  # well inside the doubly wound pentagon
    assert got_counts[core].all()


def test_stream_ordered_joins_on_two_streams_of_one_thread_do_not_share_scratch(gpk):
    """gpk_spatial_join_async keeps its scratch per (thread, stream): joins enqueued alternately on two streams, nothing
    synchronised in between, answer what the blocking join answers"""
    import torch

    from geopolars_amd.geoarrow import DeviceGeoArray
    from geopolars_amd.spatial_index import join_pairs_device, join_pairs_enqueue

    dev = torch.device("cuda", 0)
    polys_host = synth.star_polygons(500, 32)
    s0 = torch.cuda.current_stream().cuda_stream
    polys = DeviceGeoArray.upload(polys_host, stream=s0)
    index = SpatialIndex.from_device(polys, stream=s0)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    jobs = []
    for k in range(6):
        n = 300_000 + 50_000 * k
        xy = torch.from_numpy(synth.uniform_points(n, seed=200 + k).xy).to(dev)
        jobs.append({"pts": DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=s0), "n": n, "counts": torch.empty(n, dtype=torch.int32, device=dev),
                     "pairs": torch.empty((n, 2), dtype=torch.int32, device=dev), "total": torch.zeros(1, dtype=torch.int64, device=dev)})
    torch.cuda.synchronize()
    for rep in range(3):
        for k, j in enumerate(jobs):
            join_pairs_enqueue(j["pts"], polys, index, "intersects", j["counts"], j["pairs"], j["total"], stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    for j in jobs:
        counts = torch.empty(j["n"], dtype=torch.int32, device=dev)
        pairs = torch.empty((j["n"], 2), dtype=torch.int32, device=dev)
        h = join_pairs_device(j["pts"], polys, index, "intersects", counts, pairs, stream=s0)
        assert int(j["total"].item()) == h
        assert torch.equal(j["counts"], counts) and torch.equal(j["pairs"][:h], pairs[:h])


def test_index_tables_are_recycled_and_released(gpk, oracle):
    """gpk_index_free keeps an index's device blocks for the next build (no hipMalloc / hipFree per table); the joins answer the same
    before and after, with the cache switched through gpk_device_cache_release in between"""
    from geopolars_amd import _abi

    polys, pts = synth.star_polygons(300, 32), synth.uniform_points(50_000, seed=9)
    ps, qs = GeoSeries(pts), GeoSeries(polys)
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    for rep in range(4):
        ix = SpatialIndex(qs)
        pairs, counts = join_pairs(ps, qs, "intersects", r_index=ix)
        assert np.array_equal(counts, ec) and np.array_equal(pairs, ep)
        ix.free()
        if rep == 1:
            _abi.check(_abi.lib().gpk_device_cache_release())
    # two live indexes never share a block
    a, b = SpatialIndex(qs), SpatialIndex(qs)
    pa, ca = join_pairs(ps, qs, "intersects", r_index=a)
    pb, cb = join_pairs(ps, qs, "intersects", r_index=b)
    assert np.array_equal(pa, ep) and np.array_equal(pb, ep) and np.array_equal(ca, ec) and np.array_equal(cb, ec)


@pytest.mark.parametrize("with_hole", [False, True])
def test_parts_with_thousands_of_vertices_inside_one_raster_cell(gpk, oracle, with_hole):
    """The index build's level-2 records for a part whose whole ring — thousands of edges — lies inside one or two raster cells (the
    tail of a power-law column): its cells' edge lists overflow many times over (drained against their own box and refilled,
    csrc/gpk_pipindex.hip: sub_build_kernel), its slab rows are refined, and with a hole its rings' spans are laid end to end.  Points
    are thrown densely at those parts and over the whole column; every pair against the oracle."""
    rng = np.random.default_rng(77)
    n_small = 3000
    small = synth.star_polygons(n_small, 12, seed=5)
    rings, parts = [], []
    centres = rng.uniform(20.0, 80.0, (6, 2))
    for k, (cx, cy) in enumerate(centres):
        nv = (900, 2500, 6000, 15000, 333, 40000)[k]
        ang = np.linspace(0.0, 2.0 * np.pi, nv, endpoint=False)
        rad = 0.11 * (0.8 + 0.2 * np.sin(7.0 * ang + k))  # about a raster cell across for a column of this size
        ext = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1)
        ext = np.concatenate([ext, ext[:1]])
        part = [ext]
        if with_hole:
            hole = np.stack([cx + 0.4 * rad * np.cos(-ang), cy + 0.4 * rad * np.sin(-ang)], axis=1)
            part.append(np.concatenate([hole, hole[:1]]))
        parts.append(part)
    xy = np.concatenate([small.xy] + [r for p in parts for r in p])
    ring_lens = list(np.diff(small.ring_offsets)) + [len(r) for p in parts for r in p]
    ring_off = np.concatenate([[0], np.cumsum(ring_lens)]).astype(np.int32)
    geom_rings = list(np.diff(small.geom_offsets)) + [len(p) for p in parts]
    geom_off = np.concatenate([[0], np.cumsum(geom_rings)]).astype(np.int32)
    right = GeoArrowArray(_abi.GEOM_POLYGON, xy, geom_offsets=geom_off, ring_offsets=ring_off)
    near = np.concatenate([c + rng.uniform(-0.2, 0.2, (20_000, 2)) for c in centres])
    on_vertices = np.concatenate([p[0][:: max(1, len(p[0]) // 500)] for p in parts])  # boundary points: not contained
    left = GeoArrowArray.from_points(np.concatenate([synth.uniform_points(200_000, seed=3).xy, near, on_vertices]))
    ep, ec, _ = oracle.spatial_join(left, right, "intersects", mode=0)
    rs = GeoSeries(right)
    gp, gc = join_pairs(GeoSeries(left), rs, "intersects", r_index=SpatialIndex(rs))
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
    assert ec[200_000 : 200_000 + len(near)].sum() > 10_000  # the dense points do land inside the many-vertex parts


def test_the_index_a_join_builds_for_itself_stays_on_the_right_handle(gpk, oracle):
    """the reference's default call shape — SpatialJoinArgs::default() has r_index: None (spatial_index.rs:24-35,60-71) — repeated
    against one right side: the first call builds the index, later calls find it on the (immutable) handle; a different left side
    and a different arm (polygon x polygon: boxes + directory only) are served too; freeing the handle frees what it holds"""
    polys = synth.star_polygons(500, 32)
    right = GeoSeries(polys)
    for seed in (1, 2, 3):
        pts = synth.uniform_points(40_000 + seed, seed=seed)
        ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
        gp, gc = join_pairs(GeoSeries(pts), right, "intersects", r_index=None)
        assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
    other = synth.clustered_polygons(2000, seed=5)
    ep, ec, _ = oracle.spatial_join(other, polys, "intersects", mode=0)
    for _ in range(2):
        gp, gc = join_pairs(GeoSeries(other), right, "intersects", r_index=None)
        assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
    right.device().free()
