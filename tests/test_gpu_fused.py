"""GPU parity of the one-launch point join (gpk_pipflow.hip: pip_tile_flow_kernel) through the C ABI vs the CPU oracle, bit-exact on counts,
pairs and totals (`Contains<Point>`, spatial_index.rs:91-96; sorted pairs = the two index vectors of spatial_index.rs:145-159).  Tiles
decided, hits ranked and the sorted (l, r) pair list written by the same persistent work-groups:
  * a row with a polygon to be in is a hit at once — a 4-byte entry in its tile's slots of a global pool, count 1 — and `test` points wait
    on their wave's LDS list for a dense pass of the exact step; a failed test kills the entry, the count, and one of the tile's total;
  * tiles of 64 * P rows, P = 8 for long columns, 4 / 2 / 1 for short ones (GPK_FLOW_P forces one, read once per process: the forced
    sizes run in their own interpreter);
  * GPK_TILE_KERNEL=chain: the chain tile kernel + writer in the one-launch join's place (the same answers, its own interpreter).
The hit forms of rounds 4 and 5 (per-wave LDS lists, staging slots, chunks, the work-group pool) were retired in round 6.

What only this kernel has, and what is aimed at here: entries that wait in the pool until the work-groups before have published their
totals; rows the tables cannot settle (list cells, half cells without a chain, uncertifiable orientations: walked by the whole wave
from the list); rows in SEVERAL geometries (one entry that carries the count, expanded at emission); tiles that list more rows than a
wave's list holds (entries left pending, settled at the end of the tile); more than 65,535 pairs before an ordinary tile of a
work-group; left_row_base; a pair buffer smaller than the total; count-only calls; launches from two streams (they share the epoch
words); columns shorter than one tile per wave, and longer than a round of emission."""
import ctypes as C

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray, GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import MEM_HOST, PREDICATES, SpatialIndex, join_pairs, join_pairs_enqueue

pytestmark = pytest.mark.gpu


def _stacked(n_stars: int, stack: int, at=(500.0, 960.0), side=4.0, verts: int = 24) -> GeoArrowArray:
    """disjoint stars (a lean right side with chains) + `stack` identical squares in a corner no star reaches: the raster cells there
    are LIST cells, a point inside the squares lies in `stack` geometries"""
    stars = synth.star_polygons(n_stars, verts)
    x, y = at
    sq = [[[(x, y), (x + side, y), (x + side, y + side), (x, y + side)]] for _ in range(stack)]
    polys = []
    ro, xy = stars.ring_offsets, stars.xy
    for r in range(len(ro) - 1):
        polys.append([[tuple(p) for p in xy[ro[r] : ro[r + 1] - 1]]])
    return GeoArrowArray.from_polygons(polys + sq)


def _check(oracle, pts, polys, pred="intersects", base=0, want=None):
    right = GeoSeries(polys)
    index = SpatialIndex(right)
    d = index.describe()
    if want:
        for k, v in want.items():
            assert d[k] == v, (k, d)
    ep, ec, _ = oracle.spatial_join(pts, polys, pred, mode=0)
    gp, gc = join_pairs(GeoSeries(pts), right, pred, r_index=index, left_row_base=base)
    ep = ep.copy()
    ep[:, 0] += base
    assert np.array_equal(gc, ec)
    assert np.array_equal(gp, ep)
    return d, ec


def test_rows_in_many_geometries_outgrow_a_waves_slots(gpk, oracle):
    polys = _stacked(900, 37)
    rng = np.random.default_rng(3)
    inside = np.column_stack([rng.uniform(499.0, 505.0, 4000), rng.uniform(959.0, 965.0, 4000)])
    on_edge = np.array([[500.0, 961.0], [502.0, 960.0], [504.0, 964.0], [500.0, 960.0]])  # boundary: not contained
    pts = np.concatenate([synth.uniform_points(30_000, seed=4).xy, inside, on_edge])
    rng.shuffle(pts)
    d, counts = _check(oracle, GeoArrowArray.from_points(pts), polys, want={"lean": True, "chains": True, "route": True})
    assert counts.max() == 37 and counts.sum() > 37 * 1000


def test_multi_hit_rows_between_ordinary_hits_keep_the_pair_order(gpk, oracle):
    """a few multi-hit rows inside tiles full of ordinary hits: their hits open a gap in the middle of what the tile parked"""
    polys = _stacked(1000, 5, at=(990.0, 985.0), side=6.0, verts=64)
    rng = np.random.default_rng(5)
    pts = synth.uniform_points(200_000, seed=6).xy
    pts[rng.integers(0, len(pts), 300)] = np.column_stack([rng.uniform(989.0, 997.0, 300), rng.uniform(984.0, 992.0, 300)])
    for base in (0, 7_000_000):
        d, counts = _check(oracle, GeoArrowArray.from_points(pts), polys, base=base)
        assert counts.max() == 5


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 131_071, 1_048_577 + 300])
def test_column_lengths_around_tile_and_wave_boundaries(gpk, oracle, n):
    polys = synth.star_polygons(1000, 64)
    _check(oracle, synth.uniform_points(n, seed=n % 97 + 1), polys, "contains")


def test_left_validity_and_empty_points_take_the_guarded_tiles(gpk, oracle):
    rng = np.random.default_rng(8)
    polys = synth.star_polygons(1000, 64)
    xy = rng.uniform(0, 1000, (150_001, 2))
    xy[rng.integers(0, len(xy), 700)] = np.nan
    tv = np.packbits(rng.uniform(size=len(xy)) > 0.15, bitorder="little")
    _check(oracle, GeoArrowArray.from_points(xy, validity=tv), polys, base=12345)


def test_pair_buffer_smaller_than_the_total_and_count_only(gpk, oracle):
    lib = _abi.lib()
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(120_000, seed=9)
    left, right = GeoSeries(pts), GeoSeries(polys)
    index = SpatialIndex(right)
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
    n_pairs = C.c_int64(0)
    counts = np.zeros(len(pts), dtype=np.uint32)
    # count only
    _abi.check(lib.gpk_spatial_join(left.device().handle, right.device().handle, index.handle, PREDICATES["intersects"], 0, counts.ctypes.data, None, 0, C.byref(n_pairs), MEM_HOST, None))
    assert int(n_pairs.value) == len(ep) and np.array_equal(counts, ec)
    # total only
    n_pairs.value = 0
    _abi.check(lib.gpk_spatial_join(left.device().handle, right.device().handle, index.handle, PREDICATES["intersects"], 0, None, None, 0, C.byref(n_pairs), MEM_HOST, None))
    assert int(n_pairs.value) == len(ep)
    # a buffer that holds a third of the pairs: the total is reported, the prefix is right, nothing past the buffer is touched
    cap = len(ep) // 3
    buf = np.full((cap + 64, 2), 0xABCDEF01, dtype=np.uint32)
    rc = lib.gpk_spatial_join(left.device().handle, right.device().handle, index.handle, PREDICATES["intersects"], 0, counts.ctypes.data, buf.ctypes.data, cap, C.byref(n_pairs), MEM_HOST, None)
    assert rc == _abi.GPK_ERR_CAPACITY and int(n_pairs.value) == len(ep)
    assert np.all(buf[cap:] == 0xABCDEF01)


def test_launches_from_two_streams_and_back_to_back(gpk, oracle):
    import torch

    dev = torch.device("cuda", 0)
    polys = synth.star_polygons(1000, 64)
    dpolys = DeviceGeoArray.upload(polys)
    index = SpatialIndex.from_device(dpolys)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    jobs = []
    for i in range(6):
        n = 40_000 + 12_345 * i
        pts = synth.uniform_points(n, seed=20 + i)
        st = streams[i % 2]
        with torch.cuda.stream(st):
            xy = torch.from_numpy(pts.xy).to(dev, non_blocking=False)
            dp = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=st.cuda_stream)
            c = torch.empty(n, dtype=torch.int32, device=dev)
            pr = torch.empty((n, 2), dtype=torch.int32, device=dev)
            t = torch.zeros(1, dtype=torch.int64, device=dev)
            join_pairs_enqueue(dp, dpolys, index, "intersects", c, pr, t, left_row_base=i, stream=st.cuda_stream)
            join_pairs_enqueue(dp, dpolys, index, "intersects", c, pr, t, left_row_base=i, stream=st.cuda_stream)  # same buffers again
        jobs.append((pts, i, dp, xy, c, pr, t))
    torch.cuda.synchronize()
    for pts, base, dp, xy, c, pr, t in jobs:
        ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
        ep = ep.copy()
        ep[:, 0] += base
        assert int(t.item()) == len(ep)
        assert np.array_equal(c.cpu().numpy().view(np.uint32), ec)
        assert np.array_equal(pr[: len(ep)].cpu().numpy().view(np.uint32), ep)


def test_the_chain_kernel_and_writer_still_answer_the_same(gpk, oracle):
    """GPK_TILE_KERNEL=chain (read once per process): chain tile kernel + writer on the same inputs in its own interpreter"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import numpy as np\n"
        "from geopolars_amd import synth\n"
        "from geopolars_amd.geoseries import GeoSeries\n"
        "from geopolars_amd.spatial_index import SpatialIndex, join_pairs\n"
        "from oracle import pyoracle\n"
        "pyoracle.build()\n"
        "polys = synth.star_polygons(1000, 64); pts = synth.uniform_points(300_011, seed=31)\n"
        "right = GeoSeries(polys); index = SpatialIndex(right)\n"
        "ep, ec, _ = pyoracle.spatial_join(pts, polys, 'intersects', mode=0)\n"
        "gp, gc = join_pairs(GeoSeries(pts), right, 'intersects', r_index=index)\n"
        "assert np.array_equal(gc, ec) and np.array_equal(gp, ep)\n"
        "print('ok', int(ec.sum()))\n"
    )
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, GPK_TILE_KERNEL="chain"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_a_column_of_eleven_million_rows(gpk, oracle):
    """84 tiles per work-group: the round-5 pool form ended at 80 (10.49 M rows on 256 CUs)"""
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(11_000_003, seed=77)
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    gp, gc = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects", r_index=SpatialIndex(GeoSeries(polys)))
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)


_FORM_PROG = (
    "import numpy as np, sys\n"
    "sys.path.insert(0, 'tests')\n"
    "from geopolars_amd import synth\n"
    "from geopolars_amd.geoarrow import GeoArrowArray\n"
    "from geopolars_amd.geoseries import GeoSeries\n"
    "from geopolars_amd.spatial_index import SpatialIndex, join_pairs\n"
    "from oracle import pyoracle\n"
    "import test_gpu_fused as T\n"
    "pyoracle.build()\n"
    "rng = np.random.default_rng(3)\n"
    "cases = []\n"
    "polys = T._stacked(900, 37)\n"
    "pts = np.concatenate([synth.uniform_points(60_001, seed=4).xy, np.column_stack([rng.uniform(499.0, 505.0, 3000), rng.uniform(959.0, 965.0, 3000)])])\n"
    "rng.shuffle(pts)\n"
    "cases.append((GeoArrowArray.from_points(pts), polys))\n"
    "stars = synth.star_polygons(1000, 64)\n"
    "for n in (0, 1, 511, 513, 8191, 8193, 700_001):\n"
    "    cases.append((synth.uniform_points(n, seed=n % 89 + 2), stars))\n"
    "xy = rng.uniform(0, 1000, (150_001, 2)); xy[rng.integers(0, len(xy), 700)] = np.nan\n"
    "cases.append((GeoArrowArray.from_points(xy, validity=np.packbits(rng.uniform(size=len(xy)) > 0.15, bitorder='little')), stars))\n"
    "for pts, polys in cases:\n"
    "    right = GeoSeries(polys); index = SpatialIndex(right)\n"
    "    ep, ec, _ = pyoracle.spatial_join(pts, polys, 'intersects', mode=0)\n"
    "    for base in (0, 123456):\n"
    "        gp, gc = join_pairs(GeoSeries(pts), right, 'intersects', r_index=index, left_row_base=base)\n"
    "        e = ep.copy(); e[:, 0] += base\n"
    "        assert np.array_equal(gc, ec) and np.array_equal(gp, e), (len(pts), base)\n"
    "print('ok', len(cases))\n"
)


@pytest.mark.parametrize("env", [{"GPK_FLOW_P": "8"}, {"GPK_FLOW_P": "4"}, {"GPK_FLOW_P": "2"}, {"GPK_FLOW_P": "1"}, {"GPK_TILE_KERNEL": "chain"}])
def test_every_tile_size_of_the_join_on_the_same_inputs(gpk, oracle, env):
    """GPK_FLOW_P=8 | 4 | 2 | 1 (read once per process) forces one tile size for every eligible join — by default short columns take the
    small tiles —, GPK_TILE_KERNEL=chain the chain kernel + writer: rows in several geometries, left_row_base, ragged tails, null and
    empty rows, columns around the tile boundaries — in its own interpreter"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FORM_PROG], capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_tiles_that_list_more_rows_than_a_waves_list_holds(gpk, oracle):
    """points packed along the polygons' edges: a 512-row tile lists hundreds of `test` points, its wave's list holds 128 — the surplus
    stays pending in the pool and is settled by the generic walk at the end of the tile (GPK_FLOW_P=8: in its own interpreter)"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import numpy as np, ctypes as C\n"
        "from geopolars_amd import synth, _abi\n"
        "from geopolars_amd.geoarrow import GeoArrowArray\n"
        "from geopolars_amd.geoseries import GeoSeries\n"
        "from geopolars_amd.spatial_index import SpatialIndex, join_pairs\n"
        "from oracle import pyoracle\n"
        "pyoracle.build()\n"
        "polys = synth.star_polygons(400, 24)\n"
        "rng = np.random.default_rng(12)\n"
        "v = polys.xy; ro = polys.ring_offsets\n"
        "a, b = v[:-1], v[1:]; same = np.ones(len(a), dtype=bool); same[ro[1:-1] - 1] = False\n"
        "t = rng.uniform(0, 1, (same.sum(), 12, 1))\n"
        "on = (a[same][:, None, :] * (1 - t) + b[same][:, None, :] * t).reshape(-1, 2) + rng.normal(0, 0.02, (same.sum() * 12, 2))\n"
        "xy = np.concatenate([on, synth.uniform_points(40_000, seed=5).xy]); rng.shuffle(xy[: len(on) // 2])\n"
        "pts = GeoArrowArray.from_points(xy)\n"
        "right = GeoSeries(polys); index = SpatialIndex(right)\n"
        "assert index.describe()['route']\n"
        "lib = _abi.lib(); st = (C.c_int64 * 4)(); lib.gpk_join_stats_enable(1); lib.gpk_join_stats(st, 1)\n"
        "ep, ec, _ = pyoracle.spatial_join(pts, polys, 'intersects', mode=0)\n"
        "for base in (0, 999):\n"
        "    gp, gc = join_pairs(GeoSeries(pts), right, 'intersects', r_index=index, left_row_base=base)\n"
        "    e = ep.copy(); e[:, 0] += base\n"
        "    assert np.array_equal(gc, ec) and np.array_equal(gp, e)\n"
        "lib.gpk_join_stats(st, 1)\n"
        "assert st[2] > 1000, list(st)  # rows settled by the walk: the pending ones\n"
        "print('ok', int(ec.sum()), list(st))\n"
    )
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, GPK_FLOW_P="8"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_more_than_65535_pairs_before_an_ordinary_tile_of_a_work_group(gpk, oracle):
    """spatially sorted points: a long run inside a zone of 8 stacked squares (rows in 8 geometries each: far more than 65,535 pairs within
    one work-group's range), ordinary tiles after it.  (The round-5 pool form packed a row's pair offset within its work-group into 16
    bits: ADVICE r5.)"""
    polys = _stacked(900, 8, at=(500.0, 960.0), side=6.0, verts=24)
    rng = np.random.default_rng(21)
    zone = np.column_stack([rng.uniform(500.2, 505.8, 30_000), rng.uniform(960.2, 965.8, 30_000)])
    rest = synth.uniform_points(1_500_000, seed=22).xy
    rest = rest[np.lexsort((rest[:, 0], np.floor(rest[:, 1] / 8.0)))]  # rows of 8 units, x ascending within: neighbours stay neighbours
    cut = 700_000
    xy = np.concatenate([rest[:cut], zone, rest[cut:]])
    ep, ec, _ = oracle.spatial_join(GeoArrowArray.from_points(xy), polys, "intersects", mode=1)
    right = GeoSeries(polys)
    gp, gc = join_pairs(GeoSeries(GeoArrowArray.from_points(xy)), right, "intersects", r_index=SpatialIndex(right))
    assert int(ec[cut : cut + len(zone)].sum()) == 8 * len(zone) > 3 * 65536
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
