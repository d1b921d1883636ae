"""GPU parity of the one-launch point join (gpk_join.hip) through the C ABI vs the CPU oracle, bit-exact on counts, pairs and totals
(`Contains<Point>`, spatial_index.rs:91-96; sorted pairs = the two index vectors of spatial_index.rs:145-159).  Tiles decided, hits
ranked and the sorted (l, r) pair list written by the same persistent work-groups, in one of four forms that must all answer alike:
  pool     (round 5, the default up to 10.49 M rows on 256 CUs) a work-group's waves draw tiles from an LDS counter, hits in one pool of
           16-bit geometry ids per work-group, rare rows settled before the tile's hits are ranked — pip_tile_pool_kernel;
  chunked  (round 5, longer columns) chunks of 16 tiles from an agent-scope counter, a chunk's pairs written while the next is decided,
           two tiles of hits per wave in LDS whatever the column's length — pip_tile_chunked_kernel;
  wave     (round 4, GPK_FUSED_FORM=wave) a contiguous run of tiles per wave, its hits in the wave's own LDS list — pip_tile_fused_kernel<true>;
  staging  (GPK_FUSED_LDS=0, or more than 65,535 geometries) hits parked in the pair slots of the wave's own rows.
GPK_FUSED_FORM / GPK_FUSED_LDS are read once per process: the forms other than the default run in their own interpreter.

What only these kernels have, and what is aimed at here: hits that wait in LDS until the work-groups before have published their totals;
rare rows (list cells, half cells without a chain, uncertifiable orientations); rows in SEVERAL geometries (the tile is decided again,
storing at final offsets); left_row_base; a pair buffer smaller than the total; count-only calls; launches from two streams (they
share the epoch words); columns shorter than one tile per wave, and longer than any LDS list."""
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


def test_the_round_three_pair_of_kernels_still_answers_the_same(gpk, oracle):
    """GPK_TILE_KERNEL=route (read once per process): routed tile kernel + writer on the same inputs in its own interpreter"""
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
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, GPK_TILE_KERNEL="route"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_a_column_too_long_for_a_work_groups_pool_takes_the_chunked_form(gpk, oracle):
    """more than 80 tiles per work-group (> 10.49 M points on 256 CUs): chunks of 16 tiles from a counter, 1343 chunks = 6 generations"""
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(11_000_003, seed=77)
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    gp, gc = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects", r_index=SpatialIndex(GeoSeries(polys)))
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)


def test_the_staging_form_on_the_same_inputs(gpk, oracle):
    """GPK_FUSED_LDS=0 (read once per process): rows in several geometries, left_row_base, ragged tails — in its own interpreter"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, 'tests')\n"
        "from geopolars_amd import synth\n"
        "from geopolars_amd.geoarrow import GeoArrowArray\n"
        "from geopolars_amd.geoseries import GeoSeries\n"
        "from geopolars_amd.spatial_index import SpatialIndex, join_pairs\n"
        "from oracle import pyoracle\n"
        "import test_gpu_fused as T\n"
        "pyoracle.build()\n"
        "polys = T._stacked(900, 37)\n"
        "rng = np.random.default_rng(3)\n"
        "pts = np.concatenate([synth.uniform_points(60_001, seed=4).xy, np.column_stack([rng.uniform(499.0, 505.0, 3000), rng.uniform(959.0, 965.0, 3000)])])\n"
        "rng.shuffle(pts)\n"
        "pts = GeoArrowArray.from_points(pts)\n"
        "right = GeoSeries(polys); index = SpatialIndex(right)\n"
        "ep, ec, _ = pyoracle.spatial_join(pts, polys, 'intersects', mode=0)\n"
        "for base in (0, 123456):\n"
        "    gp, gc = join_pairs(GeoSeries(pts), right, 'intersects', r_index=index, left_row_base=base)\n"
        "    e = ep.copy(); e[:, 0] += base\n"
        "    assert np.array_equal(gc, ec) and np.array_equal(gp, e)\n"
        "print('ok', int(ec.sum()), int(ec.max()))\n"
    )
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, GPK_FUSED_LDS="0"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


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


@pytest.mark.parametrize("form", ["wave", "chunked", "pool"])
def test_every_form_of_the_fused_join_on_the_same_inputs(gpk, oracle, form):
    """GPK_FUSED_FORM=wave | chunked | pool (read once per process) forces one form for every eligible join: rows in several geometries,
    left_row_base, ragged tails, null and empty rows, columns around the tile / chunk boundaries — in its own interpreter"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FORM_PROG], capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, GPK_FUSED_FORM=form))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]
