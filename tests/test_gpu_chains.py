"""GPU parity of the chain kernels (gpk_join.hip: pip_tile_chain_kernel + writer; gpk_pipflow.hip: pip_tile_flow_kernel, the one-launch join — rare rows settled inside them)
through the C ABI vs the CPU oracle, bit-exact on counts and sorted (l, r) pairs (`Contains<Point>`, spatial_index.rs:91-96).

Right sides here are DISJOINT polygons — what makes an index "lean" and gives it local chains — shaped to reach every arm:
chains of one / two / several edges (the fifth vertex onwards lives in a side table), chains over the ring's closing vertex, sub-cells with more than
CHAIN_MAX edges or several boundary runs (no chain entry: deferred), parts with holes (deferred), points exactly on edges and
vertices (orientation not certifiable by the floating-point filter: deferred), list cells of a lean index (deferred), null
rows on either side, tiles that end mid-wave, rasters with and without the LDS routing image (R <= 512 / R = 1024)."""
import ctypes as C

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs

pytestmark = pytest.mark.gpu


def _deferred_rows(fn):
    """runs fn() with the join statistics on; returns (result, pairs sent to the exact step, left rows deferred)"""
    lib = _abi.lib()
    st = (C.c_int64 * 4)()
    lib.gpk_join_stats_enable(1)
    lib.gpk_join_stats(st, 1)
    try:
        out = fn()
        lib.gpk_join_stats(st, 1)
    finally:
        lib.gpk_join_stats_enable(0)
    return out, int(st[0]), int(st[2])


def check(oracle, pts: GeoArrowArray, polys: GeoArrowArray, pred="intersects", want=None):
    """join through a PREBUILT index (so that the test can look at what the index carries) against the oracle"""
    right = GeoSeries(polys)
    index = SpatialIndex(right)
    d = index.describe()
    if want is not None:
        for k, v in want.items():
            assert d[k] == v, (k, d)
    exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, pred, mode=0)
    (got, exact, deferred) = _deferred_rows(lambda: join_pairs(GeoSeries(pts), right, pred, r_index=index))
    got_pairs, got_counts = got
    assert np.array_equal(got_counts, exp_counts)
    assert np.array_equal(got_pairs, exp_pairs)
    return d, exact, deferred, exp_counts


def around(xy: np.ndarray, radius: float, per: int, seed: int) -> np.ndarray:
    """`per` points within `radius` of every row of xy, the rows themselves included"""
    rng = np.random.default_rng(seed)
    jit = rng.uniform(-radius, radius, (len(xy), per, 2))
    return np.concatenate([xy, (xy[:, None, :] + jit).reshape(-1, 2)])


def test_headline_right_side_has_chains_and_a_routing_image(gpk, oracle):
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(300_007)  # (the last tile ends mid-wave)
    d, exact, deferred, counts = check(oracle, pts, polys, want={"R": 512, "lean": True, "chains": True, "route": True})
    assert exact > 10_000          # the `test` sub-cells are exercised ...
    assert deferred < exact // 20  # ... and nearly all of them have a chain entry
    assert counts.sum() > 50_000


def test_larger_raster_runs_the_chain_kernel_without_the_lds_image(gpk, oracle):
    polys = synth.star_polygons(4100, 64)  # 266k coordinates: a raster beyond the LDS image (R > 512)
    pts = synth.uniform_points(200_001)
    d, exact, deferred, _ = check(oracle, pts, polys, want={"lean": True, "chains": True, "route": False})
    assert d["R"] > 512 and exact > 5_000


@pytest.mark.parametrize("n_polys,n_verts", [(1, 3), (2, 4), (7, 5), (40, 16), (300, 17), (1000, 8)])
def test_small_rasters(gpk, oracle, n_polys, n_verts):
    polys = synth.star_polygons(n_polys, n_verts)
    for n in (0, 1, 63, 64, 65, 255, 256, 257, 5000):
        check(oracle, synth.uniform_points(n, seed=n + 11), polys)


def test_points_on_vertices_and_edges_are_deferred_and_exact(gpk, oracle):
    """dyadic coordinates: edge midpoints are exactly representable and exactly ON the edge, so the floating-point filter
    cannot certify their orientation; vertices and points level with vertices ride the half-open rule"""
    rng = np.random.default_rng(5)
    polys_l = []
    for gx in range(12):
        for gy in range(12):
            cx, cy = 80.0 * gx + 40.0, 80.0 * gy + 40.0
            ang = np.sort(rng.uniform(0, 2 * np.pi, 11))
            rad = rng.integers(64, 256, 11) / 8.0  # multiples of 1/8 up to 32
            ring = [(cx + np.round(r * np.cos(a) * 8) / 8, cy + np.round(r * np.sin(a) * 8) / 8) for r, a in zip(rad, ang)]
            polys_l.append([ring])
    polys = GeoArrowArray.from_polygons(polys_l)
    ro = polys.ring_offsets
    v = polys.xy
    a, b = v[:-1], v[1:]
    same_ring = np.ones(len(a), dtype=bool)
    same_ring[ro[1:-1] - 1] = False
    mid = ((a + b) / 2.0)[same_ring]
    quarter = ((3 * a + b) / 4.0)[same_ring]
    level = np.concatenate([v + [0.5, 0.0], v - [0.5, 0.0], v + [0.0, 0.125]])
    pts = GeoArrowArray.from_points(np.concatenate([v, mid, quarter, level, around(v, 0.01, 3, 1)]))
    for pred in ("intersects", "contains"):
        d, exact, deferred, counts = check(oracle, pts, polys, pred, want={"lean": True, "chains": True})
        assert deferred >= len(mid) // 2  # on-edge points cannot be certified: they go through the generic walk


def test_chain_over_the_closing_vertex(gpk, oracle):
    """points packed around every ring's FIRST vertex (= its closing vertex): a chain there would wrap"""
    polys = synth.star_polygons(400, 32)
    first = polys.xy[polys.ring_offsets[:-1]]
    second = polys.xy[polys.ring_offsets[:-1] + 1]
    last = polys.xy[polys.ring_offsets[1:] - 2]
    pts = np.concatenate([around(first, 0.3, 40, 2), (first + second) / 2, (first + last) / 2, around((first + last) / 2, 0.05, 6, 3)])
    d, exact, deferred, _ = check(oracle, GeoArrowArray.from_points(pts), polys, want={"lean": True, "chains": True})
    assert exact > 1000


def _toothed_square(x, y, side, teeth, depth, width):
    """a square whose top side carries `teeth` tiny teeth (each `width` wide, `depth` deep) next to its top-left corner"""
    ring = [(x, y), (x + side, y), (x + side, y + side)]
    tx = x + width * (2 * teeth + 1)
    for t in range(teeth):  # walking the top side from right to left
        ring += [(tx - 2 * t * width, y + side), (tx - (2 * t + 0.5) * width, y + side - depth), (tx - (2 * t + 1) * width, y + side)]
    ring += [(x, y + side)]
    return [ring]


def test_subcells_with_more_edges_than_a_chain_holds(gpk, oracle):
    """thirty edges inside one sub-cell (the raster's sub-cells are ~0.25 wide here): no chain entry -> deferred"""
    polys_l = [_toothed_square(40.0 * i + 3.0, 40.0 * j + 3.0, 30.0, 10, 0.004, 0.003) for i in range(25) for j in range(25)]
    polys = GeoArrowArray.from_polygons(polys_l)
    corners = np.array([[40.0 * i + 3.0, 40.0 * j + 33.0] for i in range(25) for j in range(25)])
    pts = np.concatenate([around(corners + [0.03, -0.002], 0.04, 60, 4), synth.uniform_points(20_000, seed=9).xy])
    d, exact, deferred, counts = check(oracle, GeoArrowArray.from_points(pts), polys, want={"lean": True, "chains": True})
    assert deferred > 500


def test_two_boundary_runs_in_one_subcell(gpk, oracle):
    """a slit 0.002 wide cut into every 40-gon: both of its sides cross the same sub-cells, far apart along the ring — the arc
    that covers both is longer than a chain holds: no chain entry -> deferred"""
    polys_l = []
    for i in range(20):
        for j in range(20):
            cx, cy = 50.0 * i + 25.0, 50.0 * j + 25.0
            ang = np.linspace(-np.pi / 2, 3 * np.pi / 2, 41)[1:-1]  # 39 vertices on a circle, the gap at the bottom holds the slit
            ring = [(cx - 0.001, cy - 20.0), (cx - 0.001, cy + 5.0), (cx + 0.001, cy + 5.0), (cx + 0.001, cy - 20.0)]
            ring += [(cx + 20.0 * np.cos(a), cy + 20.0 * np.sin(a)) for a in ang]
            polys_l.append([ring])
    polys = GeoArrowArray.from_polygons(polys_l)
    slit = np.array([[50.0 * i + 25.0, 50.0 * j + 5.0 + t] for i in range(20) for j in range(20) for t in (3.0, 7.3, 12.9, 24.0, 29.9995, 30.0005)])
    pts = np.concatenate([around(slit, 0.004, 30, 6), synth.uniform_points(20_000, seed=10).xy])
    d, exact, deferred, _ = check(oracle, GeoArrowArray.from_points(pts), polys, want={"lean": True, "chains": True})
    assert exact + deferred > 500  # (decided by half-cell chains, or by the walk: where the arc is too long, and where a tile lists more rows than its wave's list holds)


def test_parts_with_holes(gpk, oracle):
    polys_l = []
    for i in range(30):
        for j in range(30):
            x, y = 33.0 * i + 2.0, 33.0 * j + 2.0
            polys_l.append([[(x, y), (x + 28, y + 1), (x + 29, y + 27), (x + 1, y + 28)], [(x + 8, y + 8), (x + 9, y + 20), (x + 21, y + 19), (x + 20, y + 9)]])
    polys = GeoArrowArray.from_polygons(polys_l)
    pts = synth.uniform_points(120_000, seed=12)
    d, exact, deferred, counts = check(oracle, pts, polys, want={"chains": True})
    assert deferred > 100 and counts.sum() > 10_000


def test_disjoint_multipolygons_map_parts_to_geometries(gpk, oracle):
    mps = []
    for i in range(20):
        for j in range(20):
            x, y = 50.0 * i, 50.0 * j
            mps.append([[[(x + 2, y + 2), (x + 20, y + 3), (x + 19, y + 21), (x + 3, y + 20)]], [[(x + 26, y + 25), (x + 46, y + 27), (x + 44, y + 47), (x + 27, y + 45)]]])
    polys = GeoArrowArray.from_multipolygons(mps)
    d, exact, deferred, counts = check(oracle, synth.uniform_points(100_000, seed=13), polys, want={"lean": True, "chains": True})
    assert counts.sum() > 10_000


def test_null_rows_on_both_sides(gpk, oracle):
    rng = np.random.default_rng(14)
    polys = synth.star_polygons(300, 24)
    pv = np.packbits(rng.uniform(size=len(polys)) > 0.2, bitorder="little")
    polys = GeoArrowArray(polys.geom_type, polys.xy, geom_offsets=polys.geom_offsets, ring_offsets=polys.ring_offsets, validity=pv)
    xy = rng.uniform(0, 1000, (60_000, 2))
    xy[rng.integers(0, len(xy), 500)] = np.nan  # empty points
    tv = np.packbits(rng.uniform(size=len(xy)) > 0.1, bitorder="little")
    d, exact, deferred, counts = check(oracle, GeoArrowArray.from_points(xy, validity=tv), polys, want={"chains": True})
    assert counts.sum() > 1000


def test_list_cells_of_a_lean_index_are_deferred(gpk, oracle):
    """neighbouring polygons whose tips come within a raster cell of each other: a few cells hold two parts"""
    polys_l = []
    for i in range(24):
        for j in range(24):
            x, y = 40.0 * i, 40.0 * j
            polys_l.append([[(x + 0.05, y + 20), (x + 20, y + 0.05), (x + 39.95, y + 20), (x + 20, y + 39.95)]])
    polys = GeoArrowArray.from_polygons(polys_l)
    tips = np.array([[40.0 * i, 40.0 * j + 20.0] for i in range(1, 24) for j in range(24)])
    pts = np.concatenate([around(tips, 0.5, 50, 15), synth.uniform_points(30_000, seed=16).xy])
    d, exact, deferred, _ = check(oracle, GeoArrowArray.from_points(pts), polys)
    if d["chains"]:
        assert deferred > 0


def test_million_rows_against_the_oracle_counts(gpk, oracle):
    polys = synth.star_polygons(1000, 64)
    pts = synth.uniform_points(1_000_003, seed=17)
    check(oracle, pts, polys, "contains", want={"R": 512, "chains": True, "route": True})


@pytest.mark.parametrize("kernel", ["chain", "flow"])
def test_every_tile_kernel_on_the_same_index(gpk, oracle, kernel):
    """GPK_TILE_KERNEL is read once per process (chain: tile kernel + writer instead of the one-launch join): each kernel gets its own interpreter, same inputs, same oracle answers"""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import numpy as np\n"
        "from geopolars_amd import synth\n"
        "from geopolars_amd.geoseries import GeoSeries\n"
        "from geopolars_amd.spatial_index import SpatialIndex, join_pairs\n"
        "from oracle import pyoracle\n"
        "pyoracle.build()\n"
        "polys = synth.star_polygons(1000, 64); pts = synth.uniform_points(400_013, seed=23)\n"
        "xy = pts.xy.copy(); xy[::1000] = polys.xy[np.arange(0, len(xy), 1000) % len(polys.xy)]  # some points ON vertices\n"
        "from geopolars_amd.geoarrow import GeoArrowArray\n"
        "pts = GeoArrowArray.from_points(xy)\n"
        "right = GeoSeries(polys); index = SpatialIndex(right)\n"
        "assert index.describe()['route']\n"
        "ep, ec, _ = pyoracle.spatial_join(pts, polys, 'intersects', mode=0)\n"
        "gp, gc = join_pairs(GeoSeries(pts), right, 'intersects', r_index=index)\n"
        "assert np.array_equal(gc, ec) and np.array_equal(gp, ep)\n"
        "print('ok', int(ec.sum()))\n"
    )
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, GPK_TILE_KERNEL=kernel))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]
