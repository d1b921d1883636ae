#!/usr/bin/env python3
"""Robustness sweep on a real MI355X: data regimes far from the benchmark configs (few huge rings, hundreds of
thousands of tiny ones, identical geometries, zero-area rings, extreme magnitudes, one geometry with thousands of
parts or holes, all-null / all-empty columns, skewed row maps), every result compared with the CPU oracle.

    python tests/stress_sweep.py [--start K] [--only NAME]

Each regime prints its name BEFORE it runs (a GPU memory fault aborts the process: the last name printed is the
culprit; rerun with GPK_DEBUG_SYNC=1 --only NAME to get the kernel).  Exit code 0 = every regime agreed.
Lives under tests/ because it checks against the CPU oracle (test infrastructure); pytest does not collect it."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)

from geopolars_amd import _abi, synth  # noqa: E402
from geopolars_amd.geoarrow import GeoArrowArray  # noqa: E402
from geopolars_amd.geoseries import GeoSeries  # noqa: E402
from geopolars_amd.spatial_index import join_pairs  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

REGIMES = []


def regime(f):
    REGIMES.append(f)
    return f


DRY = False  # --dry: oracle side only (checks the generators and the oracle's run time without a GPU)


def check_join(left, right, pred="intersects", mode=1):
    exp_pairs, exp_counts, _ = oracle.spatial_join(left, right, pred, mode=mode)
    if DRY:
        return len(exp_pairs)
    got_pairs, got_counts = join_pairs(GeoSeries(left), GeoSeries(right), pred)
    assert np.array_equal(got_counts, exp_counts), (pred, int(got_counts.sum()), int(exp_counts.sum()))
    assert np.array_equal(got_pairs, exp_pairs), pred
    return len(exp_pairs)


def close(got, exp, what, rtol=1e-9):
    got, exp = np.asarray(got, float), np.asarray(exp, float)
    assert got.shape == exp.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(exp)), what + ": NaN pattern"
    m = ~np.isnan(exp)
    assert not (np.abs(got[m] - exp[m]) > rtol * np.maximum(np.abs(exp[m]), 1e-300)).any(), what


def canon(ring):
    ring = ring[:-1] if len(ring) > 1 and np.array_equal(ring[0], ring[-1]) else ring
    if len(ring) == 0:
        return ring
    return np.roll(ring, -np.lexsort((ring[:, 1], ring[:, 0]))[0], axis=0)


def check_unary(a):
    polygonal = a.geom_type in (_abi.GEOM_POLYGON, _abi.GEOM_MULTIPOLYGON)
    e_bounds, e_len = oracle.bounds(a), oracle.euclidean_length(a)
    e_area = oracle.area(a) if polygonal else None
    e_c, _ = oracle.centroid(a)
    hx, ho = oracle.convex_hull(a)
    if DRY:
        return
    s = GeoSeries(a)
    assert np.array_equal(s.bounds(), e_bounds, equal_nan=True), "bounds"
    if polygonal:
        close(s.area(), e_area, "area")
    close(s.euclidean_length(), e_len, "length")
    close(s.centroid().array.xy, e_c, "centroid")
    h = s.convex_hull().array
    assert np.array_equal(h.ring_offsets, ho), "hull offsets"
    if not np.array_equal(h.xy, hx):
        for g in range(len(a)):
            assert np.array_equal(canon(h.xy[ho[g] : ho[g + 1]]), canon(hx[ho[g] : ho[g + 1]])), f"hull of row {g}"


def squares(cx, cy, half):
    cx, cy, half = np.broadcast_arrays(np.asarray(cx, float), np.asarray(cy, float), np.asarray(half, float))
    n = len(cx)
    xy = np.empty((n, 5, 2))
    for k, (sx, sy) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1), (-1, -1))):
        xy[:, k, 0] = cx + sx * half
        xy[:, k, 1] = cy + sy * half
    return GeoArrowArray(_abi.GEOM_POLYGON, xy.reshape(-1, 2), geom_offsets=np.arange(n + 1, dtype=np.int32), ring_offsets=np.arange(0, 5 * n + 1, 5, dtype=np.int32))


# ---- point x polygon -------------------------------------------------------------------------------------------------
@regime
def random_disjoint_right_sides_against_adversarial_point_mixes():
    """the one-launch point join (gpk_pipflow.hip) on forty random draws: disjoint star polygons of random count / vertex count / scale
    (lean right sides with chains and an LDS routing image), point columns mixing uniform points, points packed along edges, points ON
    vertices and edge midpoints, NaN rows, duplicates — lengths round the tile-size thresholds, a random left_row_base"""
    rng = np.random.default_rng(606)
    total = 0
    for it in range(40):
        n_polys = int(rng.choice([1, 3, 17, 120, 900, 2500]))
        n_verts = int(rng.choice([3, 4, 7, 24, 64, 200]))
        polys = synth.star_polygons(n_polys, n_verts, seed=1000 + it, domain=float(rng.choice([1.0, 1000.0, 3.0e6])))
        v = polys.xy
        n_uni = int(rng.choice([0, 1, 63, 64, 65, 4097, 70_000, 300_000]))
        lo, hi = v.min(axis=0), v.max(axis=0)
        parts = [rng.uniform(lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo), (n_uni, 2))]
        k = int(rng.integers(0, 4000))
        if k:
            i = rng.integers(0, len(v) - 1, k)
            t = rng.uniform(0, 1, (k, 1))
            along = v[i] * (1 - t) + v[i + 1] * t  # (pairs that straddle two rings are just more points)
            parts += [along + rng.normal(0, 1e-7 * float(hi[0] - lo[0] + 1e-30), (k, 2)), v[i], (v[i] + v[i + 1]) / 2.0]
        xy = np.concatenate(parts)
        if len(xy):
            xy[rng.integers(0, len(xy), max(1, len(xy) // 500))] = np.nan
            dup = rng.integers(0, len(xy), len(xy) // 50)
            xy = np.concatenate([xy, xy[dup]])
            rng.shuffle(xy)
        pts = GeoArrowArray.from_points(xy)
        exp_pairs, exp_counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
        total += len(exp_pairs)
        if DRY:
            continue
        base = int(rng.choice([0, 5, 1 << 20]))
        got_pairs, got_counts = join_pairs(GeoSeries(pts), GeoSeries(polys), "intersects", left_row_base=base)
        e = exp_pairs.copy()
        e[:, 0] += base
        assert np.array_equal(got_counts, exp_counts), (it, n_polys, n_verts, len(xy))
        assert np.array_equal(got_pairs, e), (it, n_polys, n_verts, len(xy))
    return total


@regime
def many_tiny_polygons_few_points():
    rng = np.random.default_rng(1)
    c = rng.uniform(0, 1000, (300_000, 2))
    polys = squares(c[:, 0], c[:, 1], 0.05)
    pts = np.concatenate([c[:500], c[500:1000] + 0.05, rng.uniform(0, 1000, (500, 2))])
    return check_join(GeoArrowArray.from_points(pts), polys)


@regime
def identical_polygons():
    polys = squares(np.full(700, 500.0), np.full(700, 500.0), 100.0)
    rng = np.random.default_rng(2)
    pts = np.concatenate([rng.uniform(350, 650, (20_000, 2)), [[400.0, 400.0], [600.0, 500.0], [500.0, 500.0]]])
    n = check_join(GeoArrowArray.from_points(pts), polys)
    assert n > 700 * 1000
    return n


@regime
def identical_points():
    polys = synth.star_polygons(400, 32)
    v = polys.xy[17]
    inside = polys.xy[0:32].mean(axis=0)
    pts = np.concatenate([np.tile(v, (100_000, 1)), np.tile(inside, (100_000, 1))])
    return check_join(GeoArrowArray.from_points(pts), polys)


@regime
def zero_area_and_spike_polygons():
    polys = GeoArrowArray.from_polygons(
        [
            [[(0, 0), (10, 0), (20, 0)]],  # collinear
            [[(0, 5), (10, 5), (10, 5), (0, 5)]],  # out and back
            [[(30, 30), (30, 30), (30, 30)]],  # one point, repeated
            [[(40, 40), (60, 40), (60, 60), (50, 60), (50, 80), (50, 60), (40, 60)]],  # spike
            [[(0, 100), (100, 100), (100, 200), (0, 200)]],
        ]
    )
    gx, gy = np.meshgrid(np.arange(-5, 106, 2.5), np.arange(-5, 206, 2.5))
    pts = GeoArrowArray.from_points(np.stack([gx.ravel(), gy.ravel()], 1))
    n = check_join(pts, polys, mode=0)
    check_unary(polys)
    return n


@regime
def extreme_magnitudes():
    total = 0
    base = synth.star_polygons(200, 24)
    pts = synth.uniform_points(50_000)
    for scale, off in ((2.0**40, 0.0), (2.0**-40, 0.0), (1.0, 2.0**33), (2.0**-20, 2.0**20), (2.0**300, 0.0), (2.0**-300, 0.0)):
        p = GeoArrowArray(base.geom_type, base.xy * scale + off, geom_offsets=base.geom_offsets, ring_offsets=base.ring_offsets)
        q = GeoArrowArray.from_points(pts.xy * scale + off)
        total += check_join(q, p)
    return total


@regime
def one_ring_with_100k_vertices():
    n = 100_000
    rng = np.random.default_rng(5)
    ang = 2 * np.pi * (np.arange(n) + rng.uniform(0, 0.9, n)) / n
    rad = rng.uniform(300, 450, n)
    ring = np.stack([500 + rad * np.cos(ang), 500 + rad * np.sin(ang)], 1)
    ring = np.concatenate([ring, ring[:1]])
    polys = GeoArrowArray(_abi.GEOM_POLYGON, ring, geom_offsets=np.array([0, 1], np.int32), ring_offsets=np.array([0, n + 1], np.int32))
    pts = np.concatenate([rng.uniform(0, 1000, (20_000, 2)), ring[::300]])
    m = check_join(GeoArrowArray.from_points(pts), polys)
    check_unary(polys)
    return m


@regime
def polygon_with_2000_holes_and_multipolygon_with_3000_parts():
    rng = np.random.default_rng(6)
    g = np.arange(45)
    hx, hy = np.meshgrid(10 + 20 * g, 10 + 20 * g)
    hx, hy = hx.ravel()[:2000], hy.ravel()[:2000]
    holes = [[(x - 4, y - 4), (x - 4, y + 4), (x + 4, y + 4), (x + 4, y - 4)] for x, y in zip(hx, hy)]
    frame = GeoArrowArray.from_polygons([[[(0, 0), (1000, 0), (1000, 1000), (0, 1000)]] + holes])
    pts = np.concatenate([rng.uniform(0, 1000, (60_000, 2)), np.stack([hx, hy], 1), np.stack([hx - 4.0, hy], 1)])
    n = check_join(GeoArrowArray.from_points(pts), frame)
    c = rng.uniform(0, 1000, (3000, 2))
    parts = [[[(x - 3, y - 3), (x + 3, y - 3), (x + 3, y + 3), (x - 3, y + 3)]] for x, y in c]
    multi = GeoArrowArray.from_multipolygons([parts, parts[:5], []])
    n += check_join(GeoArrowArray.from_points(np.concatenate([pts[:60_000], c])), multi)
    check_unary(frame)
    check_unary(multi)
    return n


@regime
def null_and_empty_columns():
    polys = synth.star_polygons(50, 16)
    pts = synth.uniform_points(5000)
    zero_p = np.zeros((len(polys) + 7) // 8, np.uint8)
    zero_q = np.zeros((len(pts) + 7) // 8, np.uint8)
    pn = GeoArrowArray(polys.geom_type, polys.xy, geom_offsets=polys.geom_offsets, ring_offsets=polys.ring_offsets, validity=zero_p)
    qn = GeoArrowArray.from_points(pts.xy, validity=zero_q)
    assert check_join(qn, polys) == 0 and check_join(pts, pn) == 0 and check_join(pn, pts) == 0
    empties = GeoArrowArray.from_polygons([[] for _ in range(40)])
    assert check_join(pts, empties) == 0 and check_join(empties, pts) == 0 and check_join(empties, empties) == 0
    nanpts = GeoArrowArray.from_points(np.full((300, 2), np.nan))
    assert check_join(nanpts, polys) == 0
    check_unary(empties)
    check_unary(pn)
    return 0


# ---- polygon x polygon -------------------------------------------------------------------------------------------------
@regime
def one_huge_polygon_against_many_small():
    rng = np.random.default_rng(8)
    huge = synth.star_polygons(1, 64)  # one star over the whole domain
    c = rng.uniform(0, 1000, (150_000, 2))
    small = squares(c[:, 0], c[:, 1], 0.5)
    n = 0
    for pred in ("intersects", "contains"):
        n += check_join(huge, small, pred)
        n += check_join(small, huge, pred)
    return n


@regime
def a_column_against_itself():
    a = synth.clustered_polygons(20_000, seed=9, mean_neighbours=6.0)
    n = check_join(a, a, "intersects") + check_join(a, a, "contains")
    m = synth.powerlaw_multipolygons(3000, seed=10)
    return n + check_join(m, m, "intersects")


@regime
def tessellation_against_itself():
    t = synth.tessellation(12, 8)
    n = check_join(t, t, "intersects")
    k = check_join(t, t, "contains")
    assert k == len(t)  # shared borders: every cell contains itself and nothing else
    return n + k


# ---- row-wise ------------------------------------------------------------------------------------------------------------
@regime
def distance_with_skewed_and_degenerate_targets():
    rng = np.random.default_rng(11)
    lines = synth.random_linestrings(2000, seed=12)
    extra = GeoArrowArray.from_linestrings([[], [(5.0, 5.0)], [(1.0, 1.0), (1.0, 1.0)], [(0.0, 0.0), (1000.0, 1000.0)]])
    xy = np.concatenate([lines.xy, extra.xy])
    off = np.concatenate([lines.geom_offsets, extra.geom_offsets[1:] + lines.geom_offsets[-1]]).astype(np.int32)
    ls = GeoArrowArray(_abi.GEOM_LINESTRING, xy, geom_offsets=off)
    pts = rng.uniform(0, 1000, (200_000, 2))
    pts[::1000] = np.nan
    p = GeoArrowArray.from_points(pts)
    total = 0
    for rows in (np.full(len(p), 7, np.uint32), np.full(len(p), len(ls) - 4, np.uint32), rng.integers(len(ls) - 4, len(ls), len(p)).astype(np.uint32), (rng.zipf(1.3, len(p)) % len(ls)).astype(np.uint32)):
        e = oracle.distance_rowwise(p, ls, rows)
        total += int(np.isfinite(e).sum())
        if DRY:
            continue
        g = GeoSeries(p).distance(GeoSeries(ls), rows)
        assert np.array_equal(g == 0, e == 0), "distance: zero pattern"
        close(g, e, "distance")
    return total


@regime
def unary_ops_on_ragged_columns():
    rng = np.random.default_rng(13)
    polys = []
    for i in range(3000):
        k = int(rng.choice([0, 1, 2, 3, 4, 5, 40, 700]))
        if k == 0:
            polys.append([])
            continue
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        r = rng.uniform(1, 5, k)
        cx, cy = rng.uniform(0, 1000, 2)
        ring = [(cx + a * np.cos(t), cy + a * np.sin(t)) for a, t in zip(r, ang)]
        polys.append([ring] + ([[(cx - 0.1, cy - 0.1), (cx - 0.1, cy + 0.1), (cx + 0.1, cy + 0.1)]] if k >= 40 else []))
    a = GeoArrowArray.from_polygons(polys)
    check_unary(a)
    mp = GeoArrowArray.from_multipolygons([polys[i : i + int(rng.integers(0, 6))] for i in range(0, 2990, 5)])
    check_unary(mp)
    ls = GeoArrowArray.from_linestrings([p[0] if p else [] for p in polys])
    check_unary(ls)
    return len(a) + len(mp) + len(ls)


@regime
def wkb_round_trip_of_ragged_columns():
    rng = np.random.default_rng(14)
    m = synth.powerlaw_multipolygons(5000, seed=15)
    total = 0
    for arr in (m, synth.random_linestrings(5000, seed=16), synth.uniform_points(5000), synth.star_polygons(300, 7)):
        if DRY:
            continue
        values, offsets = GeoSeries(arr).to_wkb()
        back = GeoSeries.from_wkb_device(values, offsets).array
        assert back.geom_type == arr.geom_type and np.array_equal(back.xy, arr.xy) and np.array_equal(back.geom_offsets, arr.geom_offsets)
        hv, ho = GeoSeries(arr).to_wkb(on_device=False)
        assert np.array_equal(hv, values) and np.array_equal(ho, offsets)
        total += len(values)
    return total


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--only", default=None)
    ap.add_argument("--dry", action="store_true", help="oracle side only (no GPU)")
    args = ap.parse_args()
    global DRY
    DRY = args.dry
    oracle.build()
    if not DRY:
        _abi.lib()
        print("device:", _abi.device_info(), flush=True)
    failed = []
    for k, f in enumerate(REGIMES):
        if k < args.start or (args.only and f.__name__ != args.only):
            continue
        print(f"[{k}] {f.__name__} ...", flush=True)
        t0 = time.perf_counter()
        try:
            r = f()
            print(f"[{k}] {f.__name__} ok ({r}) {time.perf_counter() - t0:.1f}s", flush=True)
        except AssertionError as e:
            failed.append(f.__name__)
            print(f"[{k}] {f.__name__} MISMATCH {e!r}", flush=True)
        except Exception as e:  # an error code from the library is a finding too
            failed.append(f.__name__)
            print(f"[{k}] {f.__name__} ERROR {type(e).__name__}: {e}", flush=True)
    print("failed:", failed, flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
