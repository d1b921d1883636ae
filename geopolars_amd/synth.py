"""Synthetic workloads of BASELINE.json `configs` (definitions: SURVEY.md §8d / BASELINE.md §3).

All generators are seeded (`numpy.random.default_rng`, PCG64; default seed 20241008) and scale-free:
tests call them at sizes the CPU oracle finishes in seconds, bench.py at the configured sizes.
"""
from __future__ import annotations

import numpy as np

from ._abi import GEOM_LINESTRING, GEOM_MULTIPOLYGON, GEOM_POLYGON
from .geoarrow import GeoArrowArray

SEED = 20241008
DOMAIN = 1000.0


def star_polygons(n_polys: int = 1000, n_verts: int = 64, seed: int = SEED, domain: float = DOMAIN) -> GeoArrowArray:
    """C2 right side: simple star-shaped polygons, CCW, closed (n_verts + 1 coords), no holes, one per
    cell of a jittered square grid -> non-overlapping, each point in at most one polygon."""
    rng = np.random.default_rng(seed)
    g = int(np.ceil(np.sqrt(n_polys)))
    cell = domain / g
    k = np.arange(n_polys)
    cx = (k % g + 0.5) * cell + rng.uniform(-0.05, 0.05, n_polys) * cell
    cy = (k // g + 0.5) * cell + rng.uniform(-0.05, 0.05, n_polys) * cell
    u = rng.uniform(0.0, 0.9, (n_polys, n_verts))
    ang = 2.0 * np.pi * (np.arange(n_verts)[None, :] + u) / n_verts
    rad = rng.uniform(0.5, 1.0, (n_polys, n_verts)) * 0.45 * cell
    x = cx[:, None] + rad * np.cos(ang)
    y = cy[:, None] + rad * np.sin(ang)
    xy = np.empty((n_polys, n_verts + 1, 2), dtype=np.float64)
    xy[:, :n_verts, 0] = x
    xy[:, :n_verts, 1] = y
    xy[:, n_verts] = xy[:, 0]
    ring_off = np.arange(0, n_polys * (n_verts + 1) + 1, n_verts + 1, dtype=np.int32)
    geom_off = np.arange(n_polys + 1, dtype=np.int32)
    return GeoArrowArray(GEOM_POLYGON, xy.reshape(-1, 2), geom_offsets=geom_off, ring_offsets=ring_off)


def uniform_points(n: int, seed: int = SEED + 1, domain: float = DOMAIN) -> GeoArrowArray:
    rng = np.random.default_rng(seed)
    return GeoArrowArray.from_points(rng.uniform(0.0, domain, (n, 2)))


def adversarial_points(polys: GeoArrowArray, seed: int = SEED + 2, per_poly: int = 4) -> GeoArrowArray:
    """Points exactly on vertices, on edge midpoints (as representable), level with horizontal extremes
    and at bbox corners of the first rings — the degenerate cases of coord_pos_relative_to_ring."""
    rng = np.random.default_rng(seed)
    pts = []
    ro = polys.ring_offsets
    n_r = min(polys.n_rings, 64)
    for r in range(n_r):
        ring = polys.xy[ro[r] : ro[r + 1]]
        if len(ring) < 2:
            continue
        idx = rng.integers(0, len(ring) - 1, per_poly)
        for i in idx:
            a, b = ring[i], ring[i + 1]
            pts.append(a)  # vertex
            pts.append((a + b) / 2.0)  # (nearly) on the edge
            pts.append([a[0] - 1.0, a[1]])  # level with a vertex, to its left
            pts.append([a[0] + 1.0, a[1]])  # level with a vertex, to its right
        mn, mx = ring.min(axis=0), ring.max(axis=0)
        pts += [mn, mx, [mn[0], mx[1]], [mx[0], mn[1]], (mn + mx) / 2.0]
    return GeoArrowArray.from_points(np.array(pts, dtype=np.float64))


def random_linestrings(n_lines: int = 100_000, seed: int = SEED + 3, domain: float = DOMAIN, min_log2: float = 2.0, max_log2: float = 8.0) -> GeoArrowArray:
    """C3 right side: segment counts round(2^U(2,8)) in [4, 256] (log-uniform, mean ~61), random-walk
    vertices (step ~U(0,5), uniform heading) from a uniform start."""
    rng = np.random.default_rng(seed)
    segs = np.clip(np.round(2.0 ** rng.uniform(min_log2, max_log2, n_lines)), 4, 256).astype(np.int64)
    nv = segs + 1
    off = np.zeros(n_lines + 1, dtype=np.int64)
    off[1:] = np.cumsum(nv)
    total = int(off[-1])
    step = rng.uniform(0.0, 5.0, total)
    head = rng.uniform(0.0, 2.0 * np.pi, total)
    dx, dy = step * np.cos(head), step * np.sin(head)
    start = rng.uniform(0.0, domain, (n_lines, 2))
    first = off[:-1]
    dx[first] = start[:, 0]
    dy[first] = start[:, 1]
    # segmented cumulative sum: global cumsum minus the running total before each line's start
    cx, cy = np.cumsum(dx), np.cumsum(dy)
    bx = np.repeat(cx[first] - dx[first], nv)
    by = np.repeat(cy[first] - dy[first], nv)
    xy = np.stack([cx - bx, cy - by], axis=1)
    return GeoArrowArray(GEOM_LINESTRING, xy, geom_offsets=off.astype(np.int32))


def clustered_polygons(n: int, seed: int = SEED + 4, domain: float = DOMAIN, min_verts: int = 8, max_verts: int = 64, mean_neighbours: float = 4.0) -> GeoArrowArray:
    """C4 sides: convex-ish star polygons with 8-64 vertices, sized so that a polygon's bbox overlaps
    about `mean_neighbours` others when two such sets are overlaid."""
    rng = np.random.default_rng(seed)
    nv = rng.integers(min_verts, max_verts + 1, n)
    radius = np.sqrt(mean_neighbours / max(n, 1) / np.pi) * domain / 2.0
    cx, cy = rng.uniform(0.0, domain, n), rng.uniform(0.0, domain, n)
    ring_off = np.zeros(n + 1, dtype=np.int64)
    ring_off[1:] = np.cumsum(nv + 1)
    total = int(ring_off[-1])
    pid = np.repeat(np.arange(n), nv + 1)
    j = np.arange(total) - np.repeat(ring_off[:-1], nv + 1)
    nvp = nv[pid]
    jj = np.where(j == nvp, 0, j)  # closing vertex repeats vertex 0
    # per-vertex jitter must be reproducible for the closing vertex: derive it from (pid, jj)
    u = rng.uniform(0.0, 0.9, total)
    r = rng.uniform(0.6, 1.0, total) * radius
    firsts = np.repeat(ring_off[:-1], nv + 1)
    u = np.where(j == nvp, u[firsts], u)
    r = np.where(j == nvp, r[firsts], r)
    ang = 2.0 * np.pi * (jj + u) / nvp
    xy = np.stack([cx[pid] + r * np.cos(ang), cy[pid] + r * np.sin(ang)], axis=1)
    return GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=np.arange(n + 1, dtype=np.int32), ring_offsets=ring_off.astype(np.int32))


def powerlaw_multipolygons(n: int, seed: int = SEED + 5, domain: float = DOMAIN, alpha: float = 1.5, min_verts: int = 4, cap: int = 100_000, max_parts: int = 3, hole_prob: float = 0.15, size_n: int | None = None) -> GeoArrowArray:
    """C5 right side: multipolygons whose ring vertex counts follow Pareto(alpha) (min 4, capped), 1-3
    member polygons each, some with one hole (a scaled copy of the exterior, CW).  Member radii scale with
    domain / sqrt(size_n) (default n): a column generated in chunks passes the column's total row count, so that the
    chunks together cover the domain like one column of that size."""
    rng = np.random.default_rng(seed)
    n_parts = rng.integers(1, max_parts + 1, n)
    geom_off = np.zeros(n + 1, dtype=np.int64)
    geom_off[1:] = np.cumsum(n_parts)
    P = int(geom_off[-1])
    has_hole = rng.uniform(size=P) < hole_prob
    part_off = np.zeros(P + 1, dtype=np.int64)
    part_off[1:] = np.cumsum(1 + has_hole.astype(np.int64))
    R = int(part_off[-1])
    nv_part = np.minimum((min_verts * (1.0 + rng.pareto(alpha, P))).astype(np.int64), cap)
    nv_part = np.maximum(nv_part, min_verts)
    ring_part = np.repeat(np.arange(P), 1 + has_hole.astype(np.int64))
    is_hole = np.zeros(R, dtype=bool)
    is_hole[part_off[:-1][has_hole] + 1] = True
    nv_ring = nv_part[ring_part]
    ring_off = np.zeros(R + 1, dtype=np.int64)
    ring_off[1:] = np.cumsum(nv_ring + 1)
    total = int(ring_off[-1])
    gid_part = np.repeat(np.arange(n), n_parts)
    gx, gy = rng.uniform(0.0, domain, n), rng.uniform(0.0, domain, n)
    # parts of one multipolygon sit side by side so they do not overlap
    k_in_geom = np.arange(P) - np.repeat(geom_off[:-1], n_parts)
    base_r = rng.uniform(0.2, 1.0, P) * domain / np.sqrt(max(size_n or n, 1)) * 0.5
    pcx = gx[gid_part] + k_in_geom * 2.2 * base_r
    pcy = gy[gid_part]
    rid = np.repeat(np.arange(R), nv_ring + 1)
    j = np.arange(total) - np.repeat(ring_off[:-1], nv_ring + 1)
    nvr = nv_ring[rid]
    jj = np.where(j == nvr, 0, j)
    # radius profile is a deterministic function of (part, angle) so the hole is a scaled copy
    part_of = ring_part[rid]
    phase = rng.uniform(0.0, 2.0 * np.pi, P)
    ang = 2.0 * np.pi * jj / nvr
    prof = 0.8 + 0.2 * np.sin(3.0 * ang + phase[part_of])
    scale = np.where(is_hole[rid], 0.4, 1.0)
    ang_dir = np.where(is_hole[rid], -ang, ang)  # holes run clockwise
    prof = np.where(is_hole[rid], 0.8 + 0.2 * np.sin(3.0 * (-ang_dir) + phase[part_of]), prof)
    r = base_r[part_of] * prof * scale
    xy = np.stack([pcx[part_of] + r * np.cos(ang_dir), pcy[part_of] + r * np.sin(ang_dir)], axis=1)
    return GeoArrowArray(
        GEOM_MULTIPOLYGON,
        xy,
        geom_offsets=geom_off.astype(np.int32),
        part_offsets=part_off.astype(np.int32),
        ring_offsets=ring_off.astype(np.int32),
    )


def tessellation(g: int = 32, sub: int = 16, seed: int = SEED + 6, domain: float = DOMAIN) -> GeoArrowArray:
    """g x g quads covering the domain whose sides are SHARED jittered polylines of `sub` segments: neighbours have every
    boundary vertex in common, every point of the domain lies in exactly one polygon or on a shared border (administrative
    boundaries, census tracts).  4 * sub distinct vertices per polygon."""
    rng = np.random.default_rng(seed)
    cell = domain / g
    t = np.arange(1, sub) / sub
    amp = 0.2 * cell / sub
    # interior points of every horizontal side (row j, column i) and vertical side (row j, column i); the outer frame stays straight
    hx = np.empty((g + 1, g, sub - 1, 2))
    vx = np.empty((g, g + 1, sub - 1, 2))
    hx[..., 0] = (np.arange(g)[None, :, None] + t[None, None, :]) * cell
    hx[..., 1] = np.arange(g + 1)[:, None, None] * cell + rng.uniform(-amp, amp, (g + 1, g, sub - 1)) * ((np.arange(g + 1) > 0) & (np.arange(g + 1) < g))[:, None, None]
    vx[..., 1] = (np.arange(g)[:, None, None] + t[None, None, :]) * cell
    vx[..., 0] = np.arange(g + 1)[None, :, None] * cell + rng.uniform(-amp, amp, (g, g + 1, sub - 1)) * ((np.arange(g + 1) > 0) & (np.arange(g + 1) < g))[None, :, None]
    n_v = 4 * sub + 1
    xy = np.empty((g * g, n_v, 2))
    for j in range(g):
        for i in range(g):
            ring = np.concatenate(
                [[(i * cell, j * cell)], hx[j, i], [((i + 1) * cell, j * cell)], vx[j, i + 1], [((i + 1) * cell, (j + 1) * cell)], hx[j + 1, i][::-1],
                 [(i * cell, (j + 1) * cell)], vx[j, i][::-1], [(i * cell, j * cell)]]
            )
            xy[j * g + i] = ring
    offs = (np.arange(g * g + 1) * n_v).astype(np.int32)
    return GeoArrowArray(GEOM_POLYGON, xy.reshape(-1, 2), geom_offsets=np.arange(g * g + 1, dtype=np.int32), ring_offsets=offs)
