#!/usr/bin/env python3
"""CPU prototype for DESIGN section 7, "test sub-cells decided in the owning lane" (no GPU, no product code).

Claim checked here: for a ring and a closed rectangle Q (a padded level-2 sub-cell), let T be the ring edges that meet Q and let
the LOCAL CHAIN C be T grown along the ring, at both ends of every run, while the run's end vertex has its y inside Q's closed
y-interval.  Then for every point p of Q

        winding(p) = base(Q) + sum over e in C of c_e(p),     p on the ring  <=>  p on an edge of C,

where c_e is the per-edge contribution of geo's coord_pos_relative_to_ring (dev::ring_edge) and base(Q), the summed contribution
of all other edges, does NOT depend on p: a run of non-chain edges is a polyline that misses Q and whose two end vertices lie
outside Q's y-interval, so moving p inside Q can neither cross it nor move the ray past one of its ends, and at its inner
vertices the half-open rule hands the crossing from one edge to the next.  A lane could then decide a "test" point from the
handful of edges of C plus one stored integer, with no queue and no lane group.

The script builds C and base for sampled (polygon, sub-cell) pairs of the C2 right side with plain float arithmetic (the point
is the combinatorics, not exactness) and compares against the full ring walk on random points of Q, including points ON chain
edges' supporting lines.  It also reports how long the chains are (the cost of the idea)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import synth


def orient(ax, ay, bx, by, cx, cy):
    d = (ax - cx) * (by - cy) - (ay - cy) * (bx - cx)
    return int(d > 0) - int(d < 0)


def ring_edge(sx, sy, ex, ey, cx, cy):
    """(winding contribution, on_boundary) of one edge: gpk_device.h dev::ring_edge"""
    up = sy <= cy and ey >= cy
    down = sy > cy and ey <= cy
    if not (up or down):
        return 0, False
    lo, hi = min(sx, ex), max(sx, ex)
    if cx < lo:
        return ((1 if ey != cy else 0) if up else -1), False
    if not (cx <= hi):
        return 0, False
    o = orient(sx, sy, ex, ey, cx, cy)
    if o == 0:
        return 0, True
    if up:
        return (1 if (o > 0 and ey != cy) else 0), False
    return (-1 if o < 0 else 0), False


def seg_meets_rect(ax, ay, bx, by, xl, yl, xh, yh):
    if max(ax, bx) < xl or min(ax, bx) > xh or max(ay, by) < yl or min(ay, by) > yh:
        return False
    o = [orient(ax, ay, bx, by, x, y) for x, y in ((xl, yl), (xh, yl), (xh, yh), (xl, yh))]
    return not (all(v > 0 for v in o) or all(v < 0 for v in o))


def local_chain(ring, xl, yl, xh, yh):
    """indices of the edges of the local chain (edge i = ring[i] -> ring[i + 1]; ring closed: ring[-1] == ring[0])"""
    n = len(ring) - 1
    T = [i for i in range(n) if seg_meets_rect(*ring[i], *ring[i + 1], xl, yl, xh, yh)]
    if not T:
        return []
    C = set(T)
    for i in T:
        j = i  # grow forward: the end vertex of edge j is ring[j + 1]
        while yl <= ring[(j + 1) % n][1] <= yh and len(C) < n:
            j = (j + 1) % n
            C.add(j)
        j = i  # grow backward: the start vertex of edge j is ring[j]
        while yl <= ring[j][1] <= yh and len(C) < n:
            j = (j - 1) % n
            C.add(j)
    return sorted(C)


CHAIN_MAX = 12


def local_arc(ring, xl, yl, xh, yh):
    """the chain as the BUILD KERNEL forms it (chain_aux_kernel, gpk_pipindex.hip): the shortest arc of the CYCLE of ring edges
    that covers every touching edge (the complement of the widest gap between one touched edge and the next), grown at both
    ends by the y-rule, at most CHAIN_MAX edges; None = no chain entry (the generic walk decides such points)"""
    n = len(ring) - 1  # edges 0 .. n - 1 (vertex n closes the ring: ring[n] == ring[0])
    T = sorted({i for i in range(n) if seg_meets_rect(*ring[i], *ring[i + 1], xl, yl, xh, yh)})
    if not T:
        return None
    best_gap, lo = -1, T[0]
    for a in T:
        nd, nb = n, a
        for b in T:
            d = (b - a) % n
            if 0 < d < nd:
                nd, nb = d, b
        if nd > best_gap:
            best_gap, lo = nd, nb
    length = n - best_gap + 1
    if length > CHAIN_MAX:
        return None
    while length < n:
        if not (yl <= ring[(lo + length) % n][1] <= yh):
            break
        length += 1
        if length > CHAIN_MAX:
            return None
    while length < n:
        if not (yl <= ring[lo][1] <= yh):
            break
        lo = (lo - 1) % n
        length += 1
        if length > CHAIN_MAX:
            return None
    return [(lo + j) % n for j in range(length)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    as_kernel = "--as-kernel" in sys.argv
    rng = np.random.default_rng(7)
    if which == "c2":
        polys, R = synth.star_polygons(1000, 64), 512
    elif which == "small":  # polygons of a few sub-cells: chains that wrap most of the ring
        polys, R = synth.star_polygons(1000, 16), 64
    else:  # 8-64-vertex clustered polygons (the C4 shape)
        polys, R = synth.clustered_polygons(1000), 512
    ro = polys.ring_offsets
    S = 8
    xy = polys.xy
    x0, y0 = xy.min(0)
    x1, y1 = xy.max(0)
    fw, fh = (x1 - x0) / (R - 3), (y1 - y0) / (R - 3)
    rx0, ry0 = x0 - 1.5 * fw, y0 - 1.5 * fh
    sw, sh = fw / S, fh / S
    pad_x, pad_y = fw / 65536.0 / S, fh / 65536.0 / S
    chain_len, bad, checked, cells, fallbacks = [], 0, 0, 0, 0
    print(f"workload {which}: {len(polys)} polygons, raster {R}")
    for g in rng.choice(len(polys), 60, replace=False):
        ring = [(float(p[0]), float(p[1])) for p in xy[ro[g] : ro[g + 1]]]
        n = len(ring) - 1
        # sub-cells along the boundary: take the sub-cell of a point near each edge's midpoint and of each vertex
        seen = set()
        for i in range(n):
            for t in (0.0, 0.37, 0.5, 0.81):
                px = ring[i][0] + t * (ring[i + 1][0] - ring[i][0])
                py = ring[i][1] + t * (ring[i + 1][1] - ring[i][1])
                seen.add((int((px - rx0) / sw), int((py - ry0) / sh)))
        for (si, sj) in list(seen)[:120]:
            xl, xh = rx0 + si * sw - pad_x, rx0 + (si + 1) * sw + pad_x
            yl, yh = ry0 + sj * sh - pad_y, ry0 + (sj + 1) * sh + pad_y
            C = local_arc(ring, xl, yl, xh, yh) if as_kernel else local_chain(ring, xl, yl, xh, yh)
            if C is None:
                fallbacks += 1
                continue
            if not C:
                continue
            cells += 1
            chain_len.append(len(C))
            Cs = set(C)
            cx, cy = 0.5 * (xl + xh), 0.5 * (yl + yh)
            base = sum(ring_edge(*ring[i], *ring[i + 1], cx, cy)[0] for i in range(n) if i not in Cs)
            pts = np.stack([rng.uniform(xl, xh, 24), rng.uniform(yl, yh, 24)], axis=1).tolist()
            for i in C[:3]:  # points on (the float image of) chain edges, and level with chain vertices
                a, b = ring[i], ring[i + 1]
                for t in (0.0, 0.5, 1.0):
                    q = (a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]))
                    if xl <= q[0] <= xh and yl <= q[1] <= yh:
                        pts.append(q)
                        pts.append((min(max(q[0] - sw / 3, xl), xh), q[1]))
            for px, py in pts:
                full = [ring_edge(*ring[i], *ring[i + 1], px, py) for i in range(n)]
                wn_full, on_full = sum(v[0] for v in full), any(v[1] for v in full)
                loc = [ring_edge(*ring[i], *ring[i + 1], px, py) for i in C]
                wn_loc, on_loc = base + sum(v[0] for v in loc), any(v[1] for v in loc)
                checked += 1
                if on_full != on_loc or (not on_full and (wn_full != 0) != (wn_loc != 0)) or (not on_full and wn_full != wn_loc):
                    bad += 1
    cl = np.array(chain_len)
    print(f"sub-cells with a chain: {cells}; points checked: {checked}; disagreements: {bad}" + (f"; fallbacks (no single short arc): {fallbacks}" if as_kernel else ""))
    print(f"chain length: mean {cl.mean():.2f}, median {np.median(cl):.0f}, p90 {np.percentile(cl, 90):.0f}, max {cl.max()}  (C2 rings have 64 edges; a slab row of the current index holds ~7.7 of them)")
    return 1 if bad else 0


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "census"):
    sys.exit(main())


def census():
    """python tools/proto_local_chain.py census: every test sub-cell of the C2 right side — how many there are, how long their
    chains get, what the per-sub-cell table would weigh (8 bytes: first coordinate, edge count, base winding)"""
    polys = synth.star_polygons(1000, 64)
    ro, xy = polys.ring_offsets, polys.xy
    R, S = 512, 8
    x0, y0 = xy.min(0)
    x1, y1 = xy.max(0)
    fw, fh = (x1 - x0) / (R - 3), (y1 - y0) / (R - 3)
    rx0, ry0 = x0 - 1.5 * fw, y0 - 1.5 * fh
    sw, sh = fw / S, fh / S
    pad_x, pad_y = fw / 65536.0 / S, fh / 65536.0 / S
    hist = {}
    n_cells = 0
    for g in range(0, 1000, 10):  # every tenth polygon: the right side is statistically uniform
        ring = [(float(p[0]), float(p[1])) for p in xy[ro[g] : ro[g + 1]]]
        n = len(ring) - 1
        touched = set()
        for i in range(n):
            (ax, ay), (bx, by) = ring[i], ring[i + 1]
            i0, i1 = int((min(ax, bx) - pad_x - rx0) / sw), int((max(ax, bx) + pad_x - rx0) / sw)
            j0, j1 = int((min(ay, by) - pad_y - ry0) / sh), int((max(ay, by) + pad_y - ry0) / sh)
            for si in range(i0, i1 + 1):
                for sj in range(j0, j1 + 1):
                    if (si, sj) in touched:
                        continue
                    xl, xh = rx0 + si * sw - pad_x, rx0 + (si + 1) * sw + pad_x
                    yl, yh = ry0 + sj * sh - pad_y, ry0 + (sj + 1) * sh + pad_y
                    if seg_meets_rect(ax, ay, bx, by, xl, yl, xh, yh):
                        touched.add((si, sj))
        for (si, sj) in touched:
            xl, xh = rx0 + si * sw - pad_x, rx0 + (si + 1) * sw + pad_x
            yl, yh = ry0 + sj * sh - pad_y, ry0 + (sj + 1) * sh + pad_y
            k = len(local_chain(ring, xl, yl, xh, yh))
            hist[k] = hist.get(k, 0) + 1
            n_cells += 1
    total = n_cells * 10
    print(f"test sub-cells of the C2 right side: ~{total} ({total / (R * S) ** 2:.3%} of the {R * S} x {R * S} sub-cell grid) -> 8-byte table: {total * 8 / 1e6:.1f} MB")
    tot = sum(hist.values())
    print("chain length histogram:", {k: f"{v / tot:.1%}" for k, v in sorted(hist.items())})
    print(f"mean {sum(k * v for k, v in hist.items()) / tot:.2f} edges per test sub-cell")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "census":
    census()
