"""Generate tests/golden/ops_lattice.npz: 1500 integer-lattice polygons (a third with a hole) and, per row, a lattice point;
answers computed with Python integers / fractions.Fraction by the brute-force restatements of tests/test_oracle_rational.py
and an integer monotone chain — NOT by the oracle or the library:
  area (exact), centroid (rational, rounded once), convex hull (closed, counter-clockwise, from the lexicographic minimum),
  point-in-polygon position (-1 / 0 / 1) and point-polygon distance (rational square, one sqrt).
CPU only.    python tests/golden/make_ops_golden.py
"""
import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fractions import Fraction as F  # noqa: E402

from geopolars_amd.geoarrow import GeoArrowArray  # noqa: E402
from tests.lattice import star_with_hole  # noqa: E402
from tests.test_oracle_rational import _edges, _poly_pos, _ring_moments  # noqa: E402


def int_hull(points):
    """closed counter-clockwise hull of integer points, collinear points dropped, starting at the lexicographic minimum"""
    pts = sorted(set(points))
    if len(pts) == 1:
        return [pts[0], pts[0]]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    h = lower[:-1] + upper[:-1]
    if len(h) < 3:  # all collinear: the two extremes
        return [pts[0], pts[-1], pts[0]]
    return h + [h[0]]


def main(out_dir: str = HERE) -> None:
    rng = random.Random(77)
    polys, pts = [], []
    for _ in range(1500):
        radii = [rng.randint(2, 12) for _ in range(8)]
        hole = [rng.randint(1, 11) for _ in range(8)] if rng.random() < 0.35 else None
        touch = rng.randint(0, 7) if rng.random() < 0.2 else None
        cx, cy = rng.randint(-40, 40), rng.randint(-40, 40)
        polys.append(star_with_hole(cx, cy, radii, hole, touch))
        pts.append((cx + rng.randint(-14, 14), cy + rng.randint(-14, 14)))
    area, cen, pos, dist = [], [], [], []
    hull_xy, hull_off = [], [0]
    for p, q in zip(polys, pts):
        a2 = mx = my = F(0)
        for k, ring in enumerate(p):
            r2, rx, ry = _ring_moments(ring)
            s = (1 if r2 > 0 else -1) * (1 if k == 0 else -1)
            a2 += s * r2
            mx += s * rx
            my += s * ry
        area.append(float(a2 / 2))
        cen.append((float(mx / (3 * a2)), float(my / (3 * a2))))
        h = int_hull([v for ring in p for v in ring])
        hull_xy.extend(h)
        hull_off.append(len(hull_xy))
        k = _poly_pos(p, q)
        pos.append(k)
        if k >= 0:
            dist.append(0.0)
        else:
            best = None
            for ring in p:
                for s, e in _edges(ring):
                    ab = (e[0] - s[0], e[1] - s[1])
                    ap = (q[0] - s[0], q[1] - s[1])
                    t = min(max(F(ap[0] * ab[0] + ap[1] * ab[1], ab[0] ** 2 + ab[1] ** 2), F(0)), F(1))
                    d2 = (F(ap[0]) - t * ab[0]) ** 2 + (F(ap[1]) - t * ab[1]) ** 2
                    best = d2 if best is None or d2 < best else best
            dist.append(math.sqrt(float(best)))
    a = GeoArrowArray.from_polygons(polys)
    out = os.path.join(out_dir, "ops_lattice.npz")
    np.savez_compressed(
        out,
        xy=a.xy, geom_offsets=a.geom_offsets, ring_offsets=a.ring_offsets, points=np.array(pts, dtype=np.float64),
        area=np.array(area), centroid=np.array(cen), position=np.array(pos, dtype=np.int8), distance=np.array(dist),
        hull_xy=np.array(hull_xy, dtype=np.float64), hull_offsets=np.array(hull_off, dtype=np.int32),
    )
    print(out, "rows", len(polys), "inside", pos.count(1), "boundary", pos.count(0), "outside", pos.count(-1), "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
