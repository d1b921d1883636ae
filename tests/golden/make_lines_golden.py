"""Generate tests/golden/lines_lattice.npz: 1500 integer-lattice linestrings (0-9 vertices, some closed, some with repeated
vertices) and a lattice point per row; answers from Python integers / fractions / math.fsum — NOT from the oracle or the
library: length, length-weighted centroid, bounds, contains(linestring, point) (geo 0.27: on the line and not one of the two
end points of an open linestring) and point-linestring distance (upstream's degenerate cases kept: an empty linestring is at
distance 0, a one-vertex linestring at f64::MAX unless the point is that vertex).  CPU only.
    python tests/golden/make_lines_golden.py
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
from tests.test_oracle_rational import on_segment  # noqa: E402


def main(out_dir: str = HERE) -> None:
    rng = random.Random(55)
    lines, pts = [], []
    for _ in range(1500):
        n = rng.choice([0, 1, 2, 2, 3, 4, 5, 7, 9])
        l = [(rng.randint(-20, 20), rng.randint(-20, 20)) for _ in range(n)]
        if n >= 3 and rng.random() < 0.25:
            l[-1] = l[0]  # closed
        if n >= 3 and rng.random() < 0.2:
            l[1] = l[0]  # repeated vertex
        lines.append(l)
        k = rng.random()
        if l and k < 0.3:
            pts.append(rng.choice(l))  # a vertex (possibly an end point)
        elif len(l) >= 2 and k < 0.5:
            i = rng.randrange(len(l) - 1)
            (x0, y0), (x1, y1) = l[i], l[i + 1]
            pts.append(((x0 + x1) // 2, (y0 + y1) // 2))  # near (often on) a segment
        else:
            pts.append((rng.randint(-22, 22), rng.randint(-22, 22)))
    length, cen, cen_ok, bounds, contains, dist = [], [], [], [], [], []
    for l, p in zip(lines, pts):
        segs = [(l[k], l[k + 1]) for k in range(len(l) - 1)]
        lens = [math.hypot(e[0] - s[0], e[1] - s[1]) for s, e in segs]
        tot = math.fsum(lens)
        length.append(tot)
        if not l:
            # geo-types private_utils::point_line_string_euclidean_distance: an empty linestring is at distance zero
            cen.append((np.nan, np.nan)); cen_ok.append(False); bounds.append((np.nan,) * 4); contains.append(False); dist.append(0.0)
            continue
        cen_ok.append(True)
        if tot > 0:
            cen.append((math.fsum(w * (s[0] + e[0]) / 2 for w, (s, e) in zip(lens, segs)) / tot, math.fsum(w * (s[1] + e[1]) / 2 for w, (s, e) in zip(lens, segs)) / tot))
        else:
            cen.append((float(F(sum(x for x, _ in l), len(l))), float(F(sum(y for _, y in l), len(l)))))
        bounds.append((min(x for x, _ in l), min(y for _, y in l), max(x for x, _ in l), max(y for _, y in l)))
        closed = l[0] == l[-1]
        if p == l[0] or p == l[-1]:
            contains.append(closed)
        else:
            contains.append(any(on_segment(s, e, p) for s, e in segs))
        if len(l) == 1:  # upstream folds f64::MAX over zero segments unless the point IS the single vertex
            dist.append(0.0 if p == l[0] else sys.float_info.max)
            continue
        best = None
        for s, e in segs:
            ab = (e[0] - s[0], e[1] - s[1])
            ap = (p[0] - s[0], p[1] - s[1])
            d2 = ab[0] ** 2 + ab[1] ** 2
            t = min(max(F(ap[0] * ab[0] + ap[1] * ab[1], d2), F(0)), F(1)) if d2 else F(0)
            q2 = (F(ap[0]) - t * ab[0]) ** 2 + (F(ap[1]) - t * ab[1]) ** 2
            best = q2 if best is None or q2 < best else best
        dist.append(math.sqrt(float(best)))
    a = GeoArrowArray.from_linestrings(lines)
    out = os.path.join(out_dir, "lines_lattice.npz")
    np.savez_compressed(
        out, xy=a.xy, geom_offsets=a.geom_offsets, points=np.array(pts, dtype=np.float64), length=np.array(length), centroid=np.array(cen),
        centroid_valid=np.array(cen_ok), bounds=np.array(bounds, dtype=np.float64), contains=np.array(contains), distance=np.array(dist),
    )
    print(out, "rows", len(lines), "contains", sum(contains), "zero distance", sum(1 for d in dist if d == 0), "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
