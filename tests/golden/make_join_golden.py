"""Generate tests/golden/join_lattice.npz: 150 overlapping integer-lattice polygons (some with holes), 6000 points on the
integer and half-integer lattice (vertices, edge points, interior, exterior) and the sorted (point, polygon) pairs of the
point-in-polygon join — boundary excluded, KA-1 — decided by an even-odd ring walk in Python integers
(tests/test_oracle_rational.py: in_or_on), NOT by the oracle or the library.  CPU only.
    python tests/golden/make_join_golden.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fractions import Fraction as F  # noqa: E402

from geopolars_amd.geoarrow import GeoArrowArray  # noqa: E402
from tests.lattice import star_with_hole  # noqa: E402
from tests.test_oracle_rational import _poly_pos  # noqa: E402


def main(out_dir: str = HERE) -> None:
    rng = random.Random(404)
    polys = []
    for _ in range(150):
        radii = [rng.randint(2, 12) for _ in range(8)]
        hole = [rng.randint(1, 11) for _ in range(8)] if rng.random() < 0.4 else None
        touch = rng.randint(0, 7) if rng.random() < 0.2 else None
        polys.append(star_with_hole(rng.randint(-30, 30), rng.randint(-30, 30), radii, hole, touch))
    pts = [(F(rng.randint(-90, 90), 2), F(rng.randint(-90, 90), 2)) for _ in range(6000)]
    boxes = [(min(x for x, _ in p[0]), min(y for _, y in p[0]), max(x for x, _ in p[0]), max(y for _, y in p[0])) for p in polys]
    pairs, on_boundary = [], 0
    for i, q in enumerate(pts):
        for j, (p, b) in enumerate(zip(polys, boxes)):
            if q[0] < b[0] or q[0] > b[2] or q[1] < b[1] or q[1] > b[3]:
                continue
            k = _poly_pos(p, q)
            on_boundary += k == 0
            if k > 0:
                pairs.append((i, j))
    a = GeoArrowArray.from_polygons(polys)
    out = os.path.join(out_dir, "join_lattice.npz")
    np.savez_compressed(
        out, xy=a.xy, geom_offsets=a.geom_offsets, ring_offsets=a.ring_offsets,
        points=np.array([(float(x), float(y)) for x, y in pts]), pairs=np.array(pairs, dtype=np.uint32),
    )
    print(out, "pairs", len(pairs), "boundary incidences", on_boundary, "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
