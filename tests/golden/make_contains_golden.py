"""Generate tests/golden/contains_lattice.npz: 4000 polygon pairs (integer lattice stars with and without holes, concentric
and neighbouring, a third of them scaled by 0.1 and nudged by a few ulps) and the answers of contains(a, b) and intersects(a, b) computed by the
rational brute force of tests/test_oracle_rational.py (edges cut at their exact intersection points, the mid point of every
piece located with Fraction arithmetic) — NOT by the oracle or the library.  CPU only, ~1 minute.

    python tests/golden/make_contains_golden.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fractions import Fraction as F  # noqa: E402

from geopolars_amd.geoarrow import GeoArrowArray  # noqa: E402
from tests.lattice import concentric_pair, nudged, random_pair  # noqa: E402
from tests.test_oracle_rational import contains_bruteforce, intersects_bruteforce  # noqa: E402


def main(out_dir: str = HERE) -> None:
    rng = random.Random(20241008)
    pairs = []
    for _ in range(4000):
        pa, pb = concentric_pair(rng) if rng.random() < 0.5 else random_pair(rng)
        if rng.random() < 0.33:
            pa, pb = nudged(pa, rng), nudged(pb, rng)
        pairs.append((pa, pb))
    exact = lambda poly: [[(F(x), F(y)) for x, y in ring] for ring in poly]
    expected = np.array([contains_bruteforce(exact(p), exact(q)) for p, q in pairs], dtype=np.uint8)
    meets = np.array([intersects_bruteforce(exact(p), exact(q)) for p, q in pairs], dtype=np.uint8)
    a = GeoArrowArray.from_polygons([p for p, _ in pairs])
    b = GeoArrowArray.from_polygons([q for _, q in pairs])
    out = os.path.join(out_dir, "contains_lattice.npz")
    np.savez_compressed(
        out,
        a_xy=a.xy, a_geom_offsets=a.geom_offsets, a_ring_offsets=a.ring_offsets,
        b_xy=b.xy, b_geom_offsets=b.geom_offsets, b_ring_offsets=b.ring_offsets,
        contains=expected,
        intersects=meets,
    )
    print(out, "pairs", len(pairs), "contains", int(expected.sum()), "intersects", int(meets.sum()), "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
