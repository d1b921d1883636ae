"""Integer-lattice polygons shared by the rational pins of the oracle (test_oracle_rational.py) and the GPU parity tests
of contains(polygon, polygon) (test_gpu_contains.py): small coordinates make touching, collinear and coincident
boundaries the common case instead of the exception."""

DIRS = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]


def star(cx, cy, radii):
    """simple polygon with integer vertices: one vertex per direction of a fixed fan of 8 integer directions"""
    return [(cx + r * dx, cy + r * dy) for r, (dx, dy) in zip(radii, DIRS)]


def star_with_hole(cx, cy, radii, hole_radii, touch):
    """outer star plus, optionally, a hole on the same fan (radius strictly smaller in every direction, equal in at most
    one direction when `touch`: a hole may touch the exterior in one point)"""
    outer = star(cx, cy, radii)
    if hole_radii is None:
        return [outer]
    hr = [max(1, min(h, r - 1)) for h, r in zip(hole_radii, radii)]
    if any(r < 2 for r in radii):
        return [outer]
    if touch is not None:
        hr[touch] = radii[touch]
    return [outer, star(cx, cy, hr)]


def random_pair(rng):
    """two stars (some with a hole) a few lattice steps apart; the second is usually the smaller one"""

    def one(small):
        hi = 4 if small else 10
        radii = [rng.randint(1, hi) for _ in range(8)]
        hole = [rng.randint(1, 9) for _ in range(8)] if rng.random() < 0.5 else None
        touch = rng.randint(0, 7) if rng.random() < 0.3 else None
        return star_with_hole(rng.randint(-3, 3), rng.randint(-3, 3), radii, hole, touch)

    return one(False), one(rng.random() < 0.7)


def concentric_pair(rng):
    """A = star with a hole; B = star on the same fan squeezed between A's hole and A's exterior, often touching or
    coinciding with either, with an own hole that may or may not cover A's hole: the cases rule (2) exists for."""
    ro = [rng.randint(4, 10) for _ in range(8)]
    rh = [rng.randint(1, r - 1) for r in ro]
    pa = [star(0, 0, ro)] + ([star(0, 0, rh)] if rng.random() < 0.8 else [])
    bo = [rng.randint(max(1, h - 1), r) if rng.random() < 0.8 else r for h, r in zip(rh, ro)]
    if rng.random() < 0.2:
        bo = list(ro)
    if rng.random() < 0.1:
        bo = list(rh)
    bh = [max(1, min(rng.randint(h - 1, h + 1), o - 1)) if rng.random() < 0.7 else min(h, o) for h, o in zip(rh, bo)]
    if rng.random() < 0.3:
        bh = [min(h, o) for h, o in zip(rh, bo)]
    pb = [star(0, 0, bo)]
    if rng.random() < 0.7 and all(o >= 2 for o in bo) and all(0 < h <= o for h, o in zip(bh, bo)) and sum(h == o for h, o in zip(bh, bo)) <= 1:
        pb.append(star(0, 0, bh))
    return pa, pb


def nudged(poly, rng, prob=0.1):
    """the polygon scaled by 0.1 (coordinates no longer exactly representable) with a few coordinates moved by one or two
    units of 2^-50: exact touches become hair-thin gaps or overlaps that only exact orientations resolve"""
    out = []
    for ring in poly:
        r = []
        for x, y in ring:
            fx, fy = x * 0.1, y * 0.1
            if rng.random() < prob:
                fx += rng.choice((-2, -1, 1, 2)) * 2.0**-50
            if rng.random() < prob:
                fy += rng.choice((-2, -1, 1, 2)) * 2.0**-50
            r.append((fx, fy))
        out.append(r)
    return out


def load_contains_golden(key="contains"):
    """tests/golden/contains_lattice.npz -> (a, b, expected): 4000 polygon pairs and contains(a, b) / intersects(a, b) from
    the rational brute force (tests/golden/make_contains_golden.py)"""
    import os

    import numpy as np

    from geopolars_amd import _abi
    from geopolars_amd.geoarrow import GeoArrowArray

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contains_lattice.npz"))
    a = GeoArrowArray(_abi.GEOM_POLYGON, z["a_xy"], geom_offsets=z["a_geom_offsets"], ring_offsets=z["a_ring_offsets"])
    b = GeoArrowArray(_abi.GEOM_POLYGON, z["b_xy"], geom_offsets=z["b_geom_offsets"], ring_offsets=z["b_ring_offsets"])
    return a, b, z[key].astype(bool)
