"""The oracle against a THIRD-PARTY implementation where one is importable here: scipy.spatial.ConvexHull (Qhull — quickhull, the
algorithm family of geo 0.27's `convex_hull/qhull.rs` that `oracle/gpk_oracle.c` restates).  Random clouds in general position: the
hull is unique, so the vertex SETS must agree exactly and the oracle's ring must be closed and counter-clockwise.  CPU only."""
import numpy as np
import pytest

from geopolars_amd.geoarrow import GeoArrowArray

scipy_spatial = pytest.importorskip("scipy.spatial")


def _shoelace2(ring):
    x, y = ring[:, 0], ring[:, 1]
    return float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))


@pytest.mark.parametrize("seed, n_rows", [(1, 200), (2, 60)])
def test_hull_vertices_equal_qhull(oracle, seed, n_rows):
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n_rows):
        n = int(rng.integers(3, 200))
        kind = rng.integers(0, 3)
        if kind == 0:
            pts = rng.uniform(-1000.0, 1000.0, (n, 2))
        elif kind == 1:  # points near a circle: almost every point is a hull vertex
            t = rng.uniform(0.0, 2.0 * np.pi, n)
            pts = np.stack([np.cos(t), np.sin(t)], axis=1) * rng.uniform(50.0, 60.0, (n, 1)) + rng.uniform(-5.0, 5.0, 2)
        else:  # a cloud with heavy interior: few hull vertices
            pts = rng.normal(0.0, 1.0, (n, 2)) * rng.uniform(0.1, 100.0)
        rows.append(pts)
    a = GeoArrowArray.from_linestrings([p.tolist() for p in rows])  # (convex_hull only looks at the coordinates of a row)
    xy, off = oracle.convex_hull(a)
    for k, pts in enumerate(rows):
        ring = xy[off[k] : off[k + 1]]
        assert len(ring) >= 4 and np.array_equal(ring[0], ring[-1])  # closed
        assert _shoelace2(ring) > 0.0  # counter-clockwise
        want = {tuple(p) for p in pts[scipy_spatial.ConvexHull(pts).vertices]}
        got = {tuple(p) for p in ring[:-1]}
        assert got == want, (k, len(got), len(want))


def test_area_and_perimeter_of_convex_polygons_equal_qhull(oracle):
    """Qhull reports a 2-D hull's area (`volume`) and perimeter (`area`) from its own facet arithmetic: oracle.area and
    oracle.euclidean_length of the same convex polygons agree to 1e-9 relative (the north star's tolerance)."""
    rng = np.random.default_rng(7)
    polys, want_area, want_len = [], [], []
    for _ in range(300):
        n = int(rng.integers(3, 120))
        pts = rng.uniform(-500.0, 500.0, (n, 2)) * rng.uniform(0.01, 10.0)
        h = scipy_spatial.ConvexHull(pts)
        ring = pts[h.vertices]  # counter-clockwise in 2-D
        polys.append([ring.tolist()])
        want_area.append(h.volume)
        want_len.append(h.area)
    a = GeoArrowArray.from_polygons(polys)  # (closes the rings)
    got_area, got_len = oracle.area(a), oracle.euclidean_length(a)
    assert np.all(np.abs(got_area - np.array(want_area)) <= 1e-9 * np.array(want_area))
    assert np.all(np.abs(got_len - np.array(want_len)) <= 1e-9 * np.array(want_len))


def test_point_in_convex_polygon_equals_qhull_delaunay(oracle):
    """`within(point, polygon)` for convex polygons against scipy.spatial.Delaunay.find_simplex (Qhull's triangulation of the same
    vertices): equal for every point that is not within 1e-7 of the boundary (Delaunay's own test is tolerance-based there)."""
    rng = np.random.default_rng(11)
    polys, pts, rows, want = [], [], [], []
    for k in range(120):
        cloud = rng.uniform(-100.0, 100.0, (int(rng.integers(3, 40)), 2))
        h = scipy_spatial.ConvexHull(cloud)
        ring = cloud[h.vertices]
        polys.append([ring.tolist()])
        tri = scipy_spatial.Delaunay(ring)
        q = rng.uniform(-120.0, 120.0, (150, 2))
        inside = tri.find_simplex(q) >= 0
        # distance to the boundary (plain numpy): points too close to it are left out of the comparison
        a, b = ring, np.roll(ring, -1, axis=0)
        ab = b - a
        t = np.clip(np.einsum("qed,ed->qe", q[:, None, :] - a[None, :, :], ab) / np.einsum("ed,ed->e", ab, ab), 0.0, 1.0)
        d = np.min(np.linalg.norm(q[:, None, :] - (a[None, :, :] + t[:, :, None] * ab[None, :, :]), axis=2), axis=1)
        keep = d > 1e-7
        pts.append(q[keep])
        rows.append(np.full(int(keep.sum()), k, dtype=np.uint32))
        want.append(inside[keep])
    pa = GeoArrowArray.from_points(np.concatenate(pts))
    qa = GeoArrowArray.from_polygons(polys)
    got = oracle.predicate_rowwise(pa, qa, "within", b_rows=np.concatenate(rows))
    assert np.array_equal(got, np.concatenate(want))
    assert got.sum() > 1000 and (~got).sum() > 1000


def test_haversine_length_equals_sklearn(oracle):
    """geodesic_length(method="haversine") of two-point linestrings against sklearn.metrics.pairwise.haversine_distances (great-circle
    angle) times geo's mean earth radius 6371008.8 m (HaversineLength, geo 0.27): 1e-9 relative away from the antipode / zero ends."""
    pairwise = pytest.importorskip("sklearn.metrics.pairwise")
    rng = np.random.default_rng(5)
    lon = rng.uniform(-180.0, 180.0, (4000, 2))
    lat = rng.uniform(-89.0, 89.0, (4000, 2))
    lines = [[[lon[i, 0], lat[i, 0]], [lon[i, 1], lat[i, 1]]] for i in range(4000)]
    got = oracle.geodesic_length(GeoArrowArray.from_linestrings(lines), "haversine")
    a = np.radians(np.stack([lat[:, 0], lon[:, 0]], axis=1))
    b = np.radians(np.stack([lat[:, 1], lon[:, 1]], axis=1))
    ang = np.array([pairwise.haversine_distances(a[i : i + 1], b[i : i + 1])[0, 0] for i in range(4000)])
    want = ang * 6371008.8
    ok = (ang > 1e-3) & (ang < np.pi - 1e-3)  # (the formula's conditioning at the ends is the implementations' own business)
    assert ok.sum() > 3900
    assert np.all(np.abs(got[ok] - want[ok]) <= 1e-9 * want[ok])


def _near_boundary(q, ring, eps):
    a, b = ring[:-1], ring[1:]
    ab = b - a
    den = np.einsum("ed,ed->e", ab, ab)
    den = np.where(den > 0.0, den, 1.0)
    t = np.clip(np.einsum("qed,ed->qe", q[:, None, :] - a[None, :, :], ab) / den, 0.0, 1.0)
    d = np.min(np.linalg.norm(q[:, None, :] - (a[None, :, :] + t[:, :, None] * ab[None, :, :]), axis=2), axis=1)
    return d <= eps


def test_point_in_polygon_equals_matplotlib_on_the_headline_shapes(oracle):
    """The C2 right side (64-vertex star polygons, non-convex) and multipolygons with holes against matplotlib.path.Path
    (Anti-Grain's crossing test per ring; a polygon = inside its exterior and inside none of its holes): the oracle's join counts
    equal matplotlib's on every point that is not within 1e-6 of a ring."""
    mpath = pytest.importorskip("matplotlib.path")
    from geopolars_amd import synth

    def check(right, pts):
        _, counts, _ = oracle.spatial_join(pts, right, "intersects", mode=1)
        q = pts.xy
        want = np.zeros(len(q), dtype=np.int64)
        near = np.zeros(len(q), dtype=bool)
        go, po, ro = right.geom_offsets, right.part_offsets, right.ring_offsets
        for g in range(len(right)):
            p0, p1 = (int(go[g]), int(go[g + 1])) if po is not None else (g, g + 1)
            in_geom = np.zeros(len(q), dtype=bool)
            for p in range(p0, p1):
                r0, r1 = (int(po[p]), int(po[p + 1])) if po is not None else (int(go[p]), int(go[p + 1]))
                rings = [right.xy[int(ro[r]) : int(ro[r + 1])] for r in range(r0, r1)]
                lo, hi = rings[0].min(axis=0), rings[0].max(axis=0)
                sel = np.nonzero(np.all((q >= lo - 1e-3) & (q <= hi + 1e-3), axis=1))[0]
                if len(sel) == 0:
                    continue
                def ring_test(r):  # (one Path per ring: a compound path's contains_points does not treat inner subpaths as holes)
                    return mpath.Path(r, [mpath.Path.MOVETO] + [mpath.Path.LINETO] * (len(r) - 2) + [mpath.Path.CLOSEPOLY]).contains_points(q[sel])

                inside = ring_test(rings[0])
                for hole in rings[1:]:
                    inside &= ~ring_test(hole)
                in_geom[sel] |= inside
                for r in rings:
                    near[sel] |= _near_boundary(q[sel], r, 1e-6)
            want += in_geom
        ok = ~near
        assert ok.sum() > 0.98 * len(q)
        assert np.array_equal(counts[ok].astype(np.int64), want[ok])
        return int(want[ok].sum())

    hits = check(synth.star_polygons(1000, 64), synth.uniform_points(30_000, seed=3))
    assert hits > 8_000
    mp = synth.powerlaw_multipolygons(400, seed=9)
    hits = check(mp, synth.uniform_points(30_000, seed=4))
    assert hits > 500


def test_area_centroid_distance_equal_sympy_geometry(oracle):
    """sympy.geometry (exact rational arithmetic, an independent code base): area and centroid of random simple polygons on an
    integer lattice, and the point-to-linestring distance, agree with the oracle to 1e-12."""
    sg = pytest.importorskip("sympy.geometry")
    import sympy

    rng = np.random.default_rng(21)
    polys, areas, cents = [], [], []
    for _ in range(40):
        n = int(rng.integers(3, 12))
        ang = np.sort(rng.uniform(0.0, 2.0 * np.pi, n))
        rad = rng.integers(5, 60, n)
        v = np.unique(np.stack([np.rint(rad * np.cos(ang)), np.rint(rad * np.sin(ang))], axis=1).astype(np.int64), axis=0)
        if len(v) < 3:
            continue
        v = v[np.argsort(np.arctan2(v[:, 1], v[:, 0]))]  # star-shaped about the origin: simple
        sp = sg.Polygon(*[sg.Point(int(x), int(y)) for x, y in v])
        if not isinstance(sp, sg.Polygon) or sp.area == 0:
            continue
        polys.append([v.astype(np.float64).tolist()])
        areas.append(float(abs(sp.area)))
        cents.append((float(sp.centroid.x), float(sp.centroid.y)))
    a = GeoArrowArray.from_polygons(polys)
    assert len(polys) >= 30
    assert np.allclose(oracle.area(a), np.array(areas), rtol=1e-12, atol=0.0)
    c = oracle.centroid(a)
    cxy = c[0] if isinstance(c, tuple) else c
    assert np.allclose(np.asarray(cxy).reshape(-1, 2), np.array(cents), rtol=1e-12, atol=1e-12)
    # point -> linestring: min over sympy Segment.distance
    lines, pts, want = [], [], []
    for _ in range(40):
        m = int(rng.integers(2, 7))
        lv = rng.integers(-50, 50, (m, 2))
        if np.any(np.all(lv[1:] == lv[:-1], axis=1)):
            continue
        p = rng.integers(-60, 60, 2)
        segs = [sg.Segment(sg.Point(int(lv[k, 0]), int(lv[k, 1])), sg.Point(int(lv[k + 1, 0]), int(lv[k + 1, 1]))) for k in range(m - 1)]
        d = min(float(sympy.N(s.distance(sg.Point(int(p[0]), int(p[1]))), 30)) for s in segs)
        lines.append(lv.astype(np.float64).tolist())
        pts.append(p.astype(np.float64))
        want.append(d)
    got = oracle.distance_rowwise(GeoArrowArray.from_points(np.array(pts)), GeoArrowArray.from_linestrings(lines))
    assert np.allclose(got, np.array(want), rtol=1e-12, atol=1e-12)


def test_polygon_intersects_polygon_equals_sympy_geometry(oracle):
    """`intersects(polygon, polygon)` on small lattice polygons (shared vertices and touching edges happen by construction) against
    sympy.geometry: the boundaries meet (exact `Polygon.intersection`) or one polygon strictly encloses a vertex of the other."""
    sg = pytest.importorskip("sympy.geometry")
    rng = np.random.default_rng(31)

    def lattice_polygon(cx, cy):
        n = int(rng.integers(3, 8))
        ang = np.sort(rng.uniform(0.0, 2.0 * np.pi, n))
        rad = rng.integers(2, 9, n)
        v = np.unique(np.stack([cx + np.rint(rad * np.cos(ang)), cy + np.rint(rad * np.sin(ang))], axis=1).astype(np.int64), axis=0)
        if len(v) < 3:
            return None
        v = v[np.argsort(np.arctan2(v[:, 1] - cy, v[:, 0] - cx))]
        sp = sg.Polygon(*[sg.Point(int(x), int(y)) for x, y in v])
        return (v, sp) if isinstance(sp, sg.Polygon) and sp.area != 0 else None

    left, right, want = [], [], []
    while len(want) < 60:
        a = lattice_polygon(0, 0)
        b = lattice_polygon(int(rng.integers(-12, 13)), int(rng.integers(-12, 13)))
        if a is None or b is None:
            continue
        (va, pa), (vb, pb) = a, b
        hit = bool(pa.intersection(pb)) or pa.encloses_point(pb.vertices[0]) or pb.encloses_point(pa.vertices[0])
        left.append([va.astype(np.float64).tolist()])
        right.append([vb.astype(np.float64).tolist()])
        want.append(bool(hit))
    got = oracle.predicate_rowwise(GeoArrowArray.from_polygons(left), GeoArrowArray.from_polygons(right), "intersects")
    assert np.array_equal(got, np.array(want))
    assert 10 < sum(want) < 55
