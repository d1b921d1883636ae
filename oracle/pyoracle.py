"""ctypes binding of oracle/libgpk_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else.
The product package (geopolars_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from geopolars_amd._abi import PREDICATES, GeoArrowDesc
from geopolars_amd.geoarrow import GeoArrowArray

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgpk_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gpk_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "libgpk_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        D = C.POINTER(GeoArrowDesc)
        VP = C.c_void_p
        L.gpko_orient2d.restype = C.c_int32
        L.gpko_orient2d.argtypes = [C.c_double] * 6
        L.gpko_orient2d_exact_calls.restype = C.c_int64
        L.gpko_coord_pos_ring.restype = C.c_int32
        L.gpko_coord_pos_ring.argtypes = [C.c_double, C.c_double, VP, C.c_int64]
        L.gpko_coord_pos_geom.restype = C.c_int32
        L.gpko_coord_pos_geom.argtypes = [D, C.c_int64, C.c_double, C.c_double]
        L.gpko_line_intersects_line.restype = C.c_int32
        L.gpko_line_intersects_line.argtypes = [VP, VP, VP, VP]
        L.gpko_predicate_pair.restype = C.c_int32
        L.gpko_predicate_pair.argtypes = [D, C.c_int64, D, C.c_int64, C.c_int32]
        for name in ("gpko_area",):
            getattr(L, name).restype = C.c_int32
        L.gpko_area.argtypes = [D, VP, C.c_int32]
        L.gpko_centroid.restype = C.c_int32
        L.gpko_centroid.argtypes = [D, VP, VP]
        L.gpko_bounds.restype = C.c_int32
        L.gpko_bounds.argtypes = [D, VP]
        L.gpko_euclidean_length.restype = C.c_int32
        L.gpko_euclidean_length.argtypes = [D, VP]
        L.gpko_affine_transform.restype = C.c_int32
        L.gpko_affine_transform.argtypes = [D, C.POINTER(C.c_double), VP]
        L.gpko_convex_hull.restype = C.c_int32
        L.gpko_convex_hull.argtypes = [D, VP, VP]
        L.gpko_geodesic_length.restype = C.c_int32
        L.gpko_geodesic_length.argtypes = [D, C.c_int32, VP]
        L.gpko_simplify.restype = C.c_int32
        L.gpko_simplify.argtypes = [D, C.c_double, VP, VP, C.POINTER(C.c_int64)]
        L.gpko_distance_rowwise.restype = C.c_int32
        L.gpko_distance_rowwise.argtypes = [D, D, VP, VP, C.c_int32]
        L.gpko_predicate_rowwise.restype = C.c_int32
        L.gpko_predicate_rowwise.argtypes = [D, D, VP, C.c_int32, VP, C.c_int32]
        L.gpko_spatial_join.restype = C.c_int32
        L.gpko_spatial_join.argtypes = [D, D, C.c_int32, C.c_int32, C.c_int32, VP, VP, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _ok(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


def orient2d(a, b, c) -> int:
    return int(lib().gpko_orient2d(a[0], a[1], b[0], b[1], c[0], c[1]))


def coord_pos_ring(c, ring_xy: np.ndarray) -> int:
    ring_xy = np.ascontiguousarray(ring_xy, dtype=np.float64).reshape(-1, 2)
    return int(lib().gpko_coord_pos_ring(c[0], c[1], ring_xy.ctypes.data, len(ring_xy)))


def coord_pos_geom(a: GeoArrowArray, g: int, c) -> int:
    d = a.desc()
    return int(lib().gpko_coord_pos_geom(C.byref(d), g, c[0], c[1]))


def line_intersects_line(a0, a1, b0, b1) -> bool:
    arr = [np.array(v, dtype=np.float64) for v in (a0, a1, b0, b1)]
    return bool(lib().gpko_line_intersects_line(*[x.ctypes.data for x in arr]))


def area(a: GeoArrowArray, signed: bool = False) -> np.ndarray:
    out = np.empty(len(a), dtype=np.float64)
    d = a.desc()
    _ok(lib().gpko_area(C.byref(d), out.ctypes.data, int(signed)), "area")
    return out


def centroid(a: GeoArrowArray):
    out = np.empty((len(a), 2), dtype=np.float64)
    valid = np.empty(len(a), dtype=np.uint8)
    d = a.desc()
    _ok(lib().gpko_centroid(C.byref(d), out.ctypes.data, valid.ctypes.data), "centroid")
    return out, valid.astype(bool)


def bounds(a: GeoArrowArray) -> np.ndarray:
    out = np.empty((len(a), 4), dtype=np.float64)
    d = a.desc()
    _ok(lib().gpko_bounds(C.byref(d), out.ctypes.data), "bounds")
    return out


def envelope_query(a: GeoArrowArray, boxes, mode: str = "contained"):
    """rstar 0.11 `RTree::locate_in_envelope` / `locate_in_envelope_intersecting` over the leaves the reference inserts — one per
    geometry, its bounding box (spatial_index.rs:320-334, the reference's own use: :383-393,422-429) — restated by brute force:
    `AABB::contains_envelope` (lower <= lower' and upper' <= upper) and `AABB::intersects` (closed intervals).  Null and empty
    geometries have no leaf.  Returns (pairs (query, index) sorted, counts per query)."""
    b = bounds(a)
    ok = a.is_valid() & ~np.isnan(b[:, 0])
    q = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    pairs, counts = [], np.zeros(len(q), dtype=np.uint32)
    for i, (x0, y0, x1, y1) in enumerate(q):
        x0, x1, y0, y1 = min(x0, x1), max(x0, x1), min(y0, y1), max(y0, y1)  # AABB::from_corners orders the corners (NaN stays NaN-ish: no match)
        if mode == "contained":
            m = ok & (b[:, 0] >= x0) & (b[:, 1] >= y0) & (b[:, 2] <= x1) & (b[:, 3] <= y1)
        else:
            m = ok & (b[:, 0] <= x1) & (b[:, 2] >= x0) & (b[:, 1] <= y1) & (b[:, 3] >= y0)
        idx = np.nonzero(m)[0]
        counts[i] = len(idx)
        pairs.append(np.column_stack([np.full(len(idx), i, dtype=np.uint32), idx.astype(np.uint32)]))
    return (np.concatenate(pairs) if pairs else np.zeros((0, 2), dtype=np.uint32)), counts


def euclidean_length(a: GeoArrowArray) -> np.ndarray:
    out = np.empty(len(a), dtype=np.float64)
    d = a.desc()
    _ok(lib().gpko_euclidean_length(C.byref(d), out.ctypes.data), "length")
    return out


def affine_transform(a: GeoArrowArray, m) -> np.ndarray:
    out = np.empty_like(a.xy)
    d = a.desc()
    mm = (C.c_double * 6)(*[float(v) for v in m])
    _ok(lib().gpko_affine_transform(C.byref(d), mm, out.ctypes.data), "affine")
    return out


def convex_hull(a: GeoArrowArray):
    xy = np.empty((a.n_coords + len(a), 2), dtype=np.float64)
    off = np.empty(len(a) + 1, dtype=np.int32)
    d = a.desc()
    _ok(lib().gpko_convex_hull(C.byref(d), xy.ctypes.data, off.ctypes.data), "convex_hull")
    return xy[: off[-1]], off


def geodesic_length(a: GeoArrowArray, method: str) -> np.ndarray:
    out = np.empty(len(a), dtype=np.float64)
    d = a.desc()
    _ok(lib().gpko_geodesic_length(C.byref(d), {"geodesic": 0, "haversine": 1, "vincenty": 2}[method], out.ctypes.data), "geodesic_length")
    return out


def simplify(a: GeoArrowArray, eps: float):
    """-> (xy (k, 2), innermost offsets) of the simplified array (outer offsets are the input's)"""
    n_seq = a.n_rings if a.ring_offsets is not None else len(a)
    xy = np.empty((max(a.n_coords, 1), 2), dtype=np.float64)
    off = np.zeros(n_seq + 1, dtype=np.int32)
    n_out = C.c_int64(0)
    d = a.desc()
    _ok(lib().gpko_simplify(C.byref(d), float(eps), xy.ctypes.data, off.ctypes.data, C.byref(n_out)), "simplify")
    return xy[: int(n_out.value)].copy(), off


def distance_rowwise(a: GeoArrowArray, b: GeoArrowArray, b_rows=None, n_threads: int = 0) -> np.ndarray:
    n = len(a) if a.geom_type == 0 or b_rows is not None else len(b)
    out = np.empty(n, dtype=np.float64)
    da, db = a.desc(), b.desc()
    rows = None if b_rows is None else np.ascontiguousarray(b_rows, dtype=np.uint32)
    _ok(lib().gpko_distance_rowwise(C.byref(da), C.byref(db), None if rows is None else rows.ctypes.data, out.ctypes.data, n_threads), "distance")
    return out


def predicate_rowwise(a: GeoArrowArray, b: GeoArrowArray, predicate: str, b_rows=None, n_threads: int = 0) -> np.ndarray:
    out = np.empty(len(a), dtype=np.uint8)
    da, db = a.desc(), b.desc()
    rows = None if b_rows is None else np.ascontiguousarray(b_rows, dtype=np.uint32)
    _ok(
        lib().gpko_predicate_rowwise(C.byref(da), C.byref(db), None if rows is None else rows.ctypes.data, PREDICATES[predicate], out.ctypes.data, n_threads),
        "predicate",
    )
    return out.astype(bool)


def spatial_join(left: GeoArrowArray, right: GeoArrowArray, predicate: str = "intersects", mode: int = 1, n_threads: int = 0, capacity: int = -1):
    """-> (pairs (H,2) uint32 sorted by (l, r), counts (n_left,) uint32, threads used).
    With `capacity` >= 0 the join runs ONCE into a buffer of that many pairs (timing runs); otherwise a
    count-only call sizes the buffer first."""
    L = lib()
    dl, dr = left.desc(), right.desc()
    counts = np.empty(len(left), dtype=np.uint32)
    n_pairs = C.c_int64(0)
    used = C.c_int32(0)
    if capacity < 0:
        rc = L.gpko_spatial_join(C.byref(dl), C.byref(dr), PREDICATES[predicate], mode, n_threads, counts.ctypes.data, None, 0, C.byref(n_pairs), C.byref(used))
        _ok(rc, "spatial_join(count)")
        capacity = int(n_pairs.value)
    pairs = np.empty((capacity, 2), dtype=np.uint32)
    rc = L.gpko_spatial_join(C.byref(dl), C.byref(dr), PREDICATES[predicate], mode, n_threads, counts.ctypes.data, pairs.ctypes.data, len(pairs), C.byref(n_pairs), C.byref(used))
    _ok(rc, "spatial_join")
    return pairs[: int(n_pairs.value)], counts, int(used.value)
