"""`GeoSeries` — the host-side mirror of the reference's operator surface.

Same method names, argument meaning and error behaviour as `trait GeoSeries`
(geopolars/geopolars-geo/src/geoseries.rs:10-181) and its Python accessor `GeoRustSeries`
(py-geopolars/python/geopolars/internals/georust/geoseries.py:17-320), plus the north-star
predicates (`contains` / `within` / `intersects` / `bounds`) that exist in the reference only as
dead code (geopolars/src/spatial_index.rs:89-137).  polars is not available here, so a series is a
single-chunk GeoArrow array (`GeoArrowArray`) instead of a `polars.Series` — exactly what
ffi.rs:56 rechunks to before crossing into Rust.

Every operator goes through the C ABI into HIP kernels; nothing here computes geometry on the CPU.
Cheap structural accessors (`geom_type`, `is_empty`, `x`, `y`, `exterior`, `envelope` ring assembly)
only slice offset buffers.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Union

import numpy as np

from . import _abi
from ._abi import (
    GEOM_LINESTRING,
    GEOM_MULTILINESTRING,
    GEOM_MULTIPOINT,
    GEOM_MULTIPOLYGON,
    GEOM_POINT,
    GEOM_POLYGON,
    MEM_HOST,
    PREDICATES,
)
from .geoarrow import DeviceGeoArray, GeoArrowArray

TransformOrigin = Union[str, tuple]


class GeoSeries:
    def __init__(self, array: Optional[GeoArrowArray], name: str = "geometry", device: Optional[DeviceGeoArray] = None):
        self._array = array
        self.name = name  # output column name is always "geometry" (util.rs:22,43,52)
        self._dev: Optional[DeviceGeoArray] = device

    @property
    def array(self) -> GeoArrowArray:
        """Host GeoArrow buffers; a series decoded on the GPU downloads them on first use."""
        if self._array is None:
            self._array = self._dev.download()
        return self._array

    # ---- construction --------------------------------------------------------------------------
    @staticmethod
    def from_wkb(column) -> "GeoSeries":
        return GeoSeries(GeoArrowArray.from_arrow_wkb(column))

    def to_wkb(self, on_device: bool = True) -> tuple[np.ndarray, np.ndarray]:
        """The series as a WKB column (values uint8, offsets int32) — the form geometry-valued results leave the
        reference in (from_geom_vec, util.rs:11-24).  Encoded on the GPU by default; on_device=False uses the host
        encoder (no device needed)."""
        if on_device:
            return self.device().to_wkb()
        return self.array.to_wkb()

    @staticmethod
    def from_wkb_device(values, offsets, validity=None) -> "GeoSeries":
        """WKB column decoded on the GPU: only the raw bytes are uploaded (gpk_geoarray_from_wkb)."""
        return GeoSeries(None, device=DeviceGeoArray.from_wkb(values, offsets, validity))

    @staticmethod
    def from_points(xy) -> "GeoSeries":
        return GeoSeries(GeoArrowArray.from_points(xy))

    def __len__(self) -> int:
        return self._dev.n_geoms if self._array is None else len(self._array)

    def device(self) -> DeviceGeoArray:
        """Upload on first use ("copied once to HBM"); later operators reuse the resident copy."""
        if self._dev is None:
            self._dev = DeviceGeoArray.upload(self.array)
        return self._dev

    # ---- structural accessors (offset arithmetic only) -------------------------------------------
    def geom_type(self) -> np.ndarray:
        """geoseries.rs:60-73: -1 missing, 0 Point, 1 LineString, 3 Polygon, 4.., 6 MultiPolygon."""
        out = np.full(len(self), self.array.geom_type, dtype=np.int8)
        out[~self.array.is_valid()] = -1
        return out

    def is_empty(self) -> np.ndarray:
        a = self.array
        if a.geom_type == GEOM_POINT:
            return np.isnan(a.xy).any(axis=1)
        return np.diff(a.geom_offsets) == 0

    def x(self) -> np.ndarray:
        self._require(GEOM_POINT, "x")
        return self.array.xy[:, 0].copy()

    def y(self) -> np.ndarray:
        self._require(GEOM_POINT, "y")
        return self.array.xy[:, 1].copy()

    def exterior(self) -> "GeoSeries":
        """Outer ring of each polygon as a LineString series (geoseries.rs:43-47)."""
        self._require(GEOM_POLYGON, "exterior")
        a = self.array
        first = a.geom_offsets[:-1]
        has = np.diff(a.geom_offsets) > 0
        starts = np.where(has, a.ring_offsets[np.minimum(first, a.n_rings - 1 if a.n_rings else 0)], 0)
        ends = np.where(has, a.ring_offsets[np.minimum(first + 1, a.n_rings)], 0)
        lens = ends - starts
        off = np.zeros(len(self) + 1, dtype=np.int32)
        off[1:] = np.cumsum(lens)
        idx = np.concatenate([np.arange(s, e) for s, e in zip(starts, ends)]) if len(self) else np.zeros(0, int)
        return GeoSeries(GeoArrowArray(GEOM_LINESTRING, a.xy[idx.astype(np.int64)], geom_offsets=off))

    def is_ring(self) -> np.ndarray:
        """geoseries.rs:75-81: True for features that are closed (first coordinate == last); LineStrings only."""
        self._require(GEOM_LINESTRING, "is_ring")
        a = self.array
        o = a.geom_offsets
        n = np.diff(o)
        first = a.xy[np.minimum(o[:-1], max(a.n_coords - 1, 0))] if a.n_coords else np.zeros((len(self), 2))
        last = a.xy[np.maximum(o[1:] - 1, 0)] if a.n_coords else np.zeros((len(self), 2))
        return (n > 0) & np.all(first == last, axis=1)

    def explode(self) -> "GeoSeries":
        """geoseries.rs:49-50: multi-part geometries -> one row per part (pure offset surgery: the coordinate
        buffer is shared, benches/explode.rs explodes 45,000 two-point MultiPoints this way)."""
        a = self.array
        if a.geom_type == GEOM_MULTIPOINT:
            return GeoSeries(GeoArrowArray.from_points(a.xy))
        if a.geom_type == GEOM_MULTILINESTRING:
            return GeoSeries(GeoArrowArray(GEOM_LINESTRING, a.xy, geom_offsets=a.ring_offsets))
        if a.geom_type == GEOM_MULTIPOLYGON:
            return GeoSeries(GeoArrowArray(GEOM_POLYGON, a.xy, geom_offsets=a.part_offsets, ring_offsets=a.ring_offsets))
        return GeoSeries(a)

    def _require(self, t: int, op: str) -> None:
        if self.array.geom_type != t:
            raise _abi.MismatchedGeometry(
                _abi.GPK_ERR_MISMATCHED_GEOMETRY,
                f"{op}: expected {_abi_name(t)} (found {_abi_name(self.array.geom_type)})",
            )

    # ---- unary operators (HIP) -----------------------------------------------------------------
    def area(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_area(self.device().handle, out.ctypes.data, MEM_HOST, None))
        return out

    def signed_area(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_signed_area(self.device().handle, out.ctypes.data, MEM_HOST, None))
        return out

    def euclidean_length(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_euclidean_length(self.device().handle, out.ctypes.data, MEM_HOST, None))
        return out

    def bounds(self) -> np.ndarray:
        """(n, 4) minx, miny, maxx, maxy — the north-star `bounds`; NaN for empty geometries."""
        out = np.empty((len(self), 4), dtype=np.float64)
        _abi.check(_abi.lib().gpk_bounds(self.device().handle, out.ctypes.data, MEM_HOST, None))
        return out

    def envelope(self) -> "GeoSeries":
        """geoseries.rs:28-33: the bounding rectangle as a geometry.  Points stay points; everything
        else becomes the closed 5-coordinate rectangle polygon (minx miny, maxx miny, maxx maxy,
        minx maxy, minx miny) that geo's `Rect::to_polygon` produces."""
        if self.array.geom_type == GEOM_POINT:
            return GeoSeries(self.array)
        b = self.bounds()
        n = len(self)
        ok = ~np.isnan(b[:, 0])
        xy = np.empty((n, 5, 2), dtype=np.float64)
        xy[:, 0] = b[:, [0, 1]]
        xy[:, 1] = b[:, [2, 1]]
        xy[:, 2] = b[:, [2, 3]]
        xy[:, 3] = b[:, [0, 3]]
        xy[:, 4] = b[:, [0, 1]]
        xy = xy[ok].reshape(-1, 2)
        ring_off = np.arange(0, 5 * int(ok.sum()) + 1, 5, dtype=np.int32)
        geom_off = np.zeros(n + 1, dtype=np.int32)
        geom_off[1:] = np.cumsum(ok)
        return GeoSeries(GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=geom_off, ring_offsets=ring_off))

    def centroid(self) -> "GeoSeries":
        xy = np.empty((len(self), 2), dtype=np.float64)
        valid = np.empty(len(self), dtype=np.uint8)
        _abi.check(_abi.lib().gpk_centroid(self.device().handle, xy.ctypes.data, valid.ctypes.data, MEM_HOST, None))
        return GeoSeries(GeoArrowArray.from_points(xy))

    def convex_hull(self) -> "GeoSeries":
        a = self.array
        xy = np.empty((a.n_coords + len(self), 2), dtype=np.float64)
        ring_off = np.empty(len(self) + 1, dtype=np.int32)
        _abi.check(_abi.lib().gpk_convex_hull(self.device().handle, xy.ctypes.data, ring_off.ctypes.data, MEM_HOST, None))
        xy = xy[: ring_off[-1]]
        return GeoSeries(
            GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=np.arange(len(self) + 1, dtype=np.int32), ring_offsets=ring_off)
        )

    def affine_transform(self, matrix: Sequence[float]) -> "GeoSeries":
        """matrix = [a, b, xoff, d, e, yoff] — the order `AffineTransform::from([f64; 6])` takes and
        py-geopolars/src/geo.rs:10-16 forwards (NOT the [a,b,d,e,xoff,yoff] of the Python docstring,
        georust/geoseries.py:33)."""
        m = (C.c_double * 6)(*[float(v) for v in matrix])
        a = self.array
        out = np.empty_like(a.xy)
        _abi.check(_abi.lib().gpk_affine_transform(self.device().handle, m, out.ctypes.data, MEM_HOST, None))
        return GeoSeries(
            GeoArrowArray(a.geom_type, out, a.geom_offsets, a.part_offsets, a.ring_offsets, a.validity, n_geoms=a.n_geoms)
        )

    def translate(self, xoff: float = 0.0, yoff: float = 0.0) -> "GeoSeries":
        """geoseries.rs:163-174; parameter names of the Python surface (georust/geoseries.py:278)"""
        return self.affine_transform([1.0, 0.0, xoff, 0.0, 1.0, yoff])

    # ---- operators of the reference surface that are off this backend's path -------------------
    def geodesic_length(self, method: str = "geodesic") -> np.ndarray:
        raise NotImplementedError("geodesic_length (geoseries.rs:52-58) is not on the accelerated path: use the reference's CPU implementation")

    def simplify(self, tolerance: float) -> "GeoSeries":
        raise NotImplementedError("simplify (geoseries.rs:108-116) is not on the accelerated path: use the reference's CPU implementation")

    def to_crs(self, from_crs: str, to_crs: str) -> "GeoSeries":
        raise NotImplementedError("to_crs (geoseries.rs:148-151, PROJ) is not on the accelerated path: use the reference's CPU implementation")

    def _origin(self, origin: TransformOrigin) -> np.ndarray:
        """TransformOrigin (py-geopolars/src/utils.rs:5-27): 'centroid' | 'center' (of the bbox) |
        (x, y).  Per-geometry origins -> one matrix per row."""
        if isinstance(origin, str):
            o = origin.lower()
            if o == "centroid":
                return self.centroid().array.xy
            if o == "center":
                b = self.bounds()
                return np.stack([(b[:, 0] + b[:, 2]) / 2.0, (b[:, 1] + b[:, 3]) / 2.0], axis=1)
            raise ValueError("Invalid argument")  # PyGeopolarsError::Other("Invalid argument"), utils.rs:21
        x, y = origin
        return np.tile(np.array([[float(x), float(y)]]), (len(self), 1))

    def _per_row_affine(self, mats: np.ndarray) -> "GeoSeries":
        """One matrix per geometry (per-geometry origins): a single HIP launch, G lanes per geometry."""
        if len(mats) == 0:
            return self
        a = self.array
        mats = np.ascontiguousarray(mats, dtype=np.float64).reshape(len(self), 6)
        out = np.empty_like(a.xy)
        _abi.check(_abi.lib().gpk_affine_transform_rows(self.device().handle, mats.ctypes.data, out.ctypes.data, MEM_HOST, None))
        return GeoSeries(
            GeoArrowArray(a.geom_type, out, a.geom_offsets, a.part_offsets, a.ring_offsets, a.validity, n_geoms=a.n_geoms)
        )

    def rotate(self, angle: float, origin: TransformOrigin = "center") -> "GeoSeries":
        """angle in degrees, counter-clockwise, about `origin` (geoseries.rs:85-93)."""
        t = math.radians(angle)
        c, s = math.cos(t), math.sin(t)
        o = self._origin(origin)
        mats = np.stack(
            [np.full(len(o), c), np.full(len(o), -s), o[:, 0] - c * o[:, 0] + s * o[:, 1], np.full(len(o), s), np.full(len(o), c), o[:, 1] - s * o[:, 0] - c * o[:, 1]],
            axis=1,
        )
        return self._per_row_affine(mats)

    def scale(self, xfact: float = 1.0, yfact: float = 1.0, origin: TransformOrigin = "center") -> "GeoSeries":
        o = self._origin(origin)
        z = np.zeros(len(o))
        mats = np.stack([z + xfact, z, o[:, 0] * (1 - xfact), z, z + yfact, o[:, 1] * (1 - yfact)], axis=1)
        return self._per_row_affine(mats)

    def skew(self, xs: float = 0.0, ys: float = 0.0, origin: TransformOrigin = "center") -> "GeoSeries":
        """geoseries.rs:118-139: [[1, tan(xs), xoff], [tan(ys), 1, yoff]], xoff = -origin.y*tan(xs),
        yoff = -origin.x*tan(ys); angles in degrees."""
        tx, ty = math.tan(math.radians(xs)), math.tan(math.radians(ys))
        o = self._origin(origin)
        z = np.zeros(len(o))
        mats = np.stack([z + 1.0, z + tx, -o[:, 1] * tx, z + ty, z + 1.0, -o[:, 0] * ty], axis=1)
        return self._per_row_affine(mats)

    # ---- binary row-wise operators (HIP) ---------------------------------------------------------
    def distance(self, other: "GeoSeries", other_rows: Optional[np.ndarray] = None, row_map: Optional["RowMap"] = None) -> np.ndarray:
        """geoseries.rs:141-146: 1-to-1 row-wise Euclidean distance.  `other_rows` pairs row i with
        other[other_rows[i]] (the take() a dataframe caller would have materialised); a `row_map` prepared once from
        such a pairing (RowMap(other, other_rows)) skips the per-call ordering of the rows."""
        if row_map is not None:
            out = np.empty(len(self), dtype=np.float64)
            _abi.check(_abi.lib().gpk_distance_rowmap(self.device().handle, other.device().handle, row_map.handle, out.ctypes.data, MEM_HOST, None))
            return out
        n = len(self) if self.array.geom_type == GEOM_POINT or other_rows is not None else len(other)
        out = np.empty(n, dtype=np.float64)
        rows = None
        if other_rows is not None:
            rows = np.ascontiguousarray(other_rows, dtype=np.uint32)
        _abi.check(
            _abi.lib().gpk_distance_rowwise(
                self.device().handle, other.device().handle, None if rows is None else rows.ctypes.data, out.ctypes.data, MEM_HOST, None
            )
        )
        return out

    def _predicate(self, other: "GeoSeries", name: str, other_rows=None) -> np.ndarray:
        out = np.empty(len(self), dtype=np.uint8)
        rows = None if other_rows is None else np.ascontiguousarray(other_rows, dtype=np.uint32)
        _abi.check(
            _abi.lib().gpk_predicate_rowwise(
                self.device().handle, other.device().handle, None if rows is None else rows.ctypes.data, PREDICATES[name], out.ctypes.data, MEM_HOST, None
            )
        )
        return out.astype(bool)

    def contains(self, other: "GeoSeries", other_rows=None) -> np.ndarray:
        return self._predicate(other, "contains", other_rows)

    def within(self, other: "GeoSeries", other_rows=None) -> np.ndarray:
        return self._predicate(other, "within", other_rows)

    def intersects(self, other: "GeoSeries", other_rows=None) -> np.ndarray:
        return self._predicate(other, "intersects", other_rows)


class RowMap:
    """A row pairing (left row i -> right row rows[i]) ordered once for the grouped distance kernel (gpk_rowmap_build):
    reuse it for every batch of points that joins the same linestring column through the same foreign-key column."""

    def __init__(self, right: GeoSeries, rows, stream: int = 0):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        h = C.c_void_p()
        _abi.check(_abi.lib().gpk_rowmap_build(right.device().handle, rows.ctypes.data if len(rows) else None, len(rows), MEM_HOST, stream, C.byref(h)))
        self._h = h

    @staticmethod
    def from_device(right_dev: DeviceGeoArray, rows_tensor, stream: int = 0) -> "RowMap":
        """rows_tensor: int32/uint32 CUDA tensor (the map already in HBM)"""
        self = RowMap.__new__(RowMap)
        h = C.c_void_p()
        _abi.check(_abi.lib().gpk_rowmap_build(right_dev.handle, rows_tensor.data_ptr(), rows_tensor.shape[0], _abi.MEM_DEVICE, stream, C.byref(h)))
        self._h = h
        return self

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def free(self) -> None:
        if self._h:
            _abi.lib().gpk_rowmap_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _abi_name(t: int) -> str:
    from .geoarrow import GEOM_NAMES

    return GEOM_NAMES.get(t, f"type {t}")
