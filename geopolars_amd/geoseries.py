"""`GeoSeries` — the host-side mirror of the reference's operator surface.

Same method names, argument meaning and error behaviour as `trait GeoSeries`
(geopolars/geopolars-geo/src/geoseries.rs:10-181) and its Python accessor `GeoRustSeries`
(py-geopolars/python/geopolars/internals/georust/geoseries.py:17-320), plus the north-star
predicates (`contains` / `within` / `intersects` / `bounds`) that exist in the reference only as
dead code (geopolars/src/spatial_index.rs:89-137).  polars is not available here, so a series is a
single-chunk GeoArrow array (`GeoArrowArray`) instead of a `polars.Series` — exactly what
ffi.rs:56 rechunks to before crossing into Rust.

Every operator — the structural ones (`geom_type`, `is_empty`, `x`, `y`, `exterior`, `explode`, `envelope`, `is_ring`)
included — goes through the C ABI into HIP kernels; nothing here computes geometry on the CPU.
"""
from __future__ import annotations

import ctypes as C
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
    def from_arrow(column, geom_type: int = -1) -> "GeoSeries":
        """`geopolars.from_arrow` (py-geopolars/src/ffi.rs:93-109 calls it on the way back; the Series it wraps crosses into Rust
        through the Arrow C Data Interface, :12-32): a pyarrow geometry column — WKB binary, or native GeoArrow with Struct<x, y>
        (internals/geoseries.py:86-113) or FixedSizeList<f64, 2> coordinates — handed to the library as the two exported structs
        (gpk_geoarray_from_arrow); the column lands in HBM without a host-side rewrite."""
        return GeoSeries(None, device=DeviceGeoArray.from_arrow(column, geom_type))

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

    # ---- structural operators (gpk_structural.hip: maps over rows / offset surgery on the device) ---------------
    def _row_map(self, fn, dtype) -> np.ndarray:
        out = np.empty(len(self), dtype=dtype)
        _abi.check(fn(self.device().handle, out.ctypes.data if len(out) else None, MEM_HOST, None))
        return out

    def geom_type(self) -> np.ndarray:
        """geoseries.rs:60-73: -1 missing, 0 Point, 1 LineString, 3 Polygon, 4.., 6 MultiPolygon."""
        return self._row_map(_abi.lib().gpk_geom_type, np.int8)

    def is_empty(self) -> np.ndarray:
        """geoseries.rs:75-76 (geo HasDimensions::is_empty)."""
        return self._row_map(_abi.lib().gpk_is_empty, np.uint8).astype(bool)

    def x(self) -> np.ndarray:
        self._require(GEOM_POINT, "x")
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_point_xy(self.device().handle, out.ctypes.data, None, MEM_HOST, None))
        return out

    def y(self) -> np.ndarray:
        self._require(GEOM_POINT, "y")
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_point_xy(self.device().handle, None, out.ctypes.data, MEM_HOST, None))
        return out

    def exterior(self) -> "GeoSeries":
        """Outer ring of each polygon as a LineString series (geoseries.rs:43-47); null rows stay null."""
        self._require(GEOM_POLYGON, "exterior")
        a = self.array
        xy = np.empty((max(a.n_coords, 1), 2), dtype=np.float64)
        off = np.zeros(len(self) + 1, dtype=np.int32)
        n_out = C.c_int64(0)
        _abi.check(_abi.lib().gpk_exterior(self.device().handle, xy.ctypes.data, off.ctypes.data, C.byref(n_out), MEM_HOST, None))
        return GeoSeries(GeoArrowArray(GEOM_LINESTRING, xy[: int(n_out.value)].copy(), geom_offsets=off, validity=a.validity, n_geoms=len(self)))

    def is_ring(self) -> np.ndarray:
        """geoseries.rs:78-83: True for closed features (first coordinate == last; geo-types counts an empty linestring
        as closed); LineStrings only."""
        self._require(GEOM_LINESTRING, "is_ring")
        return self._row_map(_abi.lib().gpk_is_ring, np.uint8).astype(bool)

    def explode(self, return_parents: bool = False):
        """geoseries.rs:49-50: multi-part geometries -> one row per part.  Pure offset surgery on the device: the result
        views this series' coordinate buffer (benches/explode.rs explodes 45,000 two-point MultiPoints this way).  With
        `return_parents` also the row each member came from."""
        h = C.c_void_p()
        dev = self.device()
        sizes = (C.c_int64 * 4)()  # n_coords, n_parts, n_rings, n_geoms of the handle: no host copy of a series decoded on the GPU
        _abi.check(_abi.lib().gpk_geoarray_download(dev.handle, sizes, None, None, None, None, None))
        src_type = dev.geom_type
        n_members = {GEOM_MULTIPOINT: int(sizes[0]), GEOM_MULTILINESTRING: int(sizes[2]), GEOM_MULTIPOLYGON: int(sizes[1])}.get(src_type, len(self))
        parents = np.empty(n_members, dtype=np.int32) if return_parents else None
        _abi.check(_abi.lib().gpk_explode(dev.handle, parents.ctypes.data if return_parents and n_members else None, MEM_HOST, None, C.byref(h)))
        gt = {GEOM_MULTIPOINT: GEOM_POINT, GEOM_MULTILINESTRING: GEOM_LINESTRING, GEOM_MULTIPOLYGON: GEOM_POLYGON}.get(src_type, src_type)
        view = DeviceGeoArray(h.value, gt, n_members, int(sizes[0]), keepalive=[self._dev])  # the view borrows our buffers
        out = GeoSeries(None, device=view)
        return (out, parents) if return_parents else out

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
        minx maxy, minx miny) that geo's `Rect::to_polygon` produces; null and empty rows give null."""
        if self.array.geom_type == GEOM_POINT:
            return GeoSeries(self.array)
        n = len(self)
        xy = np.empty((5 * n, 2), dtype=np.float64)
        valid = np.empty(n, dtype=np.uint8)
        _abi.check(_abi.lib().gpk_envelope(self.device().handle, xy.ctypes.data if n else None, valid.ctypes.data if n else None, MEM_HOST, None))
        validity = None if valid.all() else np.packbits(valid.astype(bool), bitorder="little")
        return GeoSeries(
            GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=np.arange(n + 1, dtype=np.int32), ring_offsets=np.arange(0, 5 * n + 1, 5, dtype=np.int32), validity=validity)
        )

    def centroid(self) -> "GeoSeries":
        xy = np.empty((len(self), 2), dtype=np.float64)
        valid = np.empty(len(self), dtype=np.uint8)
        _abi.check(_abi.lib().gpk_centroid(self.device().handle, xy.ctypes.data, valid.ctypes.data, MEM_HOST, None))
        ok = valid.astype(bool) & self.array.is_valid()  # null in, null out; an empty geometry has no centroid
        return GeoSeries(GeoArrowArray.from_points(xy, validity=None if ok.all() else np.packbits(ok, bitorder="little")))

    def convex_hull(self) -> "GeoSeries":
        a = self.array
        xy = np.empty((a.n_coords + len(self), 2), dtype=np.float64)
        ring_off = np.empty(len(self) + 1, dtype=np.int32)
        _abi.check(_abi.lib().gpk_convex_hull(self.device().handle, xy.ctypes.data, ring_off.ctypes.data, MEM_HOST, None))
        xy = xy[: ring_off[-1]]
        return GeoSeries(
            GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=np.arange(len(self) + 1, dtype=np.int32), ring_offsets=ring_off, validity=a.validity)
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

    GEODESIC_METHODS = {"geodesic": 0, "haversine": 1, "vincenty": 2}

    def geodesic_length(self, method: str = "geodesic") -> np.ndarray:
        """geoseries.rs:52-58 / georust/geoseries.py: metres, coordinates in (lon, lat) degrees; all three methods run on the
        GPU — 'geodesic' (the default) is Karney's algorithm (csrc/gpk_karney.h), as in geo's GeodesicLength."""
        m = self.GEODESIC_METHODS.get(method.lower())
        if m is None:
            raise ValueError("Geodesic calculation method not valid. Use one of geodesic, haversine or vincenty")  # geo.rs:68-71
        out = np.empty(len(self), dtype=np.float64)
        _abi.check(_abi.lib().gpk_geodesic_length(self.device().handle, m, out.ctypes.data if len(out) else None, MEM_HOST, None))
        return out

    def simplify(self, tolerance: float) -> "GeoSeries":
        """geoseries.rs:108-116: Douglas-Peucker with geo 0.27's rules (gpk_simplify); the nesting above the coordinate
        sequences is unchanged."""
        a = self.array
        if a.geom_type in (GEOM_POINT, GEOM_MULTIPOINT):
            return GeoSeries(a)
        n_seq = a.n_rings if a.ring_offsets is not None else len(self)
        xy = np.empty((max(a.n_coords, 1), 2), dtype=np.float64)
        off = np.zeros(n_seq + 1, dtype=np.int32)
        n_out = C.c_int64(0)
        _abi.check(_abi.lib().gpk_simplify(self.device().handle, float(tolerance), xy.ctypes.data, off.ctypes.data, C.byref(n_out), MEM_HOST, None))
        xy = xy[: int(n_out.value)].copy()
        if a.ring_offsets is not None:
            return GeoSeries(GeoArrowArray(a.geom_type, xy, a.geom_offsets, a.part_offsets, off, a.validity, n_geoms=a.n_geoms))
        return GeoSeries(GeoArrowArray(a.geom_type, xy, off, validity=a.validity, n_geoms=a.n_geoms))

    # ---- the one operator of the reference surface that stays off this backend ---------------------
    def to_crs(self, from_crs: str, to_crs: str) -> "GeoSeries":
        raise NotImplementedError("to_crs (geoseries.rs:148-151, PROJ) is not on the accelerated path: use the reference's CPU implementation")

    def _about_origin(self, kind: int, p0: float, p1: float, origin: TransformOrigin) -> "GeoSeries":
        """rotate / scale / skew: the per-geometry origins (TransformOrigin, py-geopolars/src/utils.rs:5-27: 'centroid' |
        'center' of the bbox | (x, y)) and matrices are computed on the device (gpk_affine_about_origin)."""
        ox = oy = 0.0
        if isinstance(origin, str):
            o = origin.lower()
            if o not in ("centroid", "center"):
                raise ValueError("Invalid argument")  # PyGeopolarsError::Other("Invalid argument"), utils.rs:21
            ok = 0 if o == "centroid" else 1
        else:
            ok = 2
            ox, oy = (float(v) for v in origin)
        a = self.array
        out = np.empty_like(a.xy)
        _abi.check(_abi.lib().gpk_affine_about_origin(self.device().handle, kind, float(p0), float(p1), ok, ox, oy, out.ctypes.data if len(out) else None, MEM_HOST, None))
        return GeoSeries(GeoArrowArray(a.geom_type, out, a.geom_offsets, a.part_offsets, a.ring_offsets, a.validity, n_geoms=a.n_geoms))

    def rotate(self, angle: float, origin: TransformOrigin = "center") -> "GeoSeries":
        """angle in degrees, counter-clockwise, about `origin` (geoseries.rs:85-93)."""
        return self._about_origin(0, angle, 0.0, origin)

    def scale(self, xfact: float = 1.0, yfact: float = 1.0, origin: TransformOrigin = "center") -> "GeoSeries":
        return self._about_origin(1, xfact, yfact, origin)

    def skew(self, xs: float = 0.0, ys: float = 0.0, origin: TransformOrigin = "center") -> "GeoSeries":
        """geoseries.rs:118-139: [[1, tan(xs), xoff], [tan(ys), 1, yoff]], xoff = -origin.y*tan(xs),
        yoff = -origin.x*tan(ys); angles in degrees."""
        return self._about_origin(2, xs, ys, origin)

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
