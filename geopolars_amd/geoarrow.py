"""Host-side GeoArrow containers (numpy buffers) and their device-resident handles.

Layout follows SURVEY.md Appendix A.7 / the reference's ragged-array code
(py-geopolars/python/geopolars/internals/geoseries.py:82-107,164-216): interleaved xy coordinates
plus Arrow `List` i32 offsets per nesting level.  Fixtures arrive as WKB `binary` columns
(data/cities.arrow, datasets/*.arrow); `from_wkb` decodes them once through the library's host
decoder (gpk_wkb_decode), replacing the per-row decode of util.rs:27-37.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._abi import (
    GEOM_LINESTRING,
    GEOM_MULTILINESTRING,
    GEOM_MULTIPOINT,
    GEOM_MULTIPOLYGON,
    GEOM_POINT,
    GEOM_POLYGON,
    GeoArrowDesc,
)

GEOM_NAMES = {
    GEOM_POINT: "Point",
    GEOM_LINESTRING: "LineString",
    GEOM_POLYGON: "Polygon",
    GEOM_MULTIPOINT: "MultiPoint",
    GEOM_MULTILINESTRING: "MultiLineString",
    GEOM_MULTIPOLYGON: "MultiPolygon",
}


def _i32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


@dataclass
class GeoArrowArray:
    """One single-chunk GeoArrow array on the host."""

    geom_type: int
    xy: np.ndarray  # (n_coords, 2) float64, C-contiguous
    geom_offsets: Optional[np.ndarray] = None
    part_offsets: Optional[np.ndarray] = None
    ring_offsets: Optional[np.ndarray] = None
    validity: Optional[np.ndarray] = None  # Arrow bitmap (uint8) or None
    n_geoms: int = field(default=-1)

    def __post_init__(self):
        self.xy = np.ascontiguousarray(self.xy, dtype=np.float64).reshape(-1, 2)
        self.geom_offsets = _i32(self.geom_offsets)
        self.part_offsets = _i32(self.part_offsets)
        self.ring_offsets = _i32(self.ring_offsets)
        if self.validity is not None:
            self.validity = np.ascontiguousarray(self.validity, dtype=np.uint8)
        if self.n_geoms < 0:
            self.n_geoms = len(self.xy) if self.geom_type == GEOM_POINT else len(self.geom_offsets) - 1

    # ---- sizes -------------------------------------------------------------------------------
    @property
    def n_coords(self) -> int:
        return len(self.xy)

    @property
    def n_parts(self) -> int:
        return 0 if self.part_offsets is None else len(self.part_offsets) - 1

    @property
    def n_rings(self) -> int:
        return 0 if self.ring_offsets is None else len(self.ring_offsets) - 1

    def __len__(self) -> int:
        return self.n_geoms

    def nbytes(self) -> int:
        n = self.xy.nbytes
        for a in (self.geom_offsets, self.part_offsets, self.ring_offsets, self.validity):
            if a is not None:
                n += a.nbytes
        return n

    def is_valid(self) -> np.ndarray:
        if self.validity is None:
            return np.ones(self.n_geoms, dtype=bool)
        return np.unpackbits(self.validity, bitorder="little")[: self.n_geoms].astype(bool)

    # ---- C ABI ---------------------------------------------------------------------------------
    def desc(self) -> GeoArrowDesc:
        """Host-space descriptor.  The numpy buffers stay owned by `self` (borrowed for the call)."""
        d = GeoArrowDesc()
        d.geom_type = self.geom_type
        d.mem_space = _abi.MEM_HOST
        d.n_geoms = self.n_geoms
        d.n_coords = self.n_coords
        d.xy = _ptr(self.xy)
        d.geom_offsets = _ptr(self.geom_offsets)
        d.part_offsets = _ptr(self.part_offsets)
        d.ring_offsets = _ptr(self.ring_offsets)
        d.n_parts = self.n_parts
        d.n_rings = self.n_rings
        d.validity = _ptr(self.validity)
        return d

    # ---- constructors ----------------------------------------------------------------------
    @staticmethod
    def from_points(xy, validity=None) -> "GeoArrowArray":
        return GeoArrowArray(GEOM_POINT, np.asarray(xy, dtype=np.float64).reshape(-1, 2), validity=validity)

    @staticmethod
    def from_linestrings(lines: Sequence[Sequence[Sequence[float]]]) -> "GeoArrowArray":
        off = np.zeros(len(lines) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(l) for l in lines])
        xy = np.array([c for l in lines for c in l], dtype=np.float64).reshape(-1, 2)
        return GeoArrowArray(GEOM_LINESTRING, xy, geom_offsets=off)

    @staticmethod
    def from_polygons(polys: Sequence[Sequence[Sequence[Sequence[float]]]], close: bool = True) -> "GeoArrowArray":
        """polys[i] = [exterior, hole, ...]; each ring a list of (x, y).  Rings are closed on the
        way in when `close` (geo's `polygon!` macro closes rings too, spatial_index.rs:399-416)."""
        rings = []
        goff = [0]
        for poly in polys:
            for ring in poly:
                r = [tuple(map(float, c)) for c in ring]
                if close and r and r[0] != r[-1]:
                    r.append(r[0])
                rings.append(r)
            goff.append(len(rings))
        roff = np.zeros(len(rings) + 1, dtype=np.int32)
        roff[1:] = np.cumsum([len(r) for r in rings])
        xy = np.array([c for r in rings for c in r], dtype=np.float64).reshape(-1, 2)
        return GeoArrowArray(GEOM_POLYGON, xy, geom_offsets=np.array(goff, dtype=np.int32), ring_offsets=roff)

    @staticmethod
    def from_multipolygons(mps, close: bool = True) -> "GeoArrowArray":
        """mps[i] = [polygon, ...] with polygon = [exterior, hole, ...]."""
        flat = [p for mp in mps for p in mp]
        inner = GeoArrowArray.from_polygons(flat, close=close)
        goff = np.zeros(len(mps) + 1, dtype=np.int32)
        goff[1:] = np.cumsum([len(mp) for mp in mps])
        return GeoArrowArray(
            GEOM_MULTIPOLYGON, inner.xy, geom_offsets=goff, part_offsets=inner.geom_offsets, ring_offsets=inner.ring_offsets
        )

    @staticmethod
    def from_wkb(values: np.ndarray, offsets: np.ndarray, validity: Optional[np.ndarray] = None) -> "GeoArrowArray":
        """Decode an Arrow BinaryArray<i32> of WKB (values + offsets buffers) in two passes."""
        lib = _abi.lib()
        values = np.ascontiguousarray(values, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        vptr = _ptr(validity) if validity is not None else None
        counts = (C.c_int64 * 5)()
        _abi.check(lib.gpk_wkb_decode(values.ctypes.data, offsets.ctypes.data, n, vptr, counts, None, None, None, None))
        gt, _, n_parts, n_rings, n_coords = (int(c) for c in counts)
        xy = np.empty((n_coords, 2), dtype=np.float64)
        go = np.empty(n + 1, dtype=np.int32) if gt != GEOM_POINT else None
        po = np.empty(n_parts + 1, dtype=np.int32) if gt == GEOM_MULTIPOLYGON else None
        ro = np.empty(n_rings + 1, dtype=np.int32) if gt in (GEOM_POLYGON, GEOM_MULTILINESTRING, GEOM_MULTIPOLYGON) else None
        _abi.check(
            lib.gpk_wkb_decode(
                values.ctypes.data, offsets.ctypes.data, n, vptr, counts, xy.ctypes.data, _ptr(go), _ptr(po), _ptr(ro)
            )
        )
        return GeoArrowArray(gt, xy, geom_offsets=go, part_offsets=po, ring_offsets=ro, validity=validity, n_geoms=n)

    @staticmethod
    def from_arrow_wkb(column) -> "GeoArrowArray":
        """pyarrow Binary(Chunked)Array of WKB -> GeoArrowArray (rechunked to one chunk, like
        py-geopolars/src/ffi.rs:56)."""
        import pyarrow as pa

        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks()
        if pa.types.is_large_binary(column.type):
            column = column.cast(pa.binary())
        bufs = column.buffers()
        n = len(column)
        offsets = np.frombuffer(bufs[1], dtype=np.int32)[column.offset : column.offset + n + 1]
        values = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None else np.zeros(0, np.uint8)
        validity = None
        if column.null_count:
            bits = np.unpackbits(np.frombuffer(bufs[0], dtype=np.uint8), bitorder="little")[column.offset : column.offset + n]
            validity = np.packbits(bits, bitorder="little")
        return GeoArrowArray.from_wkb(values, offsets, validity)

    # ---- row surgery on the host (sampling for parity checks, assembling shards) -------------------------------
    def take(self, idx) -> "GeoArrowArray":
        """Rows `idx` (any order, repeats allowed) as a self-contained array: offsets rebuilt, coordinates gathered."""
        idx = np.asarray(idx, dtype=np.int64)
        validity = None
        if self.validity is not None:
            validity = np.packbits(self.is_valid()[idx].astype(np.uint8), bitorder="little")
        if self.geom_type == GEOM_POINT:
            return GeoArrowArray(GEOM_POINT, self.xy[idx], validity=validity)

        def gather(off, rows):  # children of `rows` under `off`: (new offsets, child indices)
            lo, hi = off[rows].astype(np.int64), off[rows + 1].astype(np.int64)
            n = hi - lo
            new = np.zeros(len(rows) + 1, dtype=np.int64)
            new[1:] = np.cumsum(n)
            child = np.repeat(lo - new[:-1], n) + np.arange(int(new[-1]), dtype=np.int64)
            return new.astype(np.int32), child

        levels = [o for o in (self.geom_offsets, self.part_offsets, self.ring_offsets) if o is not None]
        rows, new_levels = idx, []
        for off in levels:
            new, rows = gather(off, rows)
            new_levels.append(new)
        names = [k for k, o in zip(("geom_offsets", "part_offsets", "ring_offsets"), (self.geom_offsets, self.part_offsets, self.ring_offsets)) if o is not None]
        return GeoArrowArray(self.geom_type, self.xy[rows], validity=validity, **dict(zip(names, new_levels)))

    @staticmethod
    def concat(arrays: Sequence["GeoArrowArray"]) -> "GeoArrowArray":
        """Row-wise concatenation of arrays of one geometry type (offsets of piece k rebased by the child lengths before it)."""
        a0 = arrays[0]
        xy = np.concatenate([a.xy for a in arrays])
        kw = {}
        for name in ("geom_offsets", "part_offsets", "ring_offsets"):
            if getattr(a0, name) is None:
                continue
            pieces, base = [], 0
            for k, a in enumerate(arrays):
                o = getattr(a, name).astype(np.int64)
                pieces.append((o if k == 0 else o[1:]) + base)
                base += int(o[-1])
            kw[name] = np.concatenate(pieces).astype(np.int32)
        validity = None
        if any(a.validity is not None for a in arrays):
            validity = np.packbits(np.concatenate([a.is_valid() for a in arrays]).astype(np.uint8), bitorder="little")
        return GeoArrowArray(a0.geom_type, xy, validity=validity, **kw)

    # ---- export --------------------------------------------------------------------------------
    def to_wkb(self) -> tuple[np.ndarray, np.ndarray]:
        """-> (values uint8, offsets int32) of the WKB BinaryArray<i32> (from_geom_vec, util.rs:11-24), encoded on
        the host (gpk_wkb_encode); null rows are zero-length."""
        lib = _abi.lib()
        d = self.desc()
        n_bytes = C.c_int64(0)
        offsets = np.zeros(self.n_geoms + 1, dtype=np.int32)
        _abi.check(lib.gpk_wkb_encode(C.byref(d), offsets.ctypes.data, None, 0, C.byref(n_bytes)))
        values = np.empty(int(n_bytes.value), dtype=np.uint8)
        if len(values):
            _abi.check(lib.gpk_wkb_encode(C.byref(d), offsets.ctypes.data, values.ctypes.data, len(values), C.byref(n_bytes)))
        return values, offsets

    def to_arrow_wkb(self):
        """pyarrow binary array of WKB (the column type of every reference fixture)."""
        import pyarrow as pa

        values, offsets = self.to_wkb()
        valid = None if self.validity is None else pa.py_buffer(np.ascontiguousarray(self.validity).tobytes())
        return pa.Array.from_buffers(pa.binary(), self.n_geoms, [valid, pa.py_buffer(offsets.tobytes()), pa.py_buffer(values.tobytes())])

    def to_pyarrow(self):
        """GeoArrow nested list array (interleaved FixedSizeList<f64,2> coordinates)."""
        import pyarrow as pa

        coords = pa.FixedSizeListArray.from_arrays(pa.array(self.xy.reshape(-1), type=pa.float64()), 2)
        arr = coords
        levels = {
            GEOM_POINT: [],
            GEOM_LINESTRING: [self.geom_offsets],
            GEOM_MULTIPOINT: [self.geom_offsets],
            GEOM_POLYGON: [self.ring_offsets, self.geom_offsets],
            GEOM_MULTILINESTRING: [self.ring_offsets, self.geom_offsets],
            GEOM_MULTIPOLYGON: [self.ring_offsets, self.part_offsets, self.geom_offsets],
        }[self.geom_type]
        for off in levels:
            arr = pa.ListArray.from_arrays(pa.array(off, type=pa.int32()), arr)
        return arr


class DeviceGeoArray:
    """A gpk_geoarray handle: the array 'copied once to HBM' (or a zero-copy view of device buffers).

    Immutable after creation and shareable between threads, like `Arc<SpatialIndex>`
    (geopolars/src/spatial_index.rs:20-21)."""

    def __init__(self, handle: int, geom_type: int, n_geoms: int, n_coords: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self.geom_type = geom_type
        self.n_geoms = n_geoms
        self.n_coords = n_coords
        self._keepalive = keepalive

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise _abi.GeopolarsHipError(_abi.GPK_ERR_INVALID_ARGUMENT, "DeviceGeoArray used after free()")
        return self._h

    @staticmethod
    def upload(host: GeoArrowArray, stream: int = 0) -> "DeviceGeoArray":
        d = host.desc()
        out = C.c_void_p()
        _abi.check(_abi.lib().gpk_geoarray_upload(C.byref(d), stream, C.byref(out)))
        return DeviceGeoArray(out.value, host.geom_type, host.n_geoms, host.n_coords)

    @staticmethod
    def from_device_buffers(
        geom_type: int, xy, geom_offsets=None, part_offsets=None, ring_offsets=None, validity=None, stream: int = 0
    ) -> "DeviceGeoArray":
        """Zero-copy view over torch CUDA tensors (xy: (n,2) float64; offsets int32).  The tensors are
        kept alive by the returned object — the borrowed-buffer contract of the C ABI.  `xy` may also be a PAIR (x, y) of float64
        tensors — Struct<x, y> coordinates: they are interleaved on the device into a buffer the handle owns."""
        d = GeoArrowDesc()
        d.geom_type = geom_type
        d.mem_space = _abi.MEM_DEVICE
        if isinstance(xy, (tuple, list)):
            xs, ys = xy
            d.n_coords = xs.shape[0]
            d.x, d.y = xs.data_ptr(), ys.data_ptr()
            tensors = [xs, ys]
            xy = xs
        else:
            d.n_coords = xy.shape[0]
            d.xy = xy.data_ptr()
            tensors = [xy]
        for name, t in (("geom_offsets", geom_offsets), ("part_offsets", part_offsets), ("ring_offsets", ring_offsets), ("validity", validity)):
            if t is not None:
                setattr(d, name, t.data_ptr())
                tensors.append(t)
        d.n_geoms = xy.shape[0] if geom_type == GEOM_POINT else geom_offsets.shape[0] - 1
        d.n_parts = 0 if part_offsets is None else part_offsets.shape[0] - 1
        d.n_rings = 0 if ring_offsets is None else ring_offsets.shape[0] - 1
        out = C.c_void_p()
        _abi.check(_abi.lib().gpk_geoarray_upload(C.byref(d), stream, C.byref(out)))
        return DeviceGeoArray(out.value, geom_type, int(d.n_geoms), int(d.n_coords), keepalive=tensors)

    @staticmethod
    def from_wkb(values: np.ndarray, offsets: np.ndarray, validity: Optional[np.ndarray] = None, stream: int = 0) -> "DeviceGeoArray":
        """Arrow BinaryArray<i32> of WKB (host buffers) -> device-resident GeoArrow, decoded ON the GPU
        (gpk_geoarray_from_wkb): only the raw WKB bytes cross PCIe."""
        values = np.ascontiguousarray(values, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        out = C.c_void_p()
        gt = C.c_int32(-1)
        _abi.check(
            _abi.lib().gpk_geoarray_from_wkb(
                values.ctypes.data if len(values) else None, offsets.ctypes.data, n, _ptr(validity) if validity is not None else None, _abi.MEM_HOST, stream, C.byref(out), C.byref(gt)
            )
        )
        self = DeviceGeoArray(out.value, int(gt.value), n, -1)
        sizes = (C.c_int64 * 4)()
        _abi.check(_abi.lib().gpk_geoarray_download(self.handle, sizes, None, None, None, None, stream))
        self.n_coords = int(sizes[0])
        self._validity = None if validity is None else np.ascontiguousarray(validity, dtype=np.uint8)
        return self

    @staticmethod
    def from_arrow(column, geom_type: int = -1, stream: int = 0) -> "DeviceGeoArray":
        """A pyarrow geometry column -> device-resident GeoArrow through the Arrow C Data Interface (gpk_geoarray_from_arrow), the
        way the reference moves every Series across its FFI boundary (py-geopolars/src/ffi.rs:12-32: `_export_to_c` into two
        structs; chunked columns are rechunked first, :56).  WKB binary / large binary columns, and native GeoArrow nestings over
        Struct<x, y> (internals/geoseries.py:86-113) or FixedSizeList<f64, 2> coordinates; slices are fine.  `geom_type`:
        GEOM_MULTIPOINT / GEOM_MULTILINESTRING to read one / two list levels as those types (default LINESTRING / POLYGON)."""
        import pyarrow as pa

        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks() if column.num_chunks != 1 else column.chunk(0)
        c_array, c_schema = _abi.ArrowArray(), _abi.ArrowSchema()
        column._export_to_c(C.addressof(c_array), C.addressof(c_schema))
        out = C.c_void_p()
        gt = C.c_int32(-1)
        try:
            _abi.check(_abi.lib().gpk_geoarray_from_arrow(C.addressof(c_array), C.addressof(c_schema), geom_type, stream, C.byref(out), C.byref(gt)))
        finally:  # the importer's duty (the library only borrowed the structs)
            if c_array.release:
                c_array.release(C.byref(c_array))
            if c_schema.release:
                c_schema.release(C.byref(c_schema))
        self = DeviceGeoArray(out.value, int(gt.value), len(column), -1)
        sizes = (C.c_int64 * 4)()
        _abi.check(_abi.lib().gpk_geoarray_download(self.handle, sizes, None, None, None, None, stream))
        self.n_coords = int(sizes[0])
        return self

    def to_arrow(self, layout: str = "struct", stream: int = 0):
        """-> a pyarrow array imported from the two C Data Interface structs gpk_geoarray_to_arrow fills (callee-owned buffers, released by
        pyarrow through the structs' release callbacks): how a result leaves the reference (`to_py_array`, py-geopolars/src/ffi.rs:35-52).
        layout: "wkb" (Binary of ISO WKB, encoded on the GPU), "struct" (GeoArrow over Struct<x, y>: what the reference's Python layer
        builds) or "interleaved" (FixedSizeList<f64, 2>)."""
        import pyarrow as pa

        code = {"wkb": _abi.ARROW_WKB, "interleaved": _abi.ARROW_INTERLEAVED, "struct": _abi.ARROW_STRUCT}[layout]
        c_array, c_schema = _abi.ArrowArray(), _abi.ArrowSchema()
        _abi.check(_abi.lib().gpk_geoarray_to_arrow(self.handle, code, stream, C.addressof(c_array), C.addressof(c_schema)))
        return pa.Array._import_from_c(C.addressof(c_array), C.addressof(c_schema))  # (moves the structs: pyarrow calls release)

    def to_wkb(self, stream: int = 0) -> tuple[np.ndarray, np.ndarray]:
        """-> (values uint8, offsets int32): the WKB column, ENCODED ON THE GPU from the device-resident buffers
        (gpk_geoarray_to_wkb); only the finished bytes cross PCIe."""
        lib = _abi.lib()
        n_bytes = C.c_int64(0)
        offsets = np.zeros(self.n_geoms + 1, dtype=np.int32)
        _abi.check(lib.gpk_geoarray_to_wkb(self.handle, offsets.ctypes.data, None, 0, C.byref(n_bytes), _abi.MEM_HOST, stream))
        values = np.empty(int(n_bytes.value), dtype=np.uint8)
        if len(values):
            _abi.check(lib.gpk_geoarray_to_wkb(self.handle, offsets.ctypes.data, values.ctypes.data, len(values), C.byref(n_bytes), _abi.MEM_HOST, stream))
        return values, offsets

    def download(self, stream: int = 0) -> GeoArrowArray:
        """Copy the device buffers back as a host GeoArrowArray."""
        lib = _abi.lib()
        sizes = (C.c_int64 * 4)()
        _abi.check(lib.gpk_geoarray_download(self.handle, sizes, None, None, None, None, stream))
        n_coords, n_parts, n_rings, n_geoms = (int(v) for v in sizes)
        gt = self.geom_type
        xy = np.empty((n_coords, 2), dtype=np.float64)
        go = np.empty(n_geoms + 1, dtype=np.int32) if gt != GEOM_POINT else None
        po = np.empty(n_parts + 1, dtype=np.int32) if gt == GEOM_MULTIPOLYGON else None
        ro = np.empty(n_rings + 1, dtype=np.int32) if gt in (GEOM_POLYGON, GEOM_MULTILINESTRING, GEOM_MULTIPOLYGON) else None
        _abi.check(lib.gpk_geoarray_download(self.handle, sizes, xy.ctypes.data if n_coords else None, _ptr(go), _ptr(po), _ptr(ro), stream))
        validity = getattr(self, "_validity", None)
        if validity is None:  # the bitmap may exist on the device only (a decoded or exploded column)
            has = C.c_int32(0)
            bm = np.zeros((n_geoms + 7) // 8, dtype=np.uint8)
            _abi.check(lib.gpk_geoarray_validity(self.handle, bm.ctypes.data if len(bm) else None, C.byref(has), stream))
            validity = bm if has.value else None
        return GeoArrowArray(gt, xy, geom_offsets=go, part_offsets=po, ring_offsets=ro, validity=validity, n_geoms=n_geoms)

    @staticmethod
    def concat(chunks: Sequence["DeviceGeoArray"], stream: int = 0) -> tuple["DeviceGeoArray", np.ndarray]:
        """K device-resident chunks of one column -> (one array, first row of every chunk): Arrow's rechunk
        (py-geopolars/src/ffi.rs:56) on the device (gpk_geoarray_concat — the all-gatherv's own assembly, fed by device copies)."""
        k = len(chunks)
        arr = (C.c_void_p * k)(*[c.handle for c in chunks])
        out = C.c_void_p()
        bases = (C.c_int64 * (k + 1))()
        nb = C.c_int64(0)
        _abi.check(_abi.lib().gpk_geoarray_concat(arr, k, stream, C.byref(out), bases, C.byref(nb)))
        total = int(bases[k])
        return DeviceGeoArray(out.value, chunks[0].geom_type, total, -1), np.array(list(bases), dtype=np.int64)

    def nbytes(self) -> int:
        n = C.c_int64(0)
        _abi.check(_abi.lib().gpk_geoarray_nbytes(self.handle, C.byref(n)))
        return int(n.value)

    def invalidate(self) -> None:
        """gpk_geoarray_invalidate: the owner of BORROWED device buffers has rewritten their offsets in place — drop what the handle
        derived from them (size classes, strip tables); coordinates may be rewritten without this"""
        _abi.check(_abi.lib().gpk_geoarray_invalidate(self.handle))

    def free(self) -> None:
        if self._h:
            _abi.lib().gpk_geoarray_free(self._h)
            self._h = C.c_void_p()
            self._keepalive = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
