"""`SpatialIndex` / `spatial_join` — host-side mirror of geopolars/src/spatial_index.rs.

    SpatialJoinArgs         spatial_index.rs:15-35   (join_type, predicate, suffixes, prebuilt indexes)
    SpatialIndex            spatial_index.rs:314-350 (TryFrom<&Series>)
    spatial_join            spatial_index.rs:37-204

The candidate generation + exact refine (spatial_index.rs:74-143) run on the GPU through
gpk_spatial_join; this module only marshals buffers and — for dataframe-shaped callers — assembles
the joined table from the (l, r) index pairs the way spatial_index.rs:145-203 does with polars
joins (here: pyarrow `take`, polars is not installed).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _abi
from ._abi import MEM_DEVICE, MEM_HOST, PREDICATES
from .geoarrow import DeviceGeoArray
from .geoseries import GeoSeries


class SpatialIndex:
    """Device-resident bbox grid directory over one series (the R-tree's replacement)."""

    def __init__(self, series: GeoSeries, stream: int = 0, for_points: bool = True, full: bool = False, light: bool = False):
        """for_points=False skips the point-in-polygon raster + edge slabs (only point x polygonal joins read them);
        full=True builds per-entry records for every list cell (GPK_INDEX_PIP_FULL: an index that serves hundreds of joins)."""
        self.series = series
        h = C.c_void_p()
        parts = _abi.INDEX_BBOX_GRID | (_abi.INDEX_PIP if for_points else 0) | (_abi.INDEX_PIP_FULL if for_points and full else 0)
        if for_points and light and not full:  # an index that serves ONE join (what gpk_spatial_join builds for itself without r_index)
            parts |= _abi.INDEX_PIP_LIGHT
        _abi.check(_abi.lib().gpk_index_build_ex(series.device().handle, parts, None, stream, C.byref(h)))
        self._h = h

    @staticmethod
    def from_device(dev: DeviceGeoArray, stream: int = 0, for_points: bool = True, bboxes=None, full: bool = False, light: bool = False) -> "SpatialIndex":
        """Index over a device-resident array.  `bboxes`: optional (n, 4) float64 CUDA tensor of precomputed leaves
        (minx, miny, maxx, maxy per geometry, e.g. all-gathered from the ranks that own the shards)."""
        self = SpatialIndex.__new__(SpatialIndex)
        self.series = None
        self._dev = dev
        h = C.c_void_p()
        parts = _abi.INDEX_BBOX_GRID | (_abi.INDEX_PIP if for_points else 0) | (_abi.INDEX_PIP_FULL if for_points and full else 0)
        if for_points and light and not full:  # an index that serves ONE join (what gpk_spatial_join builds for itself without r_index)
            parts |= _abi.INDEX_PIP_LIGHT
        _abi.check(_abi.lib().gpk_index_build_ex(dev.handle, parts, bboxes.data_ptr() if bboxes is not None else None, stream, C.byref(h)))
        self._h = h
        return self

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def nbytes(self) -> int:
        n = C.c_int64(0)
        _abi.check(_abi.lib().gpk_index_nbytes(self._h, C.byref(n)))
        return int(n.value)

    def describe(self) -> dict:
        """gpk_index_describe: which point-in-polygon tables the index carries (raster side, lean, chains, routing image)."""
        out = (C.c_int64 * 8)()
        _abi.check(_abi.lib().gpk_index_describe(self._h, out))
        return {"R": int(out[0]), "lean": bool(out[1]), "chains": bool(out[2]), "route": bool(out[3]), "list_heavy": bool(out[4])}

    def query_envelopes(self, boxes, mode: str = "contained", stream: int = 0) -> tuple[np.ndarray, np.ndarray]:
        """gpk_index_query_envelope for a batch of query boxes ((n, 4): minx, miny, maxx, maxy): the (query, geometry index) pairs
        sorted by (query, index) and the per-query counts.  mode: "contained" (rstar locate_in_envelope) | "intersecting"
        (locate_in_envelope_intersecting); closed intervals (spatial_index.rs:383-393,422-429)."""
        lib = _abi.lib()
        b = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 4))
        n = len(b)
        m = {"contained": _abi.QUERY_CONTAINED, "intersecting": _abi.QUERY_INTERSECTING}[mode]
        counts = np.zeros(n, dtype=np.uint32)
        n_pairs = C.c_int64(0)
        _abi.check(lib.gpk_index_query_envelope(self._h, b.ctypes.data, n, m, counts.ctypes.data, None, 0, C.byref(n_pairs), MEM_HOST, stream))
        pairs = np.zeros((int(n_pairs.value), 2), dtype=np.uint32)
        if len(pairs):
            _abi.check(lib.gpk_index_query_envelope(self._h, b.ctypes.data, n, m, counts.ctypes.data, pairs.ctypes.data, len(pairs), C.byref(n_pairs), MEM_HOST, stream))
        return pairs, counts

    def locate_in_envelope(self, lower, upper) -> np.ndarray:
        """`r_tree.locate_in_envelope(&AABB::from_corners(lower, upper))` (spatial_index.rs:383-387): the indexes of the geometries
        whose bounding box lies inside the closed query box, ascending."""
        pairs, _ = self.query_envelopes([[lower[0], lower[1], upper[0], upper[1]]], "contained")
        return pairs[:, 1].astype(np.int64)

    def locate_in_envelope_intersecting(self, lower, upper) -> np.ndarray:
        """rstar `locate_in_envelope_intersecting`: the geometries whose bounding box meets the closed query box."""
        pairs, _ = self.query_envelopes([[lower[0], lower[1], upper[0], upper[1]]], "intersecting")
        return pairs[:, 1].astype(np.int64)

    def free(self) -> None:
        if self._h:
            _abi.lib().gpk_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


@dataclass
class SpatialJoinArgs:
    """spatial_index.rs:15-35; defaults from `impl Default` (spatial_index.rs:24-35)."""

    join_type: str = "inner"  # JoinType::Inner | "left"
    predicate: str = "intersects"  # Predicate::Intersects
    l_suffix: Optional[str] = "_left"
    r_suffix: Optional[str] = "_right"
    l_index: Optional[SpatialIndex] = None  # accepted for signature parity; only the right index is used
    r_index: Optional[SpatialIndex] = None
    # not in the reference (its geometry columns are WKB, which names its type per row): the geometry type of a NATIVE GeoArrow column
    # whose nesting alone does not tell it — one list level is a LineString or a MultiPoint, two a Polygon or a MultiLineString — and
    # which carries no ARROW:extension:name (geoarrow.*).  spatial_join refuses such a column without a hint rather than guess.
    l_geom_type: int = -1
    r_geom_type: int = -1


def join_pairs(
    left: GeoSeries,
    right: GeoSeries,
    predicate: str = "intersects",
    r_index: Optional[SpatialIndex] = None,
    left_row_base: int = 0,
) -> tuple[np.ndarray, np.ndarray]:
    """All (l, r) index pairs with predicate(left[l], right[r]), sorted by (l, r), plus the per-left-row
    hit counts.  Host-buffer variant: sizes the pair buffer with a count-only first call."""
    lib = _abi.lib()
    n = len(left)
    counts = np.empty(n, dtype=np.uint32)
    n_pairs = C.c_int64(0)
    rh = r_index.handle if r_index is not None else None
    pred = PREDICATES[predicate]
    capacity = max(1024, 4 * n)  # one call in the common case; the ABI reports the exact total when this is too small
    while True:
        pairs = np.empty((capacity, 2), dtype=np.uint32)
        rc = lib.gpk_spatial_join(
            left.device().handle, right.device().handle, rh, pred, left_row_base, counts.ctypes.data, pairs.ctypes.data, capacity, C.byref(n_pairs), MEM_HOST, None
        )
        if rc == _abi.GPK_ERR_CAPACITY and int(n_pairs.value) > capacity:
            capacity = int(n_pairs.value)
            continue
        _abi.check(rc)
        return pairs[: int(n_pairs.value)].copy(), counts


def join_pairs_device(left: DeviceGeoArray, right: DeviceGeoArray, r_index: SpatialIndex, predicate: str, out_counts, out_pairs, left_row_base: int = 0, stream: int = 0) -> int:
    """Device-buffer variant (bench / multi-GPU path): out_counts (n,) uint32-as-int32 and out_pairs
    (cap, 2) torch CUDA tensors are filled in place on `stream`; returns the number of pairs."""
    n_pairs = C.c_int64(0)
    _abi.check(
        _abi.lib().gpk_spatial_join(
            left.handle,
            right.handle,
            r_index.handle if r_index is not None else None,
            PREDICATES[predicate],
            left_row_base,
            out_counts.data_ptr() if out_counts is not None else None,
            out_pairs.data_ptr() if out_pairs is not None else None,
            out_pairs.shape[0] if out_pairs is not None else 0,
            C.byref(n_pairs),
            MEM_DEVICE,
            stream,
        )
    )
    return int(n_pairs.value)


def join_pairs_enqueue(left: DeviceGeoArray, right: DeviceGeoArray, r_index: SpatialIndex, predicate: str, out_counts, out_pairs, n_pairs_out, left_row_base: int = 0, stream: int = 0) -> None:
    """Stream-ordered variant of join_pairs_device (gpk_spatial_join_async): the join is enqueued on `stream` and
    this returns without waiting.  n_pairs_out: a 1-element int64 CUDA tensor that receives the total (read it
    after synchronising; pairs beyond out_pairs' capacity are dropped)."""
    _abi.check(
        _abi.lib().gpk_spatial_join_async(
            left.handle,
            right.handle,
            r_index.handle,
            PREDICATES[predicate],
            left_row_base,
            out_counts.data_ptr() if out_counts is not None else None,
            out_pairs.data_ptr() if out_pairs is not None else None,
            out_pairs.shape[0] if out_pairs is not None else 0,
            n_pairs_out.data_ptr() if n_pairs_out is not None else None,
            stream,
        )
    )


JOIN_TYPES = {"inner": 0, "left": 1}  # GPK_JOIN_*


def join_indices(counts: np.ndarray, pairs: np.ndarray, join_type: str = "inner", left_row_base: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """(hit counts, sorted (l, r) pairs) -> i64 row indices (l, r) of the joined table (gpk_join_indices; the two
    u64 index Series of spatial_index.rs:147-159 followed by inner_join / left_join).  Left join: unmatched left
    rows appear once with r = -1."""
    lib = _abi.lib()
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    n_rows = C.c_int64(0)
    jt = JOIN_TYPES[join_type]
    args = (counts.ctypes.data, pairs.ctypes.data if len(pairs) else None, len(counts), len(pairs), left_row_base, jt)
    _abi.check(lib.gpk_join_indices(*args, None, None, 0, C.byref(n_rows), MEM_HOST, None))
    n = int(n_rows.value)
    li, ri = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    if n:
        _abi.check(lib.gpk_join_indices(*args, li.ctypes.data, ri.ctypes.data, n, C.byref(n_rows), MEM_HOST, None))
    return li, ri


def take_column(column, idx: np.ndarray):
    """pyarrow column gathered by i64 row indices on the GPU (gpk_take_fixed / gpk_take_binary); -1 gives a null.
    Fixed-width primitives, booleans, binary and string columns — the types the reference's fixtures carry; anything
    else is reported (there is no host fallback)."""
    import pyarrow as pa

    lib = _abi.lib()
    arr = column.combine_chunks() if isinstance(column, pa.ChunkedArray) else column
    if arr.offset != 0:
        arr = pa.concat_arrays([arr])  # re-base a sliced array so the raw buffers start at row 0
    t = arr.type
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n_idx, n = len(idx), len(arr)
    bufs = arr.buffers()
    validity = np.frombuffer(bufs[0], dtype=np.uint8) if bufs[0] is not None else None
    vptr = validity.ctypes.data if validity is not None else None
    out_valid = np.zeros((n_idx + 7) // 8, dtype=np.uint8)
    if pa.types.is_binary(t) or pa.types.is_string(t):
        offsets = np.frombuffer(bufs[1], dtype=np.int32)[: n + 1] if bufs[1] is not None else np.zeros(1, np.int32)
        values = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
        n_bytes = C.c_int64(0)
        out_off = np.zeros(n_idx + 1, dtype=np.int32)
        args = (values.ctypes.data, offsets.ctypes.data, vptr, n, idx.ctypes.data, n_idx, out_off.ctypes.data)
        _abi.check(lib.gpk_take_binary(*args, None, 0, C.byref(n_bytes), out_valid.ctypes.data, MEM_HOST, None))
        out_vals = np.empty(max(int(n_bytes.value), 1), dtype=np.uint8)
        if n_bytes.value:
            _abi.check(lib.gpk_take_binary(*args, out_vals.ctypes.data, int(n_bytes.value), C.byref(n_bytes), out_valid.ctypes.data, MEM_HOST, None))
        return pa.Array.from_buffers(t, n_idx, [pa.py_buffer(out_valid.tobytes()), pa.py_buffer(out_off.tobytes()), pa.py_buffer(out_vals[: int(n_bytes.value)].tobytes())])
    if pa.types.is_boolean(t):
        bits = 1
    elif pa.types.is_primitive(t):
        bits = t.bit_width
    else:
        raise _abi.GeopolarsHipError(_abi.GPK_ERR_INVALID_ARGUMENT, f"join assembly: column type {t} is not supported")
    data = np.frombuffer(bufs[1], dtype=np.uint8) if bufs[1] is not None else np.zeros(16, np.uint8)
    out = np.zeros((n_idx + 7) // 8 if bits == 1 else n_idx * (bits // 8), dtype=np.uint8)
    _abi.check(lib.gpk_take_fixed(data.ctypes.data, bits, vptr, n, idx.ctypes.data, n_idx, out.ctypes.data if len(out) else None, out_valid.ctypes.data, MEM_HOST, None))
    return pa.Array.from_buffers(t, n_idx, [pa.py_buffer(out_valid.tobytes()), pa.py_buffer(out.tobytes())])


def _geometry_type_of(table, hint: int, hint_name: str) -> int:
    """the geom_type to import a table's geometry column with: the caller's hint, else -1 (WKB names its types; a GeoArrow extension
    name does; three list levels and bare coordinates can only be one type) — or an error for one / two list levels without either"""
    import pyarrow as pa

    if hint >= 0:
        return hint
    t = table.schema.field("geometry").type
    ext = (table.schema.field("geometry").metadata or {}).get(b"ARROW:extension:name")
    if isinstance(t, pa.BaseExtensionType):
        ext, t = t.extension_name.encode(), t.storage_type
    depth = 0
    while pa.types.is_list(t) or pa.types.is_large_list(t):
        depth, t = depth + 1, t.value_type
    if depth in (1, 2) and not ext:
        kinds = "LineString or MultiPoint" if depth == 1 else "Polygon or MultiLineString"
        raise _abi.GeopolarsHipError(
            _abi.GPK_ERR_INVALID_ARGUMENT,
            f"spatial_join: a GeoArrow geometry column of {depth} list level(s) without an ARROW:extension:name is a {kinds} column: "
            f"name the type in SpatialJoinArgs.{hint_name}",
        )
    return -1


def spatial_join(lhs, rhs, options: Optional[SpatialJoinArgs] = None):
    """spatial_join(lhs, rhs, SpatialJoinArgs) over pyarrow Tables with a `geometry` column (spatial_index.rs:44-45) — WKB binary as the
    reference holds it, or a native GeoArrow nesting.  Returns a pyarrow Table shaped like the reference's result: suffixed left columns,
    then suffixed right columns (spatial_index.rs:165-199).  The geometry columns cross into the library the way every Series crosses
    the reference's FFI — as Arrow C Data Interface structs (py-geopolars/src/ffi.rs:12-32; gpk_geoarray_from_arrow): WKB is decoded
    on the GPU, nothing is rewritten on the host."""
    import pyarrow as pa

    options = options or SpatialJoinArgs()
    if options.join_type not in ("inner", "left"):
        # spatial_index.rs:200-202 rejects every other JoinType
        raise _abi.GeopolarsHipError(_abi.GPK_ERR_INVALID_ARGUMENT, "Failed to generate the spatial index for the left dataframe")
    lgeo = GeoSeries.from_arrow(lhs.column("geometry"), _geometry_type_of(lhs, options.l_geom_type, "l_geom_type"))
    rgeo = GeoSeries.from_arrow(rhs.column("geometry"), _geometry_type_of(rhs, options.r_geom_type, "r_geom_type"))
    r_index = options.r_index or SpatialIndex(rgeo)
    pairs, counts = join_pairs(lgeo, rgeo, options.predicate, r_index)
    li, ri = join_indices(counts, pairs, options.join_type)  # i64 row indices, r = -1 for unmatched left rows
    def as_wkb(column, geo: GeoSeries):
        """a native GeoArrow geometry column leaves the join the way the reference's geometry columns are held — WKB binary
        (from_geom_vec, util.rs:11-24) — encoded on the GPU from the series the join already uploaded"""
        if pa.types.is_binary(column.type) or pa.types.is_large_binary(column.type):
            return column
        return geo.device().to_arrow("wkb")  # (gpk_geoarray_to_arrow: the library's buffers, validity included, released by pyarrow)

    cols, names = [], []
    for name in lhs.column_names:
        cols.append(take_column(as_wkb(lhs.column(name), lgeo) if name == "geometry" else lhs.column(name), li))
        names.append(name + (options.l_suffix or ""))
    for name in rhs.column_names:
        cols.append(take_column(as_wkb(rhs.column(name), rgeo) if name == "geometry" else rhs.column(name), ri))
        names.append(name + (options.r_suffix or ""))
    return pa.table(cols, names=names)
