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

    def __init__(self, series: GeoSeries, stream: int = 0):
        self.series = series
        h = C.c_void_p()
        _abi.check(_abi.lib().gpk_index_build(series.device().handle, stream, C.byref(h)))
        self._h = h

    @staticmethod
    def from_device(dev: DeviceGeoArray, stream: int = 0) -> "SpatialIndex":
        self = SpatialIndex.__new__(SpatialIndex)
        self.series = None
        self._dev = dev
        h = C.c_void_p()
        _abi.check(_abi.lib().gpk_index_build(dev.handle, stream, C.byref(h)))
        self._h = h
        return self

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def nbytes(self) -> int:
        n = C.c_int64(0)
        _abi.check(_abi.lib().gpk_index_nbytes(self._h, C.byref(n)))
        return int(n.value)

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


def spatial_join(lhs, rhs, options: Optional[SpatialJoinArgs] = None):
    """spatial_join(lhs, rhs, SpatialJoinArgs) over pyarrow Tables with a WKB `geometry` column
    (spatial_index.rs:44-45).  Returns a pyarrow Table shaped like the reference's result:
    suffixed left columns, then suffixed right columns (spatial_index.rs:165-199)."""
    import pyarrow as pa

    options = options or SpatialJoinArgs()
    if options.join_type not in ("inner", "left"):
        # spatial_index.rs:200-202 rejects every other JoinType
        raise _abi.GeopolarsHipError(_abi.GPK_ERR_INVALID_ARGUMENT, "Failed to generate the spatial index for the left dataframe")
    lgeo = GeoSeries.from_wkb(lhs.column("geometry"))
    rgeo = GeoSeries.from_wkb(rhs.column("geometry"))
    r_index = options.r_index or SpatialIndex(rgeo)
    pairs, counts = join_pairs(lgeo, rgeo, options.predicate, r_index)
    li = pairs[:, 0].astype(np.int64)
    ri = pairs[:, 1].astype(np.int64)
    if options.join_type == "left":
        # left join keeps unmatched left rows with null right columns
        unmatched = np.nonzero(counts == 0)[0]
        li_all = np.concatenate([li, unmatched])
        ri_all = np.concatenate([ri, np.full(len(unmatched), -1, dtype=np.int64)])
        order = np.argsort(li_all, kind="stable")
        li, ri = li_all[order], ri_all[order]
    cols, names = [], []
    for name in lhs.column_names:
        cols.append(lhs.column(name).combine_chunks().take(pa.array(li)))
        names.append(name + (options.l_suffix or ""))
    ri_arr = pa.array(ri, mask=ri < 0)
    for name in rhs.column_names:
        cols.append(rhs.column(name).combine_chunks().take(ri_arr))
        names.append(name + (options.r_suffix or ""))
    return pa.table(cols, names=names)
