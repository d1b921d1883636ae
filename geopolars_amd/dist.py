"""Row sharding and the one collective of the path.

Every operator is a per-row map (geoseries.rs:141 "1-to-1 row-wise"; the join refine is independent
per candidate pair, spatial_index.rs:83-143), so N GPUs = N independent row ranges of the LEFT
series, one process per GPU.  The right side is either replicated (C2: 1 MB of polygons ->
`broadcast_geoarray`) or, when it is itself produced sharded (C4/C5), exchanged ONCE with an
all-gatherv of its GeoArrow buffers over RCCL/xGMI (`all_gatherv_buffers`): RCCL has no native
`v` collective, so every buffer is padded to the longest shard, all-gathered with ONE fixed-size
collective, trimmed and — for offsets — rebased, all on the device the shards live on: nothing is
staged through the host.  Lengths travel first in one small header all-gather, so every rank issues
exactly the same sequence of collectives whatever its own shard holds (validity bitmap or not,
empty or not).  `all_gather_leaves` ships what each rank BUILT for its shard — the per-geometry
bounding boxes, the leaves of the reference's R-tree (spatial_index.rs:206-312) — so the gathered
index is assembled from them (gpk_index_build_ex) instead of re-deriving them on every rank.
No collective touches results: output row ranges are disjoint and pairs carry a per-shard
`left_row_base`.

Works on CPU tensors with the gloo backend (tests, world_size 2) and CUDA tensors with nccl (= RCCL);
the code path is the same, torch.distributed is the plumbing.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from ._abi import GEOM_MULTIPOLYGON, GEOM_POINT
from .geoarrow import GeoArrowArray


def shard_rows(n_rows: int, world: int, rank: int, weights: Optional[np.ndarray] = None) -> tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`.  With `weights` (e.g. vertices per row) the cut points
    balance total weight instead of row count — power-law geometries (C5) need that."""
    if weights is None:
        base, rem = divmod(n_rows, world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n_rows
    cuts = np.maximum.accumulate(np.minimum(cuts, n_rows))
    return int(cuts[rank]), int(cuts[rank + 1])


def slice_rows(a: GeoArrowArray, lo: int, hi: int) -> GeoArrowArray:
    """Rows [lo, hi) as a self-contained array (offsets rebased to 0)."""
    if a.validity is not None:
        bits = np.unpackbits(a.validity, bitorder="little")[lo:hi]
        validity = np.packbits(bits, bitorder="little")
    else:
        validity = None
    if a.geom_type == GEOM_POINT:
        return GeoArrowArray(a.geom_type, a.xy[lo:hi], validity=validity)
    go = a.geom_offsets[lo : hi + 1]
    if a.ring_offsets is None:
        return GeoArrowArray(a.geom_type, a.xy[go[0] : go[-1]], geom_offsets=go - go[0], validity=validity)
    if a.geom_type != GEOM_MULTIPOLYGON:
        ro = a.ring_offsets[go[0] : go[-1] + 1]
        return GeoArrowArray(a.geom_type, a.xy[ro[0] : ro[-1]], geom_offsets=go - go[0], ring_offsets=ro - ro[0], validity=validity)
    po = a.part_offsets[go[0] : go[-1] + 1]
    ro = a.ring_offsets[po[0] : po[-1] + 1]
    return GeoArrowArray(
        a.geom_type, a.xy[ro[0] : ro[-1]], geom_offsets=go - go[0], part_offsets=po - po[0], ring_offsets=ro - ro[0], validity=validity
    )


# ---- GeoArrow buffers as tensors of one device --------------------------------------------------------
@dataclass
class GeoBuffers:
    """The buffers of one single-chunk GeoArrow array as torch tensors living on ONE device (a GPU under RCCL,
    the CPU under gloo): what the exchange moves and what DeviceGeoArray.from_device_buffers borrows."""

    geom_type: int
    xy: torch.Tensor  # (n_coords, 2) float64
    geom_offsets: Optional[torch.Tensor] = None  # int32
    part_offsets: Optional[torch.Tensor] = None
    ring_offsets: Optional[torch.Tensor] = None
    valid: Optional[torch.Tensor] = None  # (n_geoms,) uint8, one byte per row (packed to an Arrow bitmap on demand)

    @property
    def n_geoms(self) -> int:
        return int(self.xy.shape[0]) if self.geom_type == GEOM_POINT else int(self.geom_offsets.shape[0]) - 1

    @staticmethod
    def from_host(a: GeoArrowArray, device: torch.device) -> "GeoBuffers":
        def t(x):
            return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(device)

        valid = None if a.validity is None else torch.from_numpy(a.is_valid().astype(np.uint8)).to(device)
        return GeoBuffers(a.geom_type, t(a.xy), t(a.geom_offsets), t(a.part_offsets), t(a.ring_offsets), valid)

    def validity_bitmap(self) -> Optional[torch.Tensor]:
        """Arrow LSB-first bitmap of `valid`, packed on the device."""
        if self.valid is None:
            return None
        n = self.valid.shape[0]
        pad = (-n) % 8
        v = torch.cat([self.valid, torch.zeros(pad, dtype=torch.uint8, device=self.valid.device)]) if pad else self.valid
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=v.device)
        return (v.reshape(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8)

    def to_host(self) -> GeoArrowArray:
        def h(x):
            return None if x is None else x.cpu().numpy()

        bm = self.validity_bitmap()
        return GeoArrowArray(self.geom_type, h(self.xy), geom_offsets=h(self.geom_offsets), part_offsets=h(self.part_offsets), ring_offsets=h(self.ring_offsets), validity=h(bm), n_geoms=self.n_geoms)

    def to_device_geoarray(self, stream: int = 0):
        """Zero-copy handle over these tensors (they must live in HBM); the tensors are kept alive by the handle."""
        from .geoarrow import DeviceGeoArray

        return DeviceGeoArray.from_device_buffers(self.geom_type, self.xy, self.geom_offsets, self.part_offsets, self.ring_offsets, self.validity_bitmap(), stream=stream)


_LEVELS = ("geom_offsets", "part_offsets", "ring_offsets")


def _pad_gather(t: torch.Tensor, lens: list[int], group) -> list[torch.Tensor]:
    """All-gatherv of a variable-length buffer WITHOUT padding (round 5; it used to pad every shard to the longest one and run one
    fixed-size all-gather): one allocation of the exact total, every rank's piece broadcast straight to its final place — the
    grouped-broadcast form of the library's own collective (csrc/gpk_comm.hip) — so the bytes on the links are the bytes of the
    column whatever the shards' sizes.  `lens` are the first-dimension lengths of every rank's shard (from the header): no length
    exchange happens here.  Returns the per-rank views of the gathered buffer (contiguous, in rank order)."""
    world = len(lens)
    rank = dist.get_rank(group)
    out = torch.empty((sum(lens),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    views, at = [], 0
    for k in range(world):
        views.append(out[at : at + lens[k]])
        at += lens[k]
    if lens[rank]:
        views[rank].copy_(t[: lens[rank]])
    work = []
    for k in range(world):
        if lens[k]:
            src = dist.get_global_rank(group, k) if group is not None else k
            work.append(dist.broadcast(views[k], src, group=group, async_op=True))
    for w in work:
        w.wait()
    return views


def all_gatherv_buffers(local: GeoBuffers, group=None, stats: Optional[dict] = None) -> GeoBuffers:
    """Every rank contributes its shard of the RIGHT side and receives the concatenation in rank order (the north
    star's "RCCL all-gatherv over xGMI"), device-resident end to end.  Offsets of shard k are rebased by the child
    lengths of shards 0..k-1, which the header already carries (an offsets buffer ends with its child's length)."""
    world = dist.get_world_size(group)
    dev = local.xy.device
    present = [getattr(local, k) is not None for k in _LEVELS]
    hdr = torch.tensor(
        [local.xy.shape[0]] + [getattr(local, k).shape[0] if p else 0 for k, p in zip(_LEVELS, present)] + [1 if local.valid is not None else 0, local.n_geoms],
        dtype=torch.int64,
        device=dev,
    )
    all_hdr = torch.empty(world * hdr.shape[0], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_hdr, hdr, group=group)
    H = all_hdr.cpu().reshape(world, -1).tolist()  # 6 small integers per rank: the only host read of the exchange
    n_coords = [h[0] for h in H]
    nbytes = 0
    xy = torch.cat(_pad_gather(local.xy, n_coords, group))
    nbytes += xy.numel() * 8
    out = GeoBuffers(local.geom_type, xy)
    # child length of each level: the next present level's row count, the coordinates for the innermost
    child_idx = {"geom_offsets": 2 if present[1] else (3 if present[2] else 0), "part_offsets": 3, "ring_offsets": 0}
    for li, name in enumerate(_LEVELS):
        if not present[li]:
            continue
        lens = [h[1 + li] for h in H]
        parts = _pad_gather(getattr(local, name), lens, group)
        ci = child_idx[name]
        pieces, base = [], 0
        for k, p in enumerate(parts):
            child_len = (H[k][ci] - 1) if ci in (2, 3) else H[k][0]  # rows of the child level, or coordinates
            pieces.append((p if k == 0 else p[1:]) + base)
            base += max(child_len, 0)
        cat = torch.cat(pieces).to(torch.int32)
        setattr(out, name, cat)
        nbytes += cat.numel() * 4
    if any(h[4] for h in H):  # some shard carries nulls: every rank gathers one byte per row (same collective everywhere)
        n_geoms = [h[5] for h in H]
        mine = local.valid if local.valid is not None else torch.ones(local.n_geoms, dtype=torch.uint8, device=dev)
        out.valid = torch.cat(_pad_gather(mine, n_geoms, group))
        nbytes += out.valid.numel()
    if stats is not None:
        stats["gathered_bytes"] = nbytes
    return out


def all_gather_leaves(local_bbox: torch.Tensor, group=None) -> torch.Tensor:
    """(n_local, 4) float64 per-geometry boxes each rank computed for ITS shard (gpk_bounds) -> the (n_total, 4) boxes
    of the gathered right side in rank order: the R-tree leaves of spatial_index.rs:206-312 (NodeEnvelope), shipped
    instead of recomputed.  Feed the result to SpatialIndex.from_device(..., bboxes=...)."""
    world = dist.get_world_size(group)
    n = torch.tensor([local_bbox.shape[0]], dtype=torch.int64, device=local_bbox.device)
    lens = torch.empty(world, dtype=torch.int64, device=local_bbox.device)
    dist.all_gather_into_tensor(lens, n, group=group)
    return torch.cat(_pad_gather(local_bbox.reshape(-1, 4), [int(v) for v in lens.cpu().tolist()], group))


def all_gatherv_geoarray(local: GeoArrowArray, device: Optional[torch.device] = None, group=None) -> GeoArrowArray:
    """Host-array convenience over all_gatherv_buffers: the shard goes to `device` once, the exchange runs there, the
    result comes back once."""
    device = device or torch.device("cpu")
    return all_gatherv_buffers(GeoBuffers.from_host(local, device), group).to_host()


def broadcast_buffers(local: Optional[GeoBuffers], src: int = 0, device: Optional[torch.device] = None, group=None) -> GeoBuffers:  # (device: required)
    """Replicate a small right side (C2's 1k polygons) from `src` to every rank, device to device: a header, then each buffer with
    one broadcast.  `local` is read on `src` only (its tensors already live on `device`); the result stays on `device` — what
    `to_device_geoarray()` borrows (no host round trip: round 3's replicate went through numpy on every rank)."""
    if device is None:  # (every rank must name the same kind of device: the source's tensors' device is unknown elsewhere)
        raise ValueError("broadcast_buffers: pass `device` on every rank (the source's tensors live on it)")
    rank = dist.get_rank(group)
    hdr = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == src:
        lv = [getattr(local, k) for k in _LEVELS]
        hdr[:7] = torch.tensor(
            [local.geom_type, local.n_geoms, int(local.xy.shape[0])] + [-1 if t is None else int(t.shape[0]) for t in lv] + [0 if local.valid is None else 1]
        )
    dist.broadcast(hdr, src, group=group)
    gt, n_geoms, n_coords, n_go, n_po, n_ro, has_valid = (int(v) for v in hdr[:7].tolist())

    def bc(t, shape, dtype):
        t = t.contiguous() if rank == src else torch.empty(shape, dtype=dtype, device=device)
        dist.broadcast(t, src, group=group)
        return t

    xy = bc(local.xy if rank == src else None, (n_coords, 2), torch.float64)
    levels = [bc(getattr(local, k) if rank == src else None, (n,), torch.int32) if n >= 0 else None for k, n in zip(_LEVELS, (n_go, n_po, n_ro))]
    valid = bc(local.valid if rank == src else None, (n_geoms,), torch.uint8) if has_valid else None
    return GeoBuffers(gt, xy, levels[0], levels[1], levels[2], valid)


def broadcast_geoarray(a: Optional[GeoArrowArray], src: int = 0, device: Optional[torch.device] = None, group=None) -> GeoArrowArray:
    """The host-array form of broadcast_buffers (a host column in, a host column out on every rank)."""
    device = device or torch.device("cpu")
    local = GeoBuffers.from_host(a, device) if dist.get_rank(group) == src else None
    return broadcast_buffers(local, src, device, group).to_host()


# ---- the same exchange behind the C ABI (gpk_comm_*, gpk_allgatherv_*: RCCL opened by the library itself) ---------------------
def all_gather_points(xy_local: torch.Tensor, group=None) -> tuple[torch.Tensor, list[int]]:
    """Every rank's (n_k, 2) float64 point shard -> all points in rank order (+ the shard lengths).  16 bytes per row: for a
    ONE-SHOT point join the points are the cheap side to replicate (`join_partition_right`)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return xy_local, [int(xy_local.shape[0])]
    n = torch.tensor([xy_local.shape[0]], dtype=torch.int64, device=xy_local.device)
    lens_t = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens_t, n, group=group)
    lens = [int(t.item()) for t in lens_t]
    pieces = _pad_gather(xy_local.reshape(-1), [2 * k for k in lens], group)
    return torch.cat(pieces).reshape(-1, 2), lens


def join_partition_right(xy_local: torch.Tensor, right_shard, right_row_base: int, predicate: str = "within", group=None, stream: int = 0, pair_capacity: Optional[int] = None) -> dict:
    """A ONE-SHOT point x polygonal join over N GPUs that partitions the RIGHT side (the seam is where
    `spatial_join` builds the index of the series it is handed, spatial_index.rs:47-71).

    The steady-state layout — left rows sharded, right side gathered on every rank — makes every rank build the index of the
    WHOLE right side: 106 ms for C5's 5M multipolygons against a 2 ms join.  When the index serves one join only, the cheap
    side to replicate is the points (16 bytes a row): every rank all-gathers the left points, indexes ITS shard of the right
    side (1/N of the build) and joins ALL points against it.  A pair's right row is the shard's row + `right_row_base`; rank r
    holds exactly the pairs whose right row lives on rank r — disjoint sets whose union is the join (a dataframe join does
    not order its rows: spatial_index.rs:74-76 takes candidates in R-tree order).  Per-left-row hit counts are per-shard
    partial counts; `counts_total` sums them over the ranks (one all-reduce) when there is a process group.

    -> {"pairs": (H_r, 2) int32 CUDA tensor of GLOBAL (l, r), "counts": partial counts of this shard (n_total,),
        "counts_total": summed over ranks, "left_lens": shard lengths, "ms": {"gather", "index", "join"}}"""
    import time

    from . import _abi
    from .geoarrow import DeviceGeoArray
    from .spatial_index import SpatialIndex

    dev = xy_local.device
    t0 = time.perf_counter()
    xy_all, lens = all_gather_points(xy_local, group)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    index = SpatialIndex.from_device(right_shard, stream=stream, light=True)  # (serves one join: no per-entry records of list cells)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    n = int(xy_all.shape[0])
    pts = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy_all, stream=stream)
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    cap = int(pair_capacity) if pair_capacity is not None else max(n, 1024)
    pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    import ctypes as C

    from ._abi import MEM_DEVICE, PREDICATES

    def run(buf) -> tuple[int, int]:
        n_pairs = C.c_int64(0)
        rc = _abi.lib().gpk_spatial_join(pts.handle, right_shard.handle, index.handle, PREDICATES[predicate], 0, counts.data_ptr(), buf.data_ptr(), buf.shape[0],
                                         C.byref(n_pairs), MEM_DEVICE, stream)
        return rc, int(n_pairs.value)

    retry_ms = 0.0
    try:
        rc, h = run(pairs)
        if rc == _abi.GPK_ERR_CAPACITY and h > cap:  # (the ABI reports the exact total: once more with room for it — timed on its own)
            torch.cuda.synchronize(dev)
            tr = time.perf_counter()
            pairs = torch.empty((h, 2), dtype=torch.int32, device=dev)
            rc, h = run(pairs)
            torch.cuda.synchronize(dev)
            retry_ms = (time.perf_counter() - tr) * 1e3
        _abi.check(rc)
        pairs = pairs[:h]
        if right_row_base:
            pairs[:, 1] += int(right_row_base)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
    finally:
        index.free()  # (also when the join raised: the shard's index holds device memory)
    total = counts.clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return {"pairs": pairs, "counts": counts, "counts_total": total, "left_lens": lens,
            "ms": {"gather": (t1 - t0) * 1e3, "index": (t2 - t1) * 1e3, "join": (t3 - t2) * 1e3 - retry_ms, "capacity_retry": retry_ms}}


class Comm:
    """A communicator of libgeopolars_hip (include/geopolars_hip.h, "multi-GPU"): what a Rust / Polars caller of the C ABI uses
    where this module's torch.distributed helpers serve the Python mirror.  `Comm.from_torch()` draws the unique id on rank 0
    and hands it to the other ranks over the already initialised torch.distributed group (any side channel would do)."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes as C

        from . import _abi

        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _abi.check(_abi.lib().gpk_comm_init(rank, world, buf, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def mock(rank: int, world_handle, world: int) -> "Comm":
        """a communicator of the in-process test transport (gpk_comm_mock_world / gpk_comm_init_mock): threads for ranks, no RCCL"""
        import ctypes as C

        from . import _abi

        self = Comm.__new__(Comm)
        h = C.c_void_p()
        _abi.check(_abi.lib().gpk_comm_init_mock(rank, world_handle, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world
        return self

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _abi

        buf = (C.c_uint8 * 128)()
        _abi.check(_abi.lib().gpk_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def from_torch(device: torch.device, group=None) -> "Comm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        t = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            t = torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8).to(device)
        dist.broadcast(t, 0, group=group)
        return Comm(rank, world, bytes(t.cpu().numpy().tobytes()))

    @property
    def handle(self):
        return self._h

    def all_gatherv(self, shard, stream: int = 0):
        """DeviceGeoArray shard -> (the whole column as a DeviceGeoArray that owns its buffers, first row of this rank's shard,
        bytes gathered)"""
        import ctypes as C

        from . import _abi
        from .geoarrow import DeviceGeoArray

        out, base, nbytes = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _abi.check(_abi.lib().gpk_allgatherv_geoarray(self._h, shard.handle, stream, C.byref(out), C.byref(base), C.byref(nbytes)))
        n = C.c_int64(0)
        _abi.check(_abi.lib().gpk_geoarray_len(out, C.byref(n)))
        return DeviceGeoArray(out.value, shard.geom_type, int(n.value), -1), int(base.value), int(nbytes.value)

    def all_gather_rows(self, local: torch.Tensor, stream: int = 0) -> torch.Tensor:
        """(n_local, width) float64 CUDA tensor of every rank -> (n_total, width) in rank order (the leaves of the right side's
        index: gpk_bounds of each shard)"""
        import ctypes as C

        from . import _abi

        local = local.contiguous()
        width = int(local.shape[1]) if local.dim() == 2 else 1
        total = C.c_int64(0)
        counts = (C.c_int64 * self.world)()
        lib = _abi.lib()
        _abi.check(lib.gpk_allgatherv_rows_f64(self._h, local.data_ptr(), int(local.shape[0]), width, None, 0, C.byref(total), counts, stream))
        out = torch.empty((int(total.value), width), dtype=torch.float64, device=local.device)
        _abi.check(lib.gpk_allgatherv_rows_f64(self._h, local.data_ptr(), int(local.shape[0]), width, out.data_ptr(), int(total.value), C.byref(total), counts, stream))
        return out

    def free(self) -> None:
        from . import _abi

        if self._h:
            _abi.lib().gpk_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
