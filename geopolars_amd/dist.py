"""Row sharding and the one collective of the path.

Every operator is a per-row map (geoseries.rs:141 "1-to-1 row-wise"; the join refine is independent
per candidate pair, spatial_index.rs:83-143), so N GPUs = N independent row ranges of the LEFT
series, one process per GPU.  The right side is either replicated (C2: 1 MB of polygons ->
`broadcast_geoarray`) or, when it is itself produced sharded (C4/C5), exchanged ONCE with an
all-gatherv of its GeoArrow buffers over RCCL/xGMI (`all_gatherv_geoarray`): RCCL has no native
`v` collective, so buffers are padded to the longest shard, all-gathered, trimmed and the offsets
rebased.  No collective touches results: output row ranges are disjoint and pairs carry a per-shard
`left_row_base`.

Works on CPU tensors with the gloo backend (tests, world_size 2) and CUDA tensors with nccl (= RCCL).
"""
from __future__ import annotations

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


def _levels(a_type: int, geom_offsets, part_offsets, ring_offsets):
    """offset buffers from the outermost level inwards"""
    return [o for o in (geom_offsets, part_offsets, ring_offsets) if o is not None]


def _gatherv(t: torch.Tensor, group=None) -> list[torch.Tensor]:
    """all-gather of variable-length 1-D/2-D tensors: exchange lengths, pad to the max, gather, trim."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    m = max(lens)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)  # one fixed-size collective per buffer
    return [o[:k] for o, k in zip(out, lens)]


def all_gatherv_geoarray(local: GeoArrowArray, device: Optional[torch.device] = None, group=None) -> GeoArrowArray:
    """Every rank contributes its shard of the RIGHT side and receives the concatenation in rank order
    (the north star's "RCCL all-gatherv ... to broadcast the right-side R-tree leaves").  Offsets of
    shard k are rebased by the child lengths of shards 0..k-1."""
    device = device or torch.device("cpu")
    xy_parts = _gatherv(torch.from_numpy(local.xy).to(device), group)
    xy = torch.cat(xy_parts).cpu().numpy()
    levels_local = _levels(local.geom_type, local.geom_offsets, local.part_offsets, local.ring_offsets)
    levels = []
    for off in levels_local:
        parts = _gatherv(torch.from_numpy(off.astype(np.int32)).to(device), group)
        rebased, base = [], 0
        for k, p in enumerate(parts):
            p = p.cpu().numpy().astype(np.int64)
            rebased.append((p if k == 0 else p[1:]) + base)
            base += int(p[-1])
        levels.append(np.concatenate(rebased).astype(np.int32))
    names = ["geom_offsets", "part_offsets", "ring_offsets"]
    present = [n for n, o in zip(names, (local.geom_offsets, local.part_offsets, local.ring_offsets)) if o is not None]
    kw = dict(zip(present, levels))
    validity = None
    if local.validity is not None or _any_rank_has_validity(local, device, group):
        bits = local.is_valid().astype(np.uint8)
        vparts = _gatherv(torch.from_numpy(bits).to(device), group)
        validity = np.packbits(torch.cat(vparts).cpu().numpy(), bitorder="little")
    return GeoArrowArray(local.geom_type, xy, validity=validity, **kw)


def _any_rank_has_validity(local: GeoArrowArray, device, group) -> bool:
    flag = torch.tensor([1 if local.validity is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item())


def broadcast_geoarray(a: Optional[GeoArrowArray], src: int = 0, device: Optional[torch.device] = None, group=None) -> GeoArrowArray:
    """Replicate a small right side (C2's 1k polygons) from `src` to every rank: header, then each
    buffer with one broadcast."""
    device = device or torch.device("cpu")
    rank = dist.get_rank(group)
    hdr = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == src:
        hdr[:6] = torch.tensor(
            [a.geom_type, a.n_geoms, a.n_coords, a.n_parts if a.part_offsets is not None else -1, a.n_rings if a.ring_offsets is not None else -1, 0 if a.geom_offsets is None else 1]
        )
    dist.broadcast(hdr, src, group=group)
    gt, n_geoms, n_coords, n_parts, n_rings, has_go = (int(v) for v in hdr[:6].tolist())

    def bc(arr, n, dtype):
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device) if rank == src else torch.empty(n, dtype=dtype, device=device)
        dist.broadcast(t, src, group=group)
        return t.cpu().numpy()

    xy = bc(a.xy.reshape(-1) if rank == src else None, 2 * n_coords, torch.float64).reshape(-1, 2)
    go = bc(a.geom_offsets if rank == src else None, n_geoms + 1, torch.int32) if has_go else None
    po = bc(a.part_offsets if rank == src else None, n_parts + 1, torch.int32) if n_parts >= 0 else None
    ro = bc(a.ring_offsets if rank == src else None, n_rings + 1, torch.int32) if n_rings >= 0 else None
    return GeoArrowArray(gt, xy, geom_offsets=go, part_offsets=po, ring_offsets=ro, n_geoms=n_geoms)
