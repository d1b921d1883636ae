"""world_size-2 gloo tests of the N>1 path: row sharding, the right-side all-gatherv
with offset rebasing, the broadcast of a replicated right side, and shard/concat equivalence of the
join on the CPU oracle (K = 2, 4, 8 shards == unsharded)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from geopolars_amd import synth
from geopolars_amd.dist import shard_rows, slice_rows
from geopolars_amd.geoarrow import GeoArrowArray


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _same(a: GeoArrowArray, b: GeoArrowArray) -> bool:
    if a.geom_type != b.geom_type or not np.array_equal(a.xy, b.xy):
        return False
    for k in ("geom_offsets", "part_offsets", "ring_offsets"):
        x, y = getattr(a, k), getattr(b, k)
        if (x is None) != (y is None) or (x is not None and not np.array_equal(x, y)):
            return False
    return True


def _worker(rank: int, world: int, port: int, kind: str, q):
    import torch.distributed as dist

    import torch

    from geopolars_amd.dist import all_gather_leaves, all_gatherv_geoarray, broadcast_geoarray

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nulls = kind.endswith("+nulls")
        full = {"multipoly": lambda: synth.powerlaw_multipolygons(301), "poly": lambda: synth.star_polygons(57, 9), "lines": lambda: synth.random_linestrings(40), "points": lambda: synth.uniform_points(33)}[kind.split("+")[0]]()
        w = np.diff(full.geom_offsets) if full.geom_offsets is not None else None
        lo, hi = shard_rows(len(full), world, rank, weights=w)
        if nulls:  # nulls in rank 0's rows only: rank 1's shard arrives WITHOUT a bitmap (from_arrow_wkb sets one only when a shard has nulls)
            bits = np.ones(len(full), dtype=np.uint8)
            bits[: max(1, shard_rows(len(full), world, 0, weights=w)[1]) : 3] = 0
            full.validity = np.packbits(bits, bitorder="little")
        local = slice_rows(full, lo, hi)
        if nulls and rank != 0:
            assert local.is_valid().all()
            local.validity = None
        got = all_gatherv_geoarray(local)
        ok = _same(got, full)
        if nulls:
            ok = ok and got.validity is not None and np.array_equal(got.is_valid(), full.is_valid())
        rep = broadcast_geoarray(full if rank == 0 else None, 0)
        ok = ok and _same(rep, full)
        if nulls:
            ok = ok and rep.validity is not None and np.array_equal(rep.is_valid(), full.is_valid())
        # the leaves each rank built for its shard, gathered in rank order == the boxes of the whole column
        box = np.arange(4 * (hi - lo), dtype=np.float64).reshape(-1, 4) + 1000.0 * rank
        leaves = all_gather_leaves(torch.from_numpy(box)).numpy()
        ok = ok and leaves.shape == (len(full), 4) and np.array_equal(leaves[lo:hi], box)
        # the points of a right-partitioned one-shot join (dist.join_partition_right): shards of unequal length, in rank order
        from geopolars_amd.dist import all_gather_points

        mine = np.arange(2 * (5 + 3 * rank), dtype=np.float64).reshape(-1, 2) + 100.0 * rank
        allp, lens = all_gather_points(torch.from_numpy(mine))
        exp = np.concatenate([np.arange(2 * (5 + 3 * r), dtype=np.float64).reshape(-1, 2) + 100.0 * r for r in range(world)])
        ok = ok and lens == [5 + 3 * r for r in range(world)] and np.array_equal(allp.numpy(), exp)
        q.put((rank, ok, lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["multipoly", "poly", "lines", "points", "poly+nulls", "multipoly+nulls"])
def test_all_gatherv_and_broadcast_world2(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0]


def test_shard_rows_balances_weight():
    w = np.array([1] * 90 + [1000] * 10)
    cuts = [shard_rows(100, 4, r, w) for r in range(4)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 100
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    loads = [w[lo:hi].sum() for lo, hi in cuts]
    assert max(loads) <= 2 * (w.sum() / 4) + 1000
    assert [shard_rows(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


@pytest.mark.parametrize("k", [2, 4, 8])
def test_row_sharded_join_equals_unsharded(oracle, k):
    """outputs of row shards are disjoint: concat(shard results with left_row_base) == unsharded."""
    polys = synth.star_polygons(120, 16)
    pts = synth.uniform_points(20_000)
    full_pairs, full_counts, _ = oracle.spatial_join(pts, polys, "intersects", mode=1)
    parts, counts = [], []
    for r in range(k):
        lo, hi = shard_rows(len(pts), k, r)
        p, c, _ = oracle.spatial_join(slice_rows(pts, lo, hi), polys, "intersects", mode=1)
        p = p.copy()
        p[:, 0] += lo
        parts.append(p)
        counts.append(c)
    assert np.array_equal(np.concatenate(parts), full_pairs)
    assert np.array_equal(np.concatenate(counts), full_counts)


@pytest.mark.parametrize("pred", ["intersects", "contains"])
def test_row_sharded_polygon_join_equals_unsharded(oracle, pred):
    """the same property with a polygonal left side (C4: shards cut the left column by rows, offsets are rebased)"""
    left = synth.clustered_polygons(3000, seed=5, mean_neighbours=6.0)
    right = synth.clustered_polygons(20_000, seed=6, mean_neighbours=0.2, min_verts=4, max_verts=10)
    full_pairs, full_counts, _ = oracle.spatial_join(left, right, pred, mode=1)
    assert len(full_pairs) > 500
    parts, counts = [], []
    for r in range(4):
        lo, hi = shard_rows(len(left), 4, r)
        p, c, _ = oracle.spatial_join(slice_rows(left, lo, hi), right, pred, mode=1)
        p = p.copy()
        p[:, 0] += lo
        parts.append(p)
        counts.append(c)
    assert np.array_equal(np.concatenate(parts), full_pairs)
    assert np.array_equal(np.concatenate(counts), full_counts)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_right_partitioned_join_equals_unsharded(oracle, k):
    """dist.join_partition_right's decomposition (the one-shot layout: every rank indexes ITS shard of the right side and joins ALL
    left rows): the ranks' pair lists, right ids offset by the shard's first row, are disjoint and their union — re-ordered by
    (left, right) — is the unsharded join; per-row counts add up.  (The oracle is the join engine here; the GPU form of the same
    statement is tests/test_gpu_dist.py.)"""
    right = synth.powerlaw_multipolygons(600, seed=8)
    pts = synth.uniform_points(30_000, seed=9)
    full_pairs, full_counts, _ = oracle.spatial_join(pts, right, "within", mode=1)
    assert len(full_pairs) > 100
    parts, counts = [], np.zeros(len(pts), np.int64)
    for r in range(k):
        lo, hi = shard_rows(len(right), k, r)
        p, c, _ = oracle.spatial_join(pts, slice_rows(right, lo, hi), "within", mode=1)
        p = p.copy()
        p[:, 1] += lo
        parts.append(p)
        counts += c
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged[:, 1], merged[:, 0]))]
    assert np.array_equal(merged, full_pairs)
    assert np.array_equal(counts, full_counts.astype(np.int64))
