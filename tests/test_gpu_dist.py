"""GPU: the N-rank path of SURVEY.md section 8e on the one GPU a test box has — the RCCL ("nccl") process group at world size 1
(every collective degenerates, but the exact code path of bench.py --config c4 / c5 --gpus N runs: device-resident all-gatherv
of the right side, the leaves exchange, the index assembled from gathered leaves), and K-shard equivalence of the HIP join
(K = 2, 4, 8 row ranges of the left side with left_row_base, concatenated == unsharded).  No 2/4/8-GPU number exists: the
build environment hands out one GPU per call; the driver's SCALE run is the only multi-GPU measurement."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.dist import GeoBuffers, all_gather_leaves, all_gatherv_buffers, shard_rows, slice_rows
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs, join_pairs_device

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1(gpk):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_device_resident_exchange_and_index_from_gathered_leaves(nccl_world1):
    """what bench.py --config c4 does on every rank, at world 1 over RCCL: buffers stay in HBM end to end"""
    import torch

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    left = synth.clustered_polygons(60_000, seed=41, mean_neighbours=4.0)
    right = synth.clustered_polygons(60_000, seed=42, mean_neighbours=4.0)
    shard = GeoBuffers.from_host(right, dev)
    shard_arr = shard.to_device_geoarray(stream)
    box = torch.empty((len(right), 4), dtype=torch.float64, device=dev)
    _abi.check(_abi.lib().gpk_bounds(shard_arr.handle, box.data_ptr(), _abi.MEM_DEVICE, stream))
    stats = {}
    gathered = all_gatherv_buffers(shard, stats=stats)
    leaves = all_gather_leaves(box)
    assert gathered.xy.is_cuda and gathered.ring_offsets.is_cuda and leaves.is_cuda  # nothing was staged through the host
    assert stats["gathered_bytes"] == right.xy.nbytes + right.geom_offsets.nbytes + right.ring_offsets.nbytes
    assert torch.equal(gathered.xy, shard.xy) and torch.equal(gathered.ring_offsets, shard.ring_offsets) and torch.equal(leaves, box)
    rdev = gathered.to_device_geoarray(stream)
    index = SpatialIndex.from_device(rdev, stream=stream, for_points=False, bboxes=leaves)
    ldev = GeoBuffers.from_host(left, dev).to_device_geoarray(stream)
    counts = torch.empty(len(left), dtype=torch.int32, device=dev)
    pairs = torch.empty((8 * len(left), 2), dtype=torch.int32, device=dev)
    h = join_pairs_device(ldev, rdev, index, "intersects", counts, pairs, stream=stream)
    exp_pairs, exp_counts = join_pairs(GeoSeries(left), GeoSeries(right), "intersects")
    assert h == len(exp_pairs)
    assert np.array_equal(pairs[:h].cpu().numpy().astype(np.uint32), exp_pairs)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32), exp_counts)


def test_exchange_with_nulls_on_the_device(nccl_world1):
    import torch

    dev = torch.device("cuda", 0)
    a = synth.powerlaw_multipolygons(500)
    keep = np.ones(len(a), dtype=bool)
    keep[::7] = False
    a.validity = np.packbits(keep, bitorder="little")
    got = all_gatherv_buffers(GeoBuffers.from_host(a, dev)).to_host()
    assert np.array_equal(got.is_valid(), keep) and np.array_equal(got.xy, a.xy) and np.array_equal(got.part_offsets, a.part_offsets)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_row_sharded_hip_joins_equal_unsharded(gpk, k):
    """outputs of row shards are disjoint: concat(HIP shard results with left_row_base) == the unsharded HIP join — for the
    point join (C2 / C5 arms) and for the polygon join (C4), shards balanced by vertex weight"""
    polys = synth.star_polygons(300, 24)
    pts = synth.uniform_points(200_000, seed=31)
    ps, qs = GeoSeries(pts), GeoSeries(polys)
    idx = SpatialIndex(qs)
    full_pairs, full_counts = join_pairs(ps, qs, "intersects", r_index=idx)
    parts, counts = [], []
    for r in range(k):
        lo, hi = shard_rows(len(pts), k, r)
        p, c = join_pairs(GeoSeries(slice_rows(pts, lo, hi)), qs, "intersects", r_index=idx, left_row_base=lo)
        parts.append(p)
        counts.append(c)
    assert np.array_equal(np.concatenate(parts), full_pairs) and np.array_equal(np.concatenate(counts), full_counts)
    left = synth.clustered_polygons(20_000, seed=5, mean_neighbours=6.0)
    right = synth.clustered_polygons(20_000, seed=6, mean_neighbours=6.0)
    rs = GeoSeries(right)
    ridx = SpatialIndex(rs, for_points=False)
    full_pairs, full_counts = join_pairs(GeoSeries(left), rs, "intersects", r_index=ridx)
    w = np.diff(left.ring_offsets)[left.geom_offsets[:-1]]
    parts, counts = [], []
    for r in range(k):
        lo, hi = shard_rows(len(left), k, r, weights=w)
        p, c = join_pairs(GeoSeries(slice_rows(left, lo, hi)), rs, "intersects", r_index=ridx, left_row_base=lo)
        parts.append(p)
        counts.append(c)
    assert np.array_equal(np.concatenate(parts), full_pairs) and np.array_equal(np.concatenate(counts), full_counts)


@pytest.mark.parametrize(
    "config,extra",
    [
        ("c2", ["--points", "2000000"]),  # the configuration the driver's scaling run launches with --gpus N
        ("c3", ["--points", "1000000", "--lines", "20000"]),
        ("c4", ["--polygons", "150000"]),
        ("c4", ["--polygons", "150000", "--comm", "abi"]),  # the same exchange through gpk_allgatherv_* (RCCL opened by the library)
        ("c5", ["--multipolygons", "160000", "--points", "400000"]),
    ],
)
def test_bench_n_rank_path_at_world_1(gpk, config, extra):
    """bench.py --force-dist: RCCL initialised, the right side exchanged, parity-gated line printed"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity-rows", "20000"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["roofline"]["launch_ms"] > 0
    assert line["parity"].get("bit_exact", True) and line["parity"].get("max_rel_err", 0.0) <= 1e-9
    if config in ("c4", "c5"):
        assert line["config"]["right_side_exchange"]["bytes"] > 0


def test_bench_falls_back_to_the_torch_exchange_when_the_librarys_communicator_does_not_come_up(gpk):
    """--comm abi is the default exchange of C4; if gpk_comm_init / the probe collective fail or do not return within --comm-timeout on ANY
    rank, every rank agrees (over the torch group) to run geopolars_amd.dist's exchange instead and the line says which one ran"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), GPK_BENCH_FAIL_COMM="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity-rows", "20000",
           "--polygons", "150000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    ex = line["config"]["right_side_exchange"]
    assert ex["through"].startswith("torch.distributed") and "fallback" in ex["through"] and ex["bytes"] > 0 and line["parity"]["bit_exact"]
    assert "falling back to torch.distributed" in r.stderr


@pytest.mark.parametrize("shard", ["0/8", "5/8", "7/8", "2/3"])
def test_bench_c2_strong_scaled_shards(gpk, shard):
    """the headline at N > 1 is STRONG scaled — rank r owns rows [r n / W, (r + 1) n / W) of the fixed 10M points and its pairs carry
    that base; --as-shard r/W runs one rank's share in this process (one GPU here), parity-gated like any line"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--as-shard", shard, "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-default-shape", "--parity-rows", "50000"]
    if shard == "5/8":  # (one case also walks the weak-scaled leg that runs beside the headline at N > 1)
        cmd.append("--force-weak")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    k, w = (int(v) for v in shard.split("/"))
    lo, hi = k * 10_000_000 // w, (k + 1) * 10_000_000 // w
    assert line["scaling"] == "strong" and line["config"]["rows_of_this_rank"] == [lo, hi] and line["config"]["points_per_gpu"] == hi - lo
    assert line["parity"]["bit_exact"] and line["parity"]["pairs_checked"] > 10_000
    if shard == "5/8":
        weak = line["config"]["weak_scaling"]
        assert weak["points_per_gpu"] == 10_000_000 and weak["value"] > 0 and weak["ms_per_step"] > line["ms_per_step"]


def test_bench_launches_its_own_ranks(gpk):
    """`python bench.py --gpus N` with no launcher re-runs itself under torch.distributed.run (exercised here at N = 1 with
    --spawn: the same code path, RCCL initialised by the spawned rank); the line carries every rank's own step time"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--points", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--parity-rows", "20000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching 1 ranks under torch.distributed.run" in r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["ranks"]["world"] == 1 and line["ranks"]["backend"] == "nccl (RCCL)"
    assert len(line["ranks"]["ms_per_step_per_rank"]) == 1 and line["parity"]["bit_exact"]


def _roundtrip(comm, host):
    from geopolars_amd.geoarrow import DeviceGeoArray

    shard = DeviceGeoArray.upload(host)
    full, base, nbytes = comm.all_gatherv(shard)
    got = full.download()
    assert base == 0 and nbytes > 0 or len(host) == 0
    assert got.geom_type == host.geom_type and len(got) == len(host)
    assert np.array_equal(got.xy, host.xy)
    for name in ("geom_offsets", "part_offsets", "ring_offsets"):
        a, b = getattr(got, name), getattr(host, name)
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), name
    if host.validity is None:
        assert got.validity is None or got.is_valid().all()
    else:
        assert np.array_equal(got.is_valid(), host.is_valid())
    return full


def test_c_abi_communicator_at_world_1(gpk, oracle):
    """gpk_comm_* / gpk_allgatherv_*: RCCL opened by the library itself (no torch.distributed), a one-rank communicator; the
    gathered column equals the shard for every GeoArrow nesting, nulls included, and serves a join like the original"""
    import torch

    from geopolars_amd.dist import Comm
    from geopolars_amd.geoarrow import DeviceGeoArray

    comm = Comm(0, 1, Comm.unique_id())
    rng = np.random.default_rng(3)
    pts = synth.uniform_points(10_001)
    keep = rng.uniform(size=len(pts)) > 0.1
    pts_nulls = GeoArrowArray.from_points(pts.xy, validity=np.packbits(keep, bitorder="little"))
    polys = synth.clustered_polygons(3_000, seed=7)
    mps = synth.powerlaw_multipolygons(500, seed=8)
    lines = synth.random_linestrings(700, seed=9)
    for host in (pts, pts_nulls, polys, mps, lines, GeoArrowArray.from_polygons([])):
        _roundtrip(comm, host)
    # the leaves: boxes of the shard, gathered; the gathered index serves the join like one built from scratch
    right = synth.star_polygons(300, 24)
    shard = DeviceGeoArray.upload(right)
    full, _, _ = comm.all_gatherv(shard)
    boxes = torch.from_numpy(GeoSeries(right).bounds()).cuda()
    leaves = comm.all_gather_rows(boxes)
    assert torch.equal(leaves, boxes)
    index = SpatialIndex.from_device(full, bboxes=leaves)
    left = synth.uniform_points(50_000, seed=4)
    exp_pairs, exp_counts, _ = oracle.spatial_join(left, right, "intersects", mode=0)
    dl = DeviceGeoArray.upload(left)
    counts = torch.empty(len(left), dtype=torch.int32, device="cuda")
    pairs = torch.empty((len(left), 2), dtype=torch.int32, device="cuda")
    from geopolars_amd.spatial_index import join_pairs_device

    h = join_pairs_device(dl, full, index, "intersects", counts, pairs)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32), exp_counts)
    assert np.array_equal(pairs[:h].cpu().numpy().astype(np.uint32), exp_pairs)
    comm.free()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_c_abi_all_gatherv_between_ranks_that_are_apart_in_time(gpk, world):
    """gpk_allgatherv_geoarray / gpk_allgatherv_rows_f64 with MORE THAN ONE RANK: `world` threads of this process, each with its own
    communicator on the library's in-process transport (gpk_comm_init_mock: RCCL's signatures, rendezvous + device copies) and its own
    HIP stream, run the exchange as ranks do — header all-gather, agreement, grouped broadcasts, placement (shard k > 0 drops its leading
    offset), rebase by the children before, validity repack across byte boundaries — for every nesting, shards of odd and zero length,
    nulls on some ranks only, and views whose offsets do not start at 0; every rank must end with the whole column"""
    import ctypes as C
    import threading

    import torch

    from geopolars_amd.dist import Comm, slice_rows
    from geopolars_amd.geoarrow import DeviceGeoArray

    lib = _abi.lib()
    wh = C.c_void_p()
    _abi.check(lib.gpk_comm_mock_world(world, C.byref(wh)))
    rng = np.random.default_rng(40 + world)
    pts = synth.uniform_points(10_007, seed=3)
    cases = {
        "points": pts,
        "points, nulls": GeoArrowArray.from_points(pts.xy, validity=np.packbits(rng.uniform(size=len(pts)) > 0.3, bitorder="little")),
        "linestrings": synth.random_linestrings(1_203, seed=9, max_log2=5.0),
        "polygons": synth.clustered_polygons(3_001, seed=7),
        "multipolygons": synth.powerlaw_multipolygons(997, seed=8, cap=300),
    }
    errors, results = [], {}

    def rank_main(r: int):
        try:
            stream = torch.cuda.Stream()
            comm = Comm.mock(r, wh, world)
            for name, host in cases.items():
                n = len(host)
                cuts = sorted(set([0, n] + list(np.sort(rng_cuts[name]))))  # (the same cuts on every rank)
                cuts = (cuts + [n] * (world + 1))[: world + 1]
                cuts[-1] = n
                lo, hi = cuts[r], cuts[r + 1]
                piece = slice_rows(host, lo, hi)
                if name == "points, nulls" and r % 2 == 1:  # nulls on some ranks only: the others' rows are all valid
                    piece = GeoArrowArray.from_points(piece.xy)
                shard = DeviceGeoArray.upload(piece, stream=stream.cuda_stream)
                full, base, nbytes = comm.all_gatherv(shard, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                results[(name, r)] = (full.download(), base, lo)
                box = torch.full((hi - lo, 4), float(r), dtype=torch.float64, device="cuda")
                leaves = comm.all_gather_rows(box, stream=stream.cuda_stream)
                results[(name + " leaves", r)] = leaves.cpu().numpy()
                full.free()
                shard.free()
            comm.free()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    rng_cuts = {name: rng.integers(0, len(h) + 1, world - 1) for name, h in cases.items()}
    if world == 3:  # one empty shard, one of a single row
        for name, h in cases.items():
            rng_cuts[name] = np.array([1, 1])
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck inside the exchange"
    assert not errors, errors
    for name, host in cases.items():
        for r in range(world):
            got, base, lo = results[(name, r)]
            assert base == lo and got.geom_type == host.geom_type and len(got) == len(host), (name, r)
            assert np.array_equal(got.xy, host.xy), (name, r)
            for f in ("geom_offsets", "part_offsets", "ring_offsets"):
                a, b = getattr(got, f), getattr(host, f)
                assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), (name, r, f)
            if name == "points, nulls":
                cuts = sorted(set([0, len(host)] + list(np.sort(rng_cuts[name]))))
                cuts = (cuts + [len(host)] * (world + 1))[: world + 1]
                cuts[-1] = len(host)
                want = host.is_valid().copy()
                for k in range(world):
                    if k % 2 == 1:
                        want[cuts[k] : cuts[k + 1]] = True
                assert np.array_equal(got.is_valid(), want), (name, r)
            leaves = results[(name + " leaves", r)]
            assert leaves.shape == (len(host), 4) and np.all(np.diff(leaves[:, 0]) >= 0) and set(np.unique(leaves)) <= set(float(k) for k in range(world))
    _abi.check(lib.gpk_comm_mock_world_free(wh))


@pytest.mark.parametrize("k", [2, 4, 8])
def test_right_partitioned_one_shot_join_equals_the_whole_join(gpk, oracle, k):
    """dist.join_partition_right with K shards of the RIGHT side on one GPU (what ranks 0 .. K - 1 would each do): the shards'
    pair sets are disjoint, their union is the join, their partial hit counts add up (spatial_index.rs:47-71: the index of
    one shard is 1 / K of the build)"""
    import torch

    from geopolars_amd.dist import join_partition_right
    from geopolars_amd.geoarrow import DeviceGeoArray

    right = synth.powerlaw_multipolygons(6_000, seed=21)
    left = synth.uniform_points(80_001, seed=22)
    ep, ec, _ = oracle.spatial_join(left, right, "within", mode=0)
    xy = torch.from_numpy(left.xy).cuda()
    w = np.diff(right.geom_offsets)
    got_pairs, got_counts = [], np.zeros(len(left), dtype=np.int64)
    for r in range(k):
        lo, hi = shard_rows(len(right), k, r, weights=w)
        shard = DeviceGeoArray.upload(slice_rows(right, lo, hi))
        res = join_partition_right(xy, shard, lo, "within", pair_capacity=1024)  # (a buffer too small on purpose: the call sizes it from the reported total)
        p = res["pairs"].cpu().numpy().astype(np.uint32)
        assert len(p) == 0 or (p[:, 1].min() >= lo and p[:, 1].max() < hi)
        got_pairs.append(p)
        got_counts += res["counts"].cpu().numpy().astype(np.int64)
        assert set(res["ms"]) == {"gather", "index", "join", "capacity_retry"} and res["ms"]["capacity_retry"] > 0.0  # (the retry is timed on its own)
    allp = np.concatenate(got_pairs)
    allp = allp[np.lexsort((allp[:, 1], allp[:, 0]))]
    assert np.array_equal(allp, ep)
    assert np.array_equal(got_counts.astype(np.uint32), ec)
