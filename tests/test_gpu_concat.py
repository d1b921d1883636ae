"""The ASSEMBLE half of the all-gatherv (csrc/gpk_comm.hip: assemble_column) with K > 1 shards on ONE GPU, through
gpk_geoarray_concat — Arrow's rechunk (py-geopolars/src/ffi.rs:56,73,93) on the device.  What a K-rank
gpk_allgatherv_geoarray does after the lengths are known — where piece k lands, the leading offset entry every shard but the
first drops, the children of the shards before it (minus the shard's own first offset) added to its offsets, validity bits
repacked across shard boundaries that do not fall on a byte — is the same code, fed by device copies instead of broadcasts
(the seam: spatial_index.rs:37-76).  Every GeoArrow nesting, K = 2 / 4 / 8 with odd and empty shards, nulls on some shards
only, shards that are SLICES of a larger column (offsets not starting at 0), and the gathered column serving a join."""
import numpy as np
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray, GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs

pytestmark = pytest.mark.gpu


def _multipoints(n, seed):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(0, 5, n)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(cnt)
    return GeoArrowArray(_abi.GEOM_MULTIPOINT, rng.uniform(0, 100, (int(off[-1]), 2)), geom_offsets=off)


def _multilinestrings(n, seed):
    rng = np.random.default_rng(seed)
    parts = rng.integers(0, 4, n)
    go = np.zeros(n + 1, np.int32)
    go[1:] = np.cumsum(parts)
    lens = rng.integers(2, 7, int(go[-1]))
    ro = np.zeros(len(lens) + 1, np.int32)
    ro[1:] = np.cumsum(lens)
    return GeoArrowArray(_abi.GEOM_MULTILINESTRING, rng.uniform(0, 100, (int(ro[-1]), 2)), geom_offsets=go, ring_offsets=ro)


def _columns():
    rng = np.random.default_rng(11)
    pts = synth.uniform_points(1003, seed=2)
    return {
        "point": pts,
        "point+nulls": GeoArrowArray.from_points(pts.xy, validity=np.packbits(rng.uniform(size=len(pts)) > 0.2, bitorder="little")),
        "linestring": synth.random_linestrings(517, seed=3),
        "polygon": synth.clustered_polygons(701, seed=4),
        "multipoint": _multipoints(333, 5),
        "multilinestring": _multilinestrings(411, 6),
        "multipolygon": synth.powerlaw_multipolygons(257, seed=7),
    }


def _cuts(n, k, seed):
    """k shards of n rows: odd lengths (so validity bytes straddle the cuts), some of them empty"""
    rng = np.random.default_rng(seed)
    c = np.sort(rng.integers(0, n + 1, k - 1))
    if k >= 4:
        c[1] = c[0]  # an empty shard in the middle
    if k >= 8:
        c[-1] = n  # ... and one at the end
    return np.concatenate([[0], c, [n]]).astype(np.int64)


def _same(a: GeoArrowArray, b: GeoArrowArray):
    assert a.geom_type == b.geom_type and len(a) == len(b)
    assert np.array_equal(a.xy, b.xy)
    for name in ("geom_offsets", "part_offsets", "ring_offsets"):
        x, y = getattr(a, name), getattr(b, name)
        assert (x is None) == (y is None), name
        if x is not None:
            assert np.array_equal(x, y), name
    assert np.array_equal(a.is_valid(), b.is_valid())


@pytest.mark.parametrize("k", [2, 4, 8])
def test_k_shards_assemble_to_the_unsharded_column(gpk, k):
    for name, host in _columns().items():
        cuts = _cuts(len(host), k, seed=k + len(name))
        shards_host = [host.take(np.arange(cuts[i], cuts[i + 1])) for i in range(k)]
        if host.validity is not None:  # nulls on some shards only: a shard without a bitmap is all valid
            shards_host = [GeoArrowArray(s.geom_type, s.xy, s.geom_offsets, s.part_offsets, s.ring_offsets, None if s.is_valid().all() else s.validity, n_geoms=len(s)) for s in shards_host]
        shards = [DeviceGeoArray.upload(s) for s in shards_host]
        full, bases = DeviceGeoArray.concat(shards)
        assert np.array_equal(bases, cuts), name
        _same(full.download(), host)


def test_shards_that_are_slices_of_one_device_column(gpk):
    """zero-copy device views whose offsets do NOT start at 0: a sliced Arrow list array hands over its offsets unrebased, next
    to the slice of the child buffer they index into — the assembly subtracts every shard's own first offset"""
    import torch

    dev = torch.device("cuda", 0)
    for name in ("linestring", "polygon", "multilinestring", "multipolygon"):
        host = _columns()[name]
        names = [n for n in ("geom_offsets", "part_offsets", "ring_offsets") if getattr(host, n) is not None]
        off_host = [getattr(host, n) for n in names]
        off_dev = [torch.from_numpy(o).to(dev) for o in off_host]
        xy = torch.from_numpy(host.xy).to(dev)
        cuts = _cuts(len(host), 4, seed=5)
        shards = []
        for i in range(4):
            lo, hi = int(cuts[i]), int(cuts[i + 1])
            kw = {}
            for nm, oh, od in zip(names, off_host, off_dev):
                kw[nm] = od[lo : hi + 1]  # not rebased: the first entry is the slice's first child in the WHOLE child buffer
                lo, hi = int(oh[lo]), int(oh[hi])
            shards.append(DeviceGeoArray.from_device_buffers(host.geom_type, xy[lo:hi], **kw))
        # (round 5: such a view is normalised at upload — every entry point, not only concat, sees children indexed from 0)
        for i in range(4):
            want = host.take(np.arange(int(cuts[i]), int(cuts[i + 1])))
            _same(shards[i].download(), want)
            if len(want):
                from oracle import pyoracle

                pyoracle.build()
                assert np.array_equal(GeoSeries(None, device=shards[i]).bounds(), pyoracle.bounds(want), equal_nan=True)
        full, bases = DeviceGeoArray.concat(shards)
        assert np.array_equal(bases, cuts)
        _same(full.download(), host)


def test_type_mismatch_and_bad_arguments(gpk):
    import ctypes as C

    a = DeviceGeoArray.upload(synth.uniform_points(10))
    b = DeviceGeoArray.upload(synth.star_polygons(3, 8))
    with pytest.raises(_abi.GeopolarsHipError) as e:
        DeviceGeoArray.concat([a, b])
    assert e.value.code == _abi.GPK_ERR_MISMATCHED_GEOMETRY
    out = C.c_void_p()
    assert _abi.lib().gpk_geoarray_concat(None, 0, None, C.byref(out), None, None) == _abi.GPK_ERR_INVALID_ARGUMENT
    one, bases = DeviceGeoArray.concat([b])
    _same(one.download(), synth.star_polygons(3, 8))


def test_the_assembled_right_side_serves_the_join(gpk, oracle):
    right = synth.star_polygons(1000, 64)
    cuts = _cuts(len(right), 8, seed=3)
    shards = [DeviceGeoArray.upload(right.take(np.arange(cuts[i], cuts[i + 1]))) for i in range(8)]
    full, _ = DeviceGeoArray.concat(shards)
    left = synth.uniform_points(100_003, seed=9)
    ep, ec, _ = oracle.spatial_join(left, right, "intersects", mode=0)
    rs = GeoSeries(full.download())
    gp, gc = join_pairs(GeoSeries(left), rs, "intersects", r_index=SpatialIndex(rs))
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
