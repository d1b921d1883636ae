"""GPU tests of the boundary the reference actually has: the Arrow C Data Interface (py-geopolars/src/ffi.rs:12-32 — a pyarrow array
exported with `_export_to_c` into an ArrowArray / ArrowSchema pair) and the coordinate layout its Python layer builds — Struct<x, y>
fields from `pyarrow.StructArray.from_arrays([coords[:, 0], coords[:, 1]], ["x", "y"])` under 0 - 2 `ListArray` levels
(py-geopolars/python/geopolars/internals/geoseries.py:86-113).  gpk_geoarray_from_arrow must land every such column in HBM so that
the operators answer exactly as for the same geometry uploaded as interleaved GeoArrow buffers (compared through download(): bit
equality of coordinates and offsets, and through area / bounds / a join)."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray, GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs

pytestmark = pytest.mark.gpu


def _struct_coords(xy):
    return pa.StructArray.from_arrays([pa.array(xy[:, 0]), pa.array(xy[:, 1])], ["x", "y"])


def _nested(host: GeoArrowArray, coords, large=False):
    """the reference's construction (internals/geoseries.py:86-113), for every nesting"""
    levels = {
        _abi.GEOM_POINT: [],
        _abi.GEOM_LINESTRING: [host.geom_offsets],
        _abi.GEOM_MULTIPOINT: [host.geom_offsets],
        _abi.GEOM_POLYGON: [host.ring_offsets, host.geom_offsets],
        _abi.GEOM_MULTILINESTRING: [host.ring_offsets, host.geom_offsets],
        _abi.GEOM_MULTIPOLYGON: [host.ring_offsets, host.part_offsets, host.geom_offsets],
    }[host.geom_type]
    arr = coords
    for off in levels:
        if large:
            arr = pa.LargeListArray.from_arrays(pa.array(off.astype(np.int64), type=pa.int64()), arr)
        else:
            arr = pa.ListArray.from_arrays(pa.array(off, type=pa.int32()), arr)
    return arr


def _same(dev: DeviceGeoArray, host: GeoArrowArray):
    got = dev.download()
    assert got.geom_type == host.geom_type and len(got) == len(host)
    assert np.array_equal(got.xy, host.xy)
    for name in ("geom_offsets", "part_offsets", "ring_offsets"):
        a, b = getattr(got, name), getattr(host, name)
        assert (a is None) == (b is None), name
        if a is not None:
            assert np.array_equal(a, b), name


CASES = {
    "points": lambda: synth.uniform_points(5000, seed=2),
    "linestrings": lambda: synth.random_linestrings(700, seed=3, max_log2=5.0),
    "polygons": lambda: synth.star_polygons(400, 24),
    "multipolygons": lambda: synth.powerlaw_multipolygons(300, seed=5, cap=200),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("coords", ["struct", "interleaved"])
@pytest.mark.parametrize("large", [False, True])
def test_native_geoarrow_columns_through_the_c_data_interface(gpk, oracle, case, coords, large):
    host = CASES[case]()
    c = _struct_coords(host.xy) if coords == "struct" else pa.FixedSizeListArray.from_arrays(pa.array(host.xy.reshape(-1)), 2)
    col = _nested(host, c, large)
    dev = DeviceGeoArray.from_arrow(col)
    assert dev.geom_type == host.geom_type
    _same(dev, host)
    s = GeoSeries(None, device=dev)
    if host.geom_type != _abi.GEOM_POINT:
        assert np.array_equal(s.bounds(), oracle.bounds(host))
    if host.geom_type in (_abi.GEOM_POLYGON, _abi.GEOM_MULTIPOLYGON):
        np.testing.assert_allclose(s.area(), oracle.area(host), rtol=1e-9, atol=0)


def test_slices_of_every_level_and_chunked_columns(gpk, oracle):
    host = synth.star_polygons(300, 16)
    col = _nested(host, _struct_coords(host.xy))
    for lo, n in ((0, 300), (17, 100), (299, 1), (120, 0)):
        sl = col.slice(lo, n)
        want = host.take(np.arange(lo, lo + n))
        _same(DeviceGeoArray.from_arrow(sl), want)
    # a column in three chunks is rechunked first (ffi.rs:56)
    chunked = pa.chunked_array([col.slice(0, 100), col.slice(100, 150), col.slice(250, 50)])
    _same(DeviceGeoArray.from_arrow(chunked), host)
    # sliced COORDINATE children (a struct array whose fields carry their own offsets)
    pts = synth.uniform_points(1000, seed=9)
    st = _struct_coords(pts.xy).slice(100, 500)
    _same(DeviceGeoArray.from_arrow(st), pts.take(np.arange(100, 600)))
    fl = pa.FixedSizeListArray.from_arrays(pa.array(pts.xy.reshape(-1)), 2).slice(3, 77)
    _same(DeviceGeoArray.from_arrow(fl), pts.take(np.arange(3, 80)))


def test_wkb_columns_binary_and_large_binary_with_nulls(gpk, oracle):
    host = synth.star_polygons(200, 12)
    values, offsets = host.to_wkb()
    rows = [bytes(values[offsets[i] : offsets[i + 1]]) for i in range(len(host))]
    rows[5] = None
    rows[77] = None
    for typ in (pa.binary(), pa.large_binary()):
        col = pa.array(rows, type=typ)
        dev = DeviceGeoArray.from_arrow(col)
        assert dev.geom_type == _abi.GEOM_POLYGON and dev.n_geoms == 200
        got = dev.download()
        valid = got.is_valid()
        assert not valid[5] and not valid[77] and valid.sum() == 198
        area = GeoSeries(None, device=dev).area()
        want = oracle.area(host)
        ok = np.ones(200, dtype=bool)
        ok[[5, 77]] = False
        np.testing.assert_allclose(area[ok], want[ok], rtol=1e-9, atol=0)
        # a slice whose validity bitmap starts in the middle of a byte
        sl = DeviceGeoArray.from_arrow(col.slice(3, 100))
        v = sl.download().is_valid()
        assert not v[2] and not v[74] and v.sum() == 98


def test_multi_types_by_hint_and_by_extension_name(gpk):
    rng = np.random.default_rng(4)
    xy = rng.uniform(0, 100, (60, 2))
    off = np.array([0, 10, 10, 25, 60], dtype=np.int32)
    col = pa.ListArray.from_arrays(pa.array(off), _struct_coords(xy))
    assert DeviceGeoArray.from_arrow(col).geom_type == _abi.GEOM_LINESTRING
    assert DeviceGeoArray.from_arrow(col, geom_type=_abi.GEOM_MULTIPOINT).geom_type == _abi.GEOM_MULTIPOINT
    # the schema's ARROW:extension:name travels in the ArrowSchema's metadata
    field = pa.field("geometry", col.type, metadata={"ARROW:extension:name": "geoarrow.multipoint"})
    lib = _abi.lib()
    c_array, c_schema = _abi.ArrowArray(), _abi.ArrowSchema()
    col._export_to_c(C.addressof(c_array))
    field._export_to_c(C.addressof(c_schema))
    out, gt = C.c_void_p(), C.c_int32(-1)
    try:
        _abi.check(lib.gpk_geoarray_from_arrow(C.addressof(c_array), C.addressof(c_schema), -1, None, C.byref(out), C.byref(gt)))
    finally:
        c_array.release(C.byref(c_array))
        c_schema.release(C.byref(c_schema))
    assert gt.value == _abi.GEOM_MULTIPOINT
    lib.gpk_geoarray_free(out)
    # what is not a geometry column says so
    for bad in (pa.array([1.0, 2.0]), pa.array(["a", "b"]), pa.ListArray.from_arrays(pa.array(off), pa.array(rng.uniform(size=60)))):
        with pytest.raises(_abi.GeopolarsHipError) as e:
            DeviceGeoArray.from_arrow(bad)
        assert e.value.code == _abi.GPK_ERR_MISMATCHED_GEOMETRY
    with pytest.raises(_abi.GeopolarsHipError):
        DeviceGeoArray.from_arrow(col, geom_type=_abi.GEOM_POLYGON)


def test_separated_device_buffers_and_a_join_over_struct_coordinates(gpk, oracle):
    """points handed over as TWO device arrays (x, y): interleaved during the upload, then the headline join"""
    import torch

    pts = synth.uniform_points(200_000, seed=12)
    polys = synth.star_polygons(1000, 64)
    dev = torch.device("cuda", 0)
    x, y = torch.from_numpy(np.ascontiguousarray(pts.xy[:, 0])).to(dev), torch.from_numpy(np.ascontiguousarray(pts.xy[:, 1])).to(dev)
    dpts = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, (x, y))
    _same(dpts, pts)
    right = GeoSeries.from_arrow(_nested(polys, _struct_coords(polys.xy)))
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
    gp, gc = join_pairs(GeoSeries(None, device=dpts), right, "intersects", r_index=SpatialIndex(right))
    assert np.array_equal(gc, ec) and np.array_equal(gp, ep)
    # host x / y arrays through the descriptor, and the host WKB encoder reading them
    d = pts.desc()
    xs, ys = np.ascontiguousarray(pts.xy[:, 0]), np.ascontiguousarray(pts.xy[:, 1])
    d.xy, d.x, d.y = None, xs.ctypes.data, ys.ctypes.data
    out = C.c_void_p()
    _abi.check(_abi.lib().gpk_geoarray_upload(C.byref(d), None, C.byref(out)))
    _same(DeviceGeoArray(out.value, _abi.GEOM_POINT, len(pts), len(pts)), pts)
    nb = C.c_int64(0)
    off = np.zeros(len(pts) + 1, dtype=np.int32)
    _abi.check(_abi.lib().gpk_wkb_encode(C.byref(d), off.ctypes.data, None, 0, C.byref(nb)))
    vals = np.zeros(nb.value, dtype=np.uint8)
    _abi.check(_abi.lib().gpk_wkb_encode(C.byref(d), off.ctypes.data, vals.ctypes.data, len(vals), C.byref(nb)))
    v2, o2 = pts.to_wkb()
    assert np.array_equal(vals, v2) and np.array_equal(off, o2)


def test_spatial_join_over_tables_whose_geometry_columns_are_native_geoarrow(gpk, oracle):
    """the dataframe-shaped call (spatial_index.rs:37-204 over pyarrow tables): geometry columns as the reference's Python layer builds
    them — Struct<x, y> under list levels (internals/geoseries.py:86-113) — or as WKB binary give the same joined table"""
    from geopolars_amd.spatial_index import SpatialJoinArgs, spatial_join

    polys = synth.star_polygons(300, 16)
    pts = synth.uniform_points(20_000, seed=7)
    lt_native = pa.table({"id": pa.array(np.arange(len(pts))), "geometry": _struct_coords(pts.xy)})
    rt_native = pa.table({"name": pa.array([f"p{i}" for i in range(len(polys))]), "geometry": _nested(polys, _struct_coords(polys.xy))})
    lt_wkb = pa.table({"id": lt_native.column("id"), "geometry": pts.to_arrow_wkb()})
    rt_wkb = pa.table({"name": rt_native.column("name"), "geometry": polys.to_arrow_wkb()})
    # (two list levels without an ARROW:extension:name are a Polygon or a MultiLineString column: the join does not guess)
    with pytest.raises(_abi.GeopolarsHipError) as e:
        spatial_join(lt_native, rt_native)
    assert e.value.code == _abi.GPK_ERR_INVALID_ARGUMENT and "r_geom_type" in str(e.value)
    for how in ("inner", "left"):
        a = spatial_join(lt_native, rt_native, SpatialJoinArgs(join_type=how, r_geom_type=_abi.GEOM_POLYGON))
        b = spatial_join(lt_wkb, rt_wkb, SpatialJoinArgs(join_type=how))
        assert a.column_names == b.column_names and a.num_rows == b.num_rows
        assert a.column("id_left").equals(b.column("id_left")) and a.column("name_right").equals(b.column("name_right"))
        # (a native geometry column leaves the join as WKB, like the reference's: the same bytes as the WKB tables' columns)
        assert a.column("geometry_left").equals(b.column("geometry_left")) and a.column("geometry_right").equals(b.column("geometry_right"))
    ep, ec, _ = oracle.spatial_join(pts, polys, "intersects", mode=0)
    # ... or the column's field names its type the GeoArrow way
    rt_named = pa.table([rt_native.column("name"), rt_native.column("geometry")],
                        schema=pa.schema([rt_native.schema.field("name"), pa.field("geometry", rt_native.schema.field("geometry").type, metadata={"ARROW:extension:name": "geoarrow.polygon"})]))
    inner = spatial_join(lt_native, rt_named)
    assert inner.num_rows == len(ep)
    assert np.array_equal(inner.column("id_left").to_numpy(), ep[:, 0]) and inner.column("name_right").to_pylist() == [f"p{j}" for j in ep[:, 1]]


def test_empty_columns_cross_the_c_data_interface(gpk):
    """zero-row geometry columns (a producer may export them with a NULL or 0-byte offsets buffer): an empty array, an empty join"""
    from geopolars_amd.spatial_index import SpatialJoinArgs, spatial_join

    assert DeviceGeoArray.from_arrow(pa.array([], type=pa.binary())).n_geoms == 0
    empty_poly = pa.ListArray.from_arrays(pa.array([0], type=pa.int32()), pa.ListArray.from_arrays(pa.array([0], type=pa.int32()), _struct_coords(np.zeros((0, 2)))))
    assert DeviceGeoArray.from_arrow(empty_poly).n_geoms == 0
    # the structs by hand, offsets buffer NULL
    c_array, c_schema = _abi.ArrowArray(), _abi.ArrowSchema()
    pa.array([], type=pa.binary())._export_to_c(C.addressof(c_array), C.addressof(c_schema))
    bufs = C.cast(c_array.buffers, C.POINTER(C.c_void_p))
    keep = bufs[1]
    bufs[1] = None
    out, gt = C.c_void_p(), C.c_int32(-1)
    try:
        _abi.check(_abi.lib().gpk_geoarray_from_arrow(C.addressof(c_array), C.addressof(c_schema), -1, None, C.byref(out), C.byref(gt)))
    finally:
        bufs[1] = keep
        c_array.release(C.byref(c_array))
        c_schema.release(C.byref(c_schema))
    _abi.lib().gpk_geoarray_free(out)
    polys = synth.star_polygons(50, 8)
    lt = pa.table({"id": pa.array([], type=pa.int64()), "geometry": pa.array([], type=pa.binary())})
    rt = pa.table({"name": pa.array([f"p{i}" for i in range(len(polys))]), "geometry": polys.to_arrow_wkb()})
    assert spatial_join(lt, rt).num_rows == 0
    assert spatial_join(rt.rename_columns(["id", "geometry"]), lt.rename_columns(["name", "geometry"]), SpatialJoinArgs(join_type="left")).num_rows == len(polys)


def _with_nulls(host: GeoArrowArray, seed: int) -> GeoArrowArray:
    rng = np.random.default_rng(seed)
    return GeoArrowArray(host.geom_type, host.xy, geom_offsets=host.geom_offsets, part_offsets=host.part_offsets, ring_offsets=host.ring_offsets,
                         validity=np.packbits(rng.uniform(size=len(host)) > 0.2, bitorder="little"), n_geoms=len(host))


def _multi_cases():
    ls = synth.random_linestrings(300, seed=9, max_log2=4.0)
    mp = GeoArrowArray(_abi.GEOM_MULTIPOINT, ls.xy, geom_offsets=ls.geom_offsets)
    polys = synth.star_polygons(200, 12)
    mls = GeoArrowArray(_abi.GEOM_MULTILINESTRING, polys.xy, geom_offsets=polys.geom_offsets, ring_offsets=polys.ring_offsets)
    return {"multipoints": mp, "multilinestrings": mls}


@pytest.mark.parametrize("case", list(CASES) + ["multipoints", "multilinestrings"])
@pytest.mark.parametrize("nulls", [False, True])
def test_results_leave_through_the_c_data_interface(gpk, case, nulls):
    """gpk_geoarray_to_arrow (the return half of ffi.rs:12-52): pyarrow imports the two structs the library fills — callee-owned buffers,
    real release callbacks — for all six nestings in both native layouts and as WKB, with and without nulls; importing the exported
    column again gives the same device array (the extension name carries MULTIPOINT / MULTILINESTRING)"""
    host = CASES[case]() if case in CASES else _multi_cases()[case]
    if nulls:
        host = _with_nulls(host, 3)
    dev = DeviceGeoArray.upload(host)
    n_null = 0 if host.validity is None else int(len(host) - np.unpackbits(host.validity, bitorder="little")[: len(host)].sum())
    for layout in ("struct", "interleaved"):
        col = dev.to_arrow(layout)
        assert len(col) == len(host) and col.null_count == n_null
        col.validate(full=True)
        # the nesting and the coordinates, level by level
        a = col
        depth = 0
        while pa.types.is_list(a.type):
            a = a.values if a.offset == 0 else a.flatten()
            depth += 1
        assert depth == {_abi.GEOM_POINT: 0, _abi.GEOM_LINESTRING: 1, _abi.GEOM_MULTIPOINT: 1, _abi.GEOM_POLYGON: 2, _abi.GEOM_MULTILINESTRING: 2, _abi.GEOM_MULTIPOLYGON: 3}[host.geom_type]
        if layout == "struct":
            assert pa.types.is_struct(a.type) and [a.type.field(i).name for i in range(2)] == ["x", "y"]
            xy = np.column_stack([a.field(0).to_numpy(zero_copy_only=False), a.field(1).to_numpy(zero_copy_only=False)])
        else:
            assert pa.types.is_fixed_size_list(a.type) and a.type.list_size == 2
            xy = a.values.to_numpy(zero_copy_only=False).reshape(-1, 2)
        assert np.array_equal(xy, host.xy)
        # round trip: the exported field's extension name decides the type (pyarrow keeps it as field metadata only through a schema,
        # so it is handed over as the hint here)
        back = DeviceGeoArray.from_arrow(col, geom_type=host.geom_type)
        _same(back, host)
        del col, a  # (pyarrow calls the release callbacks: the library's buffers are freed here)
    wkb = dev.to_arrow("wkb")
    assert pa.types.is_binary(wkb.type) and len(wkb) == len(host) and wkb.null_count == n_null
    v2, o2 = host.to_wkb()
    got_off = np.frombuffer(wkb.buffers()[1], dtype=np.int32)[: len(host) + 1]
    assert np.array_equal(got_off, o2) and np.array_equal(np.frombuffer(wkb.buffers()[2], dtype=np.uint8)[: len(v2)], v2)
    _same(DeviceGeoArray.from_arrow(wkb), host) if not nulls and host.geom_type in (_abi.GEOM_POINT, _abi.GEOM_LINESTRING, _abi.GEOM_POLYGON, _abi.GEOM_MULTIPOLYGON) else None


def test_to_arrow_extension_names_and_release(gpk):
    """the schema a native export carries names its type the GeoArrow way (the same names gpk_geoarray_from_arrow reads), and a released
    pair is marked released"""
    lib = _abi.lib()
    host = _multi_cases()["multipoints"]
    dev = DeviceGeoArray.upload(host)
    c_array, c_schema = _abi.ArrowArray(), _abi.ArrowSchema()
    _abi.check(lib.gpk_geoarray_to_arrow(dev.handle, _abi.ARROW_STRUCT, None, C.addressof(c_array), C.addressof(c_schema)))
    assert c_schema.format == b"+l" and c_array.length == len(host) and c_array.n_children == 1
    md = C.string_at(c_schema.metadata, 4 + 4 + 20 + 4 + len("geoarrow.multipoint"))
    assert md.endswith(b"geoarrow.multipoint") and b"ARROW:extension:name" in md
    # the structs go straight back in: the extension name makes it a MULTIPOINT column without a hint
    out, gt = C.c_void_p(), C.c_int32(-1)
    _abi.check(lib.gpk_geoarray_from_arrow(C.addressof(c_array), C.addressof(c_schema), -1, None, C.byref(out), C.byref(gt)))
    assert gt.value == _abi.GEOM_MULTIPOINT
    lib.gpk_geoarray_free(out)
    c_array.release(C.byref(c_array))
    c_schema.release(C.byref(c_schema))
    assert not c_array.release and not c_schema.release
    with pytest.raises(_abi.GeopolarsHipError):
        _abi.check(lib.gpk_geoarray_to_arrow(dev.handle, 7, None, C.addressof(c_array), C.addressof(c_schema)))
