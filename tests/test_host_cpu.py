"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
the host WKB decoder reproduces the reference fixtures, the oracle reproduces the committed golden
values and the fixtures' own known answers, and the product fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from geopolars_amd import _abi
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def golden_array(z) -> GeoArrowArray:
    f = lambda k: z[k] if k in z.files else None
    return GeoArrowArray(int(z["geom_type"]), z["xy"], f("geom_offsets"), f("part_offsets"), f("ring_offsets"))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "geopolars_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, not prose
    declared = set(re.findall(r"\b(gpk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(_abi.EXPORTED_SYMBOLS), declared ^ set(_abi.EXPORTED_SYMBOLS)
    lib = _abi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.gpk_version()


def test_integration_doc_names_every_symbol():
    """INTEGRATION.md's table and Rust shim cover the whole ABI: every entry point of the header appears as an extern line"""
    hdr = open(os.path.join(ROOT, "include", "geopolars_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpk_[a-z0-9_]+)\s*\(", hdr))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = doc[doc.index("extern \"C\" {") : doc.index("fn check(rc: i32)")]
    missing = sorted(n for n in declared if not re.search(r"fn %s\(" % n, shim))
    assert not missing, missing
    table = doc[: doc.index("## 2. The Rust shim")]
    assert not sorted(n for n in declared if n not in table)


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when no gfx950 is present (this container has no GPU)."""
    if _abi.device_count() > 0:
        pytest.skip("a GPU is visible")
    s = GeoSeries(GeoArrowArray.from_points([[0.0, 0.0]]))
    with pytest.raises(_abi.GeopolarsHipError) as e:
        s.area()
    assert e.value.code == _abi.GPK_ERR_DEVICE
    src = "".join(open(os.path.join(ROOT, "geopolars_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "geopolars_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("CPU oracle", "").replace("the oracle", "").lower() or "import oracle" not in src
    assert "from oracle" not in src and "import oracle" not in src and "pyoracle" not in src


def test_argument_validation_errors():
    lib = _abi.lib()
    d = GeoArrowArray.from_points([[0.0, 0.0]]).desc()
    d.geom_type = 7  # GeometryCollection: unsupported
    h = C.c_void_p()
    assert lib.gpk_geoarray_upload(C.byref(d), None, C.byref(h)) == _abi.GPK_ERR_MISMATCHED_GEOMETRY
    assert "unsupported geometry type" in _abi.last_error()
    bad = GeoArrowArray(_abi.GEOM_LINESTRING, np.zeros((3, 2)), geom_offsets=np.array([0, 2, 1], np.int32))
    d = bad.desc()
    rc = lib.gpk_geoarray_upload(C.byref(d), None, C.byref(h))
    assert rc in (_abi.GPK_ERR_INVALID_OFFSETS, _abi.GPK_ERR_DEVICE)


@pytest.mark.parametrize("name,n_rows", [("cities", 202), ("naturalearth_cities", 243), ("naturalearth_lowres", 177), ("nybb", 5)])
def test_wkb_decoder_matches_independent_parser(name, n_rows):
    """gpk_wkb_decode vs the struct-based parser that wrote the golden files (make_golden.py)."""
    z = load(name)
    a = GeoArrowArray.from_wkb(z["wkb_values"], z["wkb_offsets"])
    g = golden_array(z)
    assert len(a) == n_rows  # py-geopolars/tests/unit/internals/test_geoseries.py:4-5 pins 243
    assert a.geom_type == g.geom_type
    assert np.array_equal(a.xy, g.xy)
    for k in ("geom_offsets", "part_offsets", "ring_offsets"):
        x, y = getattr(a, k), getattr(g, k)
        assert (x is None) == (y is None)
        if x is not None:
            assert np.array_equal(x, y), k


def test_wkb_decoder_big_endian_ewkb_and_errors():
    import struct

    pt_be = struct.pack(">BIdd", 0, 1, 1.5, -2.5)
    pt_srid = struct.pack("<BIIdd", 1, 1 | 0x20000000, 4326, 3.0, 4.0)
    vals = np.frombuffer(pt_be + pt_srid, dtype=np.uint8)
    a = GeoArrowArray.from_wkb(vals, np.array([0, len(pt_be), len(pt_be) + len(pt_srid)], np.int32))
    assert a.geom_type == _abi.GEOM_POINT and a.xy.tolist() == [[1.5, -2.5], [3.0, 4.0]]
    # mixed Polygon + MultiPolygon promotes to MULTIPOLYGON
    ring = struct.pack("<I", 4) + b"".join(struct.pack("<dd", *c) for c in [(0, 0), (1, 0), (0, 1), (0, 0)])
    poly = struct.pack("<BII", 1, 3, 1) + ring
    mpoly = struct.pack("<BII", 1, 6, 2) + poly + poly
    vals = np.frombuffer(poly + mpoly, dtype=np.uint8)
    a = GeoArrowArray.from_wkb(vals, np.array([0, len(poly), len(poly) + len(mpoly)], np.int32))
    assert a.geom_type == _abi.GEOM_MULTIPOLYGON
    assert a.geom_offsets.tolist() == [0, 1, 3] and a.part_offsets.tolist() == [0, 1, 2, 3] and a.ring_offsets.tolist() == [0, 4, 8, 12]
    # Z / M ordinates are read past (geozero's to_geo, which the reference decodes rows with, yields 2D geometries): ISO 1000-codes,
    # EWKB flag bits, either byte order, nested in multi-geometries
    z_iso = struct.pack("<BIddd", 1, 1001, 7.0, 8.0, 99.0)
    m_iso = struct.pack(">BIddd", 0, 2001, 1.0, 2.0, 55.0)
    zm_iso = struct.pack("<BIdddd", 1, 3001, 3.0, 4.0, 9.0, 9.5)
    z_ewkb = struct.pack("<BIIddd", 1, 1 | 0x80000000 | 0x20000000, 4326, 5.0, 6.0, -1.0)
    zm_ewkb = struct.pack("<BIdddd", 1, 1 | 0xC0000000, 10.0, 11.0, 1.0, 2.0)
    rows = [z_iso, m_iso, zm_iso, z_ewkb, zm_ewkb]
    off = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    a = GeoArrowArray.from_wkb(np.frombuffer(b"".join(rows), np.uint8), off)
    assert a.geom_type == _abi.GEOM_POINT and a.xy.tolist() == [[7.0, 8.0], [1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [10.0, 11.0]]
    ring_z = struct.pack("<I", 4) + b"".join(struct.pack("<ddd", *c) for c in [(0, 0, 5), (1, 0, 5), (0, 1, 5), (0, 0, 5)])
    poly_z = struct.pack("<BII", 1, 1003, 1) + ring_z
    mpoly_z = struct.pack("<BII", 1, 1006, 2) + poly_z + poly_z
    a = GeoArrowArray.from_wkb(np.frombuffer(poly_z + mpoly_z, np.uint8), np.array([0, len(poly_z), len(poly_z) + len(mpoly_z)], np.int32))
    assert a.geom_type == _abi.GEOM_MULTIPOLYGON and a.ring_offsets.tolist() == [0, 4, 8, 12]
    assert a.xy[:4].tolist() == [[0, 0], [1, 0], [0, 1], [0, 0]] and np.array_equal(a.xy[:4], a.xy[8:12])
    # a GeometryCollection (type 7) has no GeoArrow nesting: rejected, as are mixed families and truncated buffers
    gc = struct.pack("<BII", 1, 7, 0)
    with pytest.raises(_abi.MismatchedGeometry):
        GeoArrowArray.from_wkb(np.frombuffer(gc, np.uint8), np.array([0, len(gc)], np.int32))
    with pytest.raises(_abi.MismatchedGeometry):
        v = np.frombuffer(pt_be + poly, np.uint8)
        GeoArrowArray.from_wkb(v, np.array([0, len(pt_be), len(pt_be) + len(poly)], np.int32))
    with pytest.raises(_abi.GeopolarsHipError):
        GeoArrowArray.from_wkb(np.frombuffer(poly[:-4], np.uint8), np.array([0, len(poly) - 4], np.int32))


def test_wkb_nulls_keep_row_alignment():
    import struct

    pt = struct.pack("<BIdd", 1, 1, 7.0, 8.0)
    vals = np.frombuffer(pt + pt, dtype=np.uint8)
    validity = np.packbits(np.array([1, 0, 1], np.uint8), bitorder="little")
    a = GeoArrowArray.from_wkb(vals, np.array([0, len(pt), len(pt), 2 * len(pt)], np.int32), validity)
    assert len(a) == 3 and a.is_valid().tolist() == [True, False, True]
    assert a.xy[0].tolist() == [7.0, 8.0] and np.isnan(a.xy[1]).all() and a.xy[2].tolist() == [7.0, 8.0]


def test_config1_cities_centroid_and_bounds_oracle(oracle):
    """BASELINE.json configs[0]: data/cities.arrow -> centroid + bounds on the CPU reference path.
    centroid(point) = point, bounds(point) = (x, y, x, y); row 0 is Vatican City."""
    z = load("cities")
    a = GeoArrowArray.from_wkb(z["wkb_values"], z["wkb_offsets"])
    c, v = oracle.centroid(a)
    b = oracle.bounds(a)
    assert v.all() and np.array_equal(c, a.xy)
    assert np.array_equal(b, np.concatenate([a.xy, a.xy], axis=1))
    assert a.xy[0].tolist() == [12.453386544971766, 41.903282179960115]
    assert np.allclose([a.xy[:, 0].min(), a.xy[:, 1].min(), a.xy[:, 0].max(), a.xy[:, 1].max()], [-175.2206, -41.3000, 179.2166, 64.1500], atol=1e-3)


@pytest.mark.parametrize("name", ["cities", "naturalearth_cities", "naturalearth_lowres", "nybb"])
def test_oracle_reproduces_golden(oracle, name):
    z = load(name)
    a = golden_array(z)
    assert np.array_equal(oracle.bounds(a), z["oracle_bounds"], equal_nan=True)
    assert np.array_equal(oracle.area(a), z["oracle_area"])
    assert np.array_equal(oracle.centroid(a)[0], z["oracle_centroid"], equal_nan=True)
    assert np.array_equal(oracle.euclidean_length(a), z["oracle_length"])


def all_rings_as_multilinestring(a: GeoArrowArray) -> GeoArrowArray:
    """every ring (exteriors and holes of every member) of a polygonal row as the members of one MultiLineString row"""
    go = a.geom_offsets if a.part_offsets is None else a.part_offsets[a.geom_offsets]
    return GeoArrowArray(_abi.GEOM_MULTILINESTRING, a.xy, geom_offsets=go.astype(np.int32), ring_offsets=a.ring_offsets)


def test_nybb_shape_area_and_length_known_answers(oracle):
    """nybb.arrow ships Shape_Area / Shape_Leng columns: a weak (1e-5) known-answer check of shoelace
    area and perimeter on real data."""
    z = load("nybb")
    a = golden_array(z)
    area = oracle.area(a)
    assert np.allclose(area, z["Shape_Area"], rtol=5e-6)
    # Shape_Leng is the total boundary length (all rings); euclidean_length follows the trait doc and
    # measures exterior rings only (geoseries.rs:38-40), so it can only be <= and close
    length = oracle.euclidean_length(a)
    assert (length <= z["Shape_Leng"] * (1 + 1e-4)).all()
    assert np.allclose(length, z["Shape_Leng"], rtol=2e-2)
    # the second reference-held pin for `length`: every ring of a borough as one MultiLineString -> its length IS Shape_Leng
    # (the fixture's own column; agreement 7e-6 .. 3.3e-5, SURVEY.md 8c — the column was computed by other software)
    rings = all_rings_as_multilinestring(a)
    assert np.allclose(oracle.euclidean_length(rings), z["Shape_Leng"], rtol=5e-5)


def test_lowres_contains_its_cities_oracle(oracle):
    """real-data join on the CPU oracle: every hit is confirmed by the pure-Python rational ring walk."""
    from tests.test_oracle_exact import py_ring_pos

    countries = golden_array(load("naturalearth_lowres"))
    cities = golden_array(load("naturalearth_cities"))
    pairs, counts, _ = oracle.spatial_join(cities, countries, "intersects", mode=1)
    pairs0, counts0, _ = oracle.spatial_join(cities, countries, "intersects", mode=0)
    assert np.array_equal(pairs, pairs0) and np.array_equal(counts, counts0)
    assert 150 < len(pairs) <= 243  # most capitals are inside a (low-resolution) country polygon
    for l, r in pairs[:25]:
        p0, p1 = countries.geom_offsets[r], countries.geom_offsets[r + 1]
        hit = False
        for p in range(p0, p1):
            r0 = countries.part_offsets[p]
            ring = countries.xy[countries.ring_offsets[r0] : countries.ring_offsets[r0 + 1]]
            hit |= py_ring_pos(tuple(cities.xy[l]), [tuple(c) for c in ring]) == 2
        assert hit


def test_geoarrow_to_pyarrow_layout():
    polys = GeoArrowArray.from_polygons([[[(0, 0), (4, 0), (4, 4), (0, 4)], [(1, 1), (1, 2), (2, 2), (2, 1)]], [], [[(5, 5), (6, 5), (6, 6)]]])
    arr = polys.to_pyarrow()
    assert len(arr) == 3 and arr[0].as_py()[0][0] == [0.0, 0.0]


# ---- GeoArrow -> WKB, host encoder (from_geom_vec, util.rs:11-24) -------------------------------------
@pytest.mark.parametrize("name", ["cities", "naturalearth_cities", "naturalearth_lowres", "nybb"])
def test_wkb_encoder_reproduces_the_reference_fixture_bytes(name):
    """decode the reference's own WKB columns, encode them again: byte-identical whenever the fixture is plain
    little-endian ISO WKB of one type; always identical after one more decode."""
    z = load(name)
    a = GeoArrowArray.from_wkb(z["wkb_values"], z["wkb_offsets"])
    values, offsets = a.to_wkb()
    b = GeoArrowArray.from_wkb(values, offsets)
    assert b.geom_type == a.geom_type and np.array_equal(a.xy, b.xy)
    for k in ("geom_offsets", "part_offsets", "ring_offsets"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None) and (x is None or np.array_equal(x, y)), k
    if name in ("cities", "naturalearth_cities"):  # point columns: nothing to promote, so the bytes come back as they were
        assert np.array_equal(values, z["wkb_values"]) and np.array_equal(offsets, z["wkb_offsets"])


def test_wkb_encoder_matches_independent_writer_and_handles_nulls():
    from geopolars_amd import synth
    from tests.wkb_util import encode_wkb

    arrays = [
        synth.uniform_points(500),
        synth.random_linestrings(300),
        synth.star_polygons(200, 9),
        synth.powerlaw_multipolygons(300, seed=5),
        GeoArrowArray(_abi.GEOM_MULTIPOINT, np.arange(20.0).reshape(10, 2), geom_offsets=np.array([0, 3, 3, 10], np.int32)),
    ]
    ml = synth.random_linestrings(64)
    arrays.append(GeoArrowArray(_abi.GEOM_MULTILINESTRING, ml.xy, geom_offsets=np.array([0, 1, 1, 10, 64], np.int32), ring_offsets=ml.geom_offsets))
    for a in arrays:
        ev, eo = encode_wkb(a)
        v, o = a.to_wkb()
        assert np.array_equal(o, eo) and np.array_equal(v, ev), a.geom_type
    polys = synth.star_polygons(50, 6)
    keep = np.ones(50, np.uint8)
    keep[::3] = 0
    pn = GeoArrowArray(polys.geom_type, polys.xy, polys.geom_offsets, ring_offsets=polys.ring_offsets, validity=np.packbits(keep, bitorder="little"))
    ev, eo = encode_wkb(pn)
    v, o = pn.to_wkb()
    assert np.array_equal(o, eo) and np.array_equal(v, ev)
    assert np.all(np.diff(o)[keep == 0] == 0)  # null rows are zero-length
    col = pn.to_arrow_wkb()
    assert col.null_count == int((keep == 0).sum()) and col[1].as_py() == bytes(v[o[1] : o[2]])
    # count-only contract and capacity error
    lib = _abi.lib()
    d = pn.desc()
    nb = C.c_int64(0)
    assert lib.gpk_wkb_encode(C.byref(d), None, None, 0, C.byref(nb)) == _abi.GPK_OK and nb.value == len(v)
    small = np.empty(10, np.uint8)
    assert lib.gpk_wkb_encode(C.byref(d), None, small.ctypes.data, 10, C.byref(nb)) == _abi.GPK_ERR_CAPACITY and nb.value == len(v)


def build_and_run_c_client(tmp_path):
    """include/geopolars_hip.h compiled as strict C99 by gcc, linked against the library, run as a process -> the finished process"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "geopolars_amd")
    exe = str(tmp_path / "c_abi_client")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi_client.c"),
           "-o", exe, "-L", lib_dir, "-lgeopolars_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    return r


def test_plain_c_client_links_and_fails_loudly_without_a_device(tmp_path):
    r = build_and_run_c_client(tmp_path)
    assert "no device" in r.stdout or "area" in r.stdout


def test_off_path_operators_say_so():
    """to_crs (PROJ) belongs to the reference surface but not to this backend: a clear error instead of an
    AttributeError; an unknown geodesic method is rejected with the reference's message before any device is touched;
    translate takes the Python surface's parameter names"""
    import inspect

    from geopolars_amd.geoseries import GeoSeries

    s = GeoSeries(GeoArrowArray.from_points([(0.0, 0.0)]))
    with pytest.raises(NotImplementedError, match="not on the accelerated path"):
        s.to_crs("EPSG:4326", "EPSG:3857")
    with pytest.raises(ValueError, match="Geodesic calculation method not valid"):
        s.geodesic_length("rhumb")
    assert list(inspect.signature(GeoSeries.translate).parameters)[1:] == ["xoff", "yoff"]
    assert list(inspect.signature(GeoSeries.simplify).parameters)[1:] == ["tolerance"]


def test_wkb_host_codec_round_trip_on_random_structures():
    """encode -> decode is the identity on ragged columns: multipolygons with empty rows, empty members' neighbours, holes,
    rings of 1..9 coordinates; linestrings and multipoints with empty rows (host codec, gpk_wkb.cpp)"""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    coord = st.tuples(st.integers(-1000, 1000), st.integers(-1000, 1000))
    ring = st.lists(coord, min_size=1, max_size=9)
    polygon = st.lists(ring, min_size=1, max_size=3)
    multipolygon = st.lists(polygon, min_size=0, max_size=3)

    @settings(max_examples=60, deadline=None)
    @given(st.lists(multipolygon, min_size=1, max_size=6), st.lists(st.lists(coord, min_size=0, max_size=6), min_size=1, max_size=6))
    def run(mps, lines):
        cases = [
            GeoArrowArray.from_multipolygons(mps, close=False),
            GeoArrowArray.from_linestrings(lines),
            GeoArrowArray(_abi.GEOM_MULTIPOINT, np.array([c for l in lines for c in l], dtype=np.float64).reshape(-1, 2),
                          geom_offsets=np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)),
        ]
        for a in cases:
            v, o = a.to_wkb()
            b = GeoArrowArray.from_wkb(v, o)
            assert b.geom_type == a.geom_type and np.array_equal(b.xy, a.xy) and np.array_equal(b.geom_offsets, a.geom_offsets)
            for name in ("part_offsets", "ring_offsets"):
                x, y = getattr(a, name), getattr(b, name)
                assert (x is None) == (y is None) and (x is None or np.array_equal(x, y)), name

    run()


def test_bench_helpers_sampling_and_traffic_records(tmp_path, monkeypatch):
    """bench.py's host-side helpers: the parity sample, the pair extraction the sampled check relies on, and the rule that a PMC
    traffic record is reported only for the sources it was measured at"""
    import importlib
    import json as _json

    bench = importlib.import_module("bench")
    rows = bench.sample_rows(1000, 50, seed=1)
    assert len(rows) == 50 and len(set(rows.tolist())) == 50 and np.all(np.diff(rows) > 0) and rows.max() < 1000
    assert len(bench.sample_rows(10, 50, seed=1)) == 10  # never more rows than there are
    pairs = np.array([[0, 5], [2, 1], [2, 7], [3, 3], [9, 0], [9, 4]], dtype=np.uint32)
    got = bench.pairs_of_rows(pairs, np.array([2, 4, 9], dtype=np.int64))
    assert got.tolist() == [[0, 1], [0, 7], [2, 0], [2, 4]]  # l renumbered to the position in the sample; row 4 has no pair
    assert bench.pairs_of_rows(np.zeros((0, 2), dtype=np.uint32), np.array([1, 2])).shape == (0, 2)
    # traffic records: the committed one matches this tree ...
    rec, why = bench.pmc_record("gpk_pip_tile")
    assert (rec is not None and why is None and rec["source_hash"] == bench.source_hash()) or (rec is None and "stale" in why)
    # ... and a record measured at other sources is refused with the reason
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_pmc_traffic.json").write_text(_json.dumps({"kernel": "gpk_pip_tile", "source_hash": "0" * 16, "traffic_bytes_per_launch": 1}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "f" * 16)
    rec, why = bench.pmc_record("gpk_pip_tile")
    assert rec is None and "stale" in why and "0000" in why
    rec, why = bench.pmc_config_record("c4")
    assert rec is None and "no PMC record" in why
    # the record of THIS tree is found by its hash, whatever the files are called (tags do not sort by time: r03t < r03zz)
    (prof / "r03a_pmc_traffic.json").write_text(_json.dumps({"kernel": "gpk_pip_tile", "source_hash": "f" * 16, "traffic_bytes_per_launch": 7}))
    (prof / "r03a_pmc_traffic_configs.json").write_text(_json.dumps({"source_hash": "f" * 16, "read_factor_of_the_16_byte_stream": 1.9,
                                                                      "configs": {"c4": {"kernel": "pair_refine", "traffic_bytes_raw": 5, "traffic_bytes_with_read_factor": 9}}}))
    (prof / "r99_pmc_traffic_configs.json").write_text(_json.dumps({"source_hash": "0" * 16, "configs": {"c4": {"kernel": "pair_refine", "traffic_bytes_raw": 1}}}))
    rec, why = bench.pmc_record("gpk_pip_tile")
    assert rec is not None and why is None and rec["traffic_bytes_per_launch"] == 7
    rec, why = bench.pmc_config_record("c4")
    assert rec is not None and rec["traffic_bytes_raw"] == 5


def test_bench_gpus_n_without_a_launcher_spawns_n_ranks():
    """`python bench.py --gpus 2` must not exit asking for a launcher: it re-runs itself as 2 ranks under torch.distributed.run.
    There is no GPU here, so each rank stops at "no GPU visible" — which shows that both were started."""
    import subprocess, sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert "launching 2 ranks under torch.distributed.run" in r.stderr
    import torch

    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stderr.count("no GPU visible") == 2


def test_arrow_c_data_import_walks_the_structs_before_it_needs_a_device():
    """gpk_geoarray_from_arrow (the reference's FFI seam, py-geopolars/src/ffi.rs:12-32) is host code up to the upload: what is not a
    geometry column is refused by its FORMAT (no device involved), what is one gets as far as the device and — here, without one —
    says GPK_ERR_DEVICE; separated x / y coordinates reach the host WKB encoder through the same descriptor"""
    import ctypes as C

    import pyarrow as pa

    from geopolars_amd.geoarrow import DeviceGeoArray

    from geopolars_amd import synth as _s

    host = _s.star_polygons(6, 8)
    st = pa.StructArray.from_arrays([pa.array(host.xy[:, 0]), pa.array(host.xy[:, 1])], ["x", "y"])
    col = pa.ListArray.from_arrays(pa.array(host.geom_offsets), pa.ListArray.from_arrays(pa.array(host.ring_offsets), st))
    on_gpu = _abi.device_count() > 0
    for bad in (pa.array([1.0, 2.0]), pa.array(["a"]), pa.ListArray.from_arrays(pa.array([0, 2], type=pa.int32()), pa.array([1.0, 2.0]))):
        with pytest.raises(_abi.GeopolarsHipError) as e:
            DeviceGeoArray.from_arrow(bad)
        assert e.value.code == _abi.GPK_ERR_MISMATCHED_GEOMETRY
    with pytest.raises(_abi.GeopolarsHipError) as e:
        DeviceGeoArray.from_arrow(col, geom_type=_abi.GEOM_POINT)  # two list levels cannot be points
    assert e.value.code == _abi.GPK_ERR_MISMATCHED_GEOMETRY
    if not on_gpu:
        for ok in (col, col.slice(1, 3), pa.array([b"\x01"], type=pa.large_binary())):
            with pytest.raises(_abi.GeopolarsHipError) as e:
                DeviceGeoArray.from_arrow(ok)
            assert e.value.code == _abi.GPK_ERR_DEVICE
    # the host encoder reads separated coordinates
    d = host.desc()
    xs, ys = np.ascontiguousarray(host.xy[:, 0]), np.ascontiguousarray(host.xy[:, 1])
    d.xy, d.x, d.y = None, xs.ctypes.data, ys.ctypes.data
    nb = C.c_int64(0)
    off = np.zeros(len(host) + 1, dtype=np.int32)
    _abi.check(_abi.lib().gpk_wkb_encode(C.byref(d), off.ctypes.data, None, 0, C.byref(nb)))
    vals = np.zeros(nb.value, dtype=np.uint8)
    _abi.check(_abi.lib().gpk_wkb_encode(C.byref(d), off.ctypes.data, vals.ctypes.data, len(vals), C.byref(nb)))
    v2, o2 = host.to_wkb()
    assert np.array_equal(vals, v2) and np.array_equal(off, o2)
    # coordinates given twice are refused
    d.xy = host.xy.ctypes.data
    out = C.c_void_p()
    assert _abi.lib().gpk_geoarray_upload(C.byref(d), None, C.byref(out)) in (_abi.GPK_ERR_INVALID_ARGUMENT, _abi.GPK_ERR_DEVICE)


def test_bench_always_prints_a_line_when_a_rank_dies_or_the_run_hangs(tmp_path):
    """rank 0 prints ONE JSON line whatever happens: an exception in the rank (the line carries `error`), a run that outlives the
    watchdog, another rank's death (its marker file wakes the survivors' watchdogs, which would otherwise wait in a collective)"""
    import json, subprocess, sys as _sys, time as _time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = [_sys.executable, os.path.join(root, "bench.py"), "--config", "c4", "--steps", "1", "--warmup", "1"]

    def line_of(r):
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])

    r = subprocess.run(bench, capture_output=True, text=True, timeout=120, env=dict(base, GPK_BENCH_RAISE="boom", TMPDIR=str(tmp_path)))
    assert r.returncode == 5
    line = line_of(r)
    assert "boom" in line["error"] and line["value"] is None and line["n_gpus"] == 1 and "intersects" in line["metric"]
    r = subprocess.run(bench + ["--watchdog", "2"], capture_output=True, text=True, timeout=120, env=dict(base, GPK_BENCH_HANG="1", TMPDIR=str(tmp_path)))
    assert r.returncode == 4 and "watchdog" in line_of(r)["error"]
    # rank 1 of 2 dies; rank 0 (hanging, as it would inside a collective) learns of it and prints the line
    env2 = dict(base, WORLD_SIZE="2", MASTER_PORT="29999", TORCHELASTIC_RUN_ID="t", TMPDIR=str(tmp_path))
    p0 = subprocess.Popen(bench, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(env2, RANK="0", GPK_BENCH_HANG="1"))
    _time.sleep(1.0)
    r1 = subprocess.run(bench, capture_output=True, text=True, timeout=120, env=dict(env2, RANK="1", GPK_BENCH_RAISE="rank one is gone"))
    assert r1.returncode == 5 and '"metric"' not in r1.stdout  # (only rank 0 prints)
    out, _ = p0.communicate(timeout=60)
    assert p0.returncode == 4
    line = json.loads([l for l in out.splitlines() if l.startswith('{"metric"')][-1])
    assert "rank1 failed" in line["error"] and "rank one is gone" in line["error"] and line["n_gpus"] == 2
