"""The one-pass form of area / signed_area / euclidean_length (csrc/gpk_ringstream.hip: strips of 1024 coordinates a wave, ring values
in an LDS table, geometries folded at the end of the strip, strip-crossing geometries from per-column ring records) against the oracle
and against the two-stage form of csrc/gpk_unary.hip, on columns built to stress what is new in it:

  * rings and geometries cut by strip boundaries at every offset (ring lengths coprime to 1024, a sweep of leading paddings);
  * rings longer than a strip, than many strips (partial sums of whole strips), geometries of many rings across several strips;
  * rings that begin and end inside one lane's 8 coordinates (triangles), the cap of rings per strip, columns that are NOT eligible
    (a zero-length ring) and must take the two-stage form;
  * open rings (area 0 by area.rs), collinear rings (area exactly 0), null rows, empty polygons, multipolygons with holes;
  * 24 seeded random columns (ring lengths log-uniform in 3 .. 5000, 1 in 20 left open, 1 - 4 polygons a geometry with up to 3 holes,
    empty geometries, null rows; POLYGON and MULTIPOLYGON);
  * bit-reproducibility run to run (the sums' order is fixed by the column's layout).

GPK_RING_STREAM (read once per process) selects the form: each form runs in its own interpreter.  Tolerance: 1e-9 relative (north
star), exact where the oracle's value is exactly 0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_PROG = r"""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from geopolars_amd import synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from oracle import pyoracle as oracle
oracle.build(); oracle.lib()

def ring(n, r, cx, cy, phase=0.0, cw=False):
    t = phase + 2 * np.pi * np.arange(n) / n
    if cw: t = -t
    rr = r * (1.0 + 0.3 * np.sin(5 * t))
    return [(cx + rr[i] * np.cos(t[i]), cy + rr[i] * np.sin(t[i])) for i in range(n)]

def check(name, a, exact_zero_rows=()):
    s = GeoSeries(a)
    for op, got, exp in (("area", s.area(), oracle.area(a)), ("signed_area", s.signed_area(), oracle.area(a, signed=True)),
                         ("length", s.euclidean_length(), oracle.euclidean_length(a))):
        got, exp = np.asarray(got), np.asarray(exp)
        assert got.shape == exp.shape, (name, op)
        assert np.array_equal(np.isnan(got), np.isnan(exp)), (name, op, "null rows")
        m = ~np.isnan(exp)
        bad = np.abs(got[m] - exp[m]) > 1e-9 * np.maximum(np.abs(exp[m]), 1e-300)
        assert not bad.any(), (name, op, np.nonzero(bad)[0][:5], got[m][bad][:5], exp[m][bad][:5])
        again = np.asarray(getattr(s, op if op != "length" else "euclidean_length")())
        assert np.array_equal(again, got, equal_nan=True), (name, op, "not reproducible")
    assert np.array_equal(s.bounds(), oracle.bounds(a), equal_nan=True), (name, "bounds")

rng = np.random.default_rng(7)
cases = {}
# 1. every alignment of rings against strips: ring lengths coprime to 1024, a few hundred rings each
for n in (3, 4, 7, 9, 63, 65, 127, 511, 1023, 1025, 2049, 5000):
    k = max(3, 20000 // n)
    cases["rings_%d" % n] = GeoArrowArray.from_polygons([[ring(n, 5.0 + (i % 7), 30.0 * i, 11.0)] for i in range(k)])
# 2. a sweep of paddings: the same polygons after 0 .. 9 leading triangles (every phase of a boundary within a lane)
body = [[ring(37, 4.0, 10.0 * i, 3.0)] for i in range(300)]
for pad in range(10):
    cases["pad_%d" % pad] = GeoArrowArray.from_polygons([[ring(3, 1.0, -5.0 * j, -9.0)] for j in range(pad)] + body)
# 3. giants: rings of many strips, with holes that are giants too, between small neighbours
cases["giants"] = GeoArrowArray.from_polygons(
    [[ring(11, 2.0, 0, 0)], [ring(100001, 500.0, 2000.0, 0.0), ring(30011, 100.0, 2000.0, 0.0, 0.3, cw=True)], [ring(5, 1.0, 9, 9)],
     [ring(3000, 40.0, -900.0, 0.0), ring(1500, 10.0, -900.0, 0.0, cw=True), ring(700, 5.0, -880.0, 5.0, cw=True)], [ring(8, 1.0, 1, 1)]])
# 4. multipolygons: power-law ring sizes (the benchmark's shape), and geometries of MANY rings across several strips
cases["powerlaw"] = synth.powerlaw_multipolygons(20000)
mp = []
for g in range(40):
    parts = []
    for p in range(1 + (g * 7) % 23):
        rings_ = [ring(4 + (g + p) % 90, 3.0, 50.0 * g + 8.0 * p, 0.0)]
        for h in range((g + p) % 4):
            rings_.append(ring(3 + (h + p) % 40, 0.5, 50.0 * g + 8.0 * p, 0.0, cw=True))
        parts.append(rings_)
    mp.append(parts)
cases["many_rings"] = GeoArrowArray.from_multipolygons(mp)
# 5. degenerate rings: open (area 0), collinear (area exactly 0), repeated points, two-coordinate rings; empty polygons; nulls
line = [(float(i), 3.0 * i) for i in range(2600)]
r9 = ring(9, 2.0, 5, 5)
deg = [[[(0, 0), (4, 0), (4, 4), (0, 4.5)]], [line + line[-2::-1]], [[(1, 1), (1, 1), (1, 1), (1, 1)]], [], [r9 + r9[:1]], [[(0, 0), (1, 1), (0, 0)]], [[(2, 2), (3, 3)]]]
a = GeoArrowArray.from_polygons(deg * 30, close=False)  # (the first ring stays open: area 0 by area.rs)
cases["degenerate"] = a
v = np.ones(a.n_geoms, dtype=bool)
v[::5] = False
cases["nulls"] = GeoArrowArray(a.geom_type, a.xy, geom_offsets=a.geom_offsets, ring_offsets=a.ring_offsets, validity=np.packbits(v, bitorder="little"))
# 7. seeded random columns: ring lengths log-uniform in 3 .. 5000 (closed or, 1 in 20, left open), 1 - 4 polygons a geometry with 0 - 3 holes,
#    empty geometries, nulls; POLYGON and MULTIPOLYGON
for seed in range(24):
    r_ = np.random.default_rng(1000 + seed)
    n_geoms = int(r_.integers(5, 400))
    big = seed % 3 == 0
    geoms = []
    for g in range(n_geoms):
        parts = []
        for p in range(0 if r_.random() < 0.05 else int(r_.integers(1, 5))):
            rings_ = []
            for h in range(int(r_.integers(1, 5))):
                n = int(np.exp(r_.uniform(np.log(3), np.log(5000 if big else 60))))
                pts = ring(n, float(r_.uniform(0.5, 50.0)), float(r_.uniform(-1e4, 1e4)), float(r_.uniform(-1e4, 1e4)), cw=h > 0)
                rings_.append(pts + ([] if r_.random() < 0.05 else pts[:1]))
            parts.append(rings_)
        geoms.append(parts)
    if seed % 2 == 0:
        arr = GeoArrowArray.from_multipolygons(geoms, close=False)
    else:
        arr = GeoArrowArray.from_polygons([pp[0] if pp else [] for pp in geoms], close=False)
    if seed % 4 == 1:
        vv = np.packbits(r_.random(arr.n_geoms) > 0.2, bitorder="little")
        arr = GeoArrowArray(arr.geom_type, arr.xy, geom_offsets=arr.geom_offsets, part_offsets=arr.part_offsets, ring_offsets=arr.ring_offsets, validity=vv)
    cases["random_%d" % seed] = arr
for name, arr in cases.items():
    check(name, arr)
# 6. a column that is not eligible (a zero-length ring): the two-stage form answers, whatever GPK_RING_STREAM says
ro = np.array([0, 4, 4, 9], dtype=np.int32)
xy = np.array([(0, 0), (2, 0), (2, 2), (0, 0), (5, 5), (7, 5), (7, 7), (5, 7), (5, 5)], dtype=np.float64)
check("zero_length_ring", GeoArrowArray(cases["giants"].geom_type, xy, geom_offsets=np.array([0, 2, 3], dtype=np.int32), ring_offsets=ro))
print("RINGSTREAM_OK", len(cases) + 1)
"""


@pytest.mark.parametrize("mode", ["1", "0", None])
def test_one_pass_and_two_stage_forms_agree_with_the_oracle(mode):
    """mode 1: every eligible column through the one-pass form; 0: none; None: the product's choice by column shape"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("GPK_RING_STREAM", None)
    if mode is not None:
        env["GPK_RING_STREAM"] = mode
    r = subprocess.run([sys.executable, "-c", _PROG], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0 and "RINGSTREAM_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_a_borrowed_view_whose_offsets_are_rewritten_is_invalidated(gpk, oracle):
    """The strip table and the ring records of the one-pass form are derived from a handle's OFFSETS once (include/geopolars_hip.h,
    gpk_geoarray_invalidate): a handle over borrowed device tensors answers for the new layout after its owner rewrote the offsets in
    place and said so; rewriting coordinates needs nothing."""
    import numpy as np
    import torch

    from geopolars_amd import synth
    from geopolars_amd.geoarrow import DeviceGeoArray, GeoArrowArray

    a = synth.powerlaw_multipolygons(3000)
    dev = torch.device("cuda", 0)
    xy = torch.from_numpy(np.ascontiguousarray(a.xy)).to(dev)
    go = torch.from_numpy(a.geom_offsets).to(dev)
    po = torch.from_numpy(a.part_offsets).to(dev)
    ro = torch.from_numpy(a.ring_offsets).to(dev)
    h = DeviceGeoArray.from_device_buffers(a.geom_type, xy, go, po, ro, stream=torch.cuda.current_stream().cuda_stream)

    def area_of(handle, n):
        out = torch.empty(n, dtype=torch.float64, device=dev)
        from geopolars_amd import _abi

        _abi.check(_abi.lib().gpk_area(handle.handle, out.data_ptr(), _abi.MEM_DEVICE, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def close(got, exp):
        m = ~np.isnan(exp)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert not (np.abs(got[m] - exp[m]) > 1e-9 * np.maximum(np.abs(exp[m]), 1e-300)).any()

    close(area_of(h, a.n_geoms), oracle.area(a))
    # coordinates rewritten in place (a translation keeps every ring closed): no invalidation needed
    xy += 3.0
    moved = GeoArrowArray(a.geom_type, a.xy + 3.0, geom_offsets=a.geom_offsets, part_offsets=a.part_offsets, ring_offsets=a.ring_offsets)
    close(area_of(h, a.n_geoms), oracle.area(moved))
    # the same coordinates cut into OTHER geometries: every polygon becomes a geometry of its own (geometry offsets 0 .. n_parts), in place
    n_parts = len(a.part_offsets) - 1
    assert n_parts + 1 >= a.n_geoms + 1
    go2 = torch.arange(0, a.n_geoms + 1, dtype=torch.int32, device=dev)  # geometry g = polygon g alone; the polygons beyond a.n_geoms are dropped with their offsets
    go.copy_(go2)
    h.invalidate()
    relaid = GeoArrowArray(a.geom_type, a.xy + 3.0, geom_offsets=np.arange(0, a.n_geoms + 1, dtype=np.int32), part_offsets=a.part_offsets[: a.n_geoms + 1],
                           ring_offsets=a.ring_offsets[: a.part_offsets[a.n_geoms] + 1])
    relaid = GeoArrowArray(relaid.geom_type, relaid.xy[: relaid.ring_offsets[-1]], geom_offsets=relaid.geom_offsets, part_offsets=relaid.part_offsets, ring_offsets=relaid.ring_offsets)
    # (the handle still spans the old buffers: rings and coordinates beyond the last geometry are no longer any geometry's — the one-pass
    # form's eligibility check sees offsets that do not cover the coordinates and the two-stage form answers)
    close(area_of(h, a.n_geoms), oracle.area(relaid))
    h.free()
