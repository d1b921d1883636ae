#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench.py): the streaming operators and the C3 distance
kernel, device-resident inputs and outputs, HIP-event kernel times from gpk_profile_*.

    python tools/bench_ops.py [--scale 1.0] > profiles/<round>_ops.jsonl

Each line: op, workload, kernel ms, algorithmic bytes (SURVEY.md §8d formulas), GB/s, fraction of 8 TB/s.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth  # noqa: E402
from geopolars_amd.geoarrow import DeviceGeoArray  # noqa: E402

PEAK = 8000.0


def dev_array(a, dev):
    t = lambda x, dt: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return DeviceGeoArray.from_device_buffers(
        a.geom_type, t(a.xy, None), t(a.geom_offsets, None), t(a.part_offsets, None), t(a.ring_offsets, None), stream=torch.cuda.current_stream().cuda_stream
    )


def timed_impl(lib, fn, reps=10):
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        fn(stream)
    torch.cuda.synchronize()
    lib.gpk_profile_reset()
    lib.gpk_profile_enable(1)
    for _ in range(reps):
        fn(stream)
    torch.cuda.synchronize()
    lib.gpk_profile_enable(0)
    ms, cnt = C.c_double(0), C.c_int64(0)
    lib.gpk_profile_query(b"", C.byref(ms), C.byref(cnt))
    out = {}
    # per-kernel breakdown
    # per-kernel breakdown: ms per call of the operator (a name may cover several launches)
    for name in (b"gpk_ring_area", b"gpk_seq_bbox", b"gpk_ring_centroid", b"gpk_seq_long_combine", b"gpk_area_combine", b"gpk_bounds_combine", b"gpk_centroid_combine", b"gpk_affine", b"gpk_distance", b"gpk_length_combine", b"gpk_seq_length", b"gpk_hull", b"gpk_wkb", b"gpk_ring_stream_area", b"gpk_ring_stream_length", b"gpk_ring_stream_bounds", b"gpk_ring_stream_fix"):
        m, c = C.c_double(0), C.c_int64(0)
        lib.gpk_profile_query(name, C.byref(m), C.byref(c))
        if c.value:
            out[name.decode()] = m.value / reps
    lib.gpk_profile_reset()
    # wall time per call on the stream (library kernels that are not bracketed, e.g. rocPRIM sorts, and launch gaps included)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(stream)
    e1.record()
    torch.cuda.synchronize()
    out["wall_ms_per_call"] = e0.elapsed_time(e1) / reps
    return ms.value / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--ops", default="", help="comma-separated subset of operators (default: all)")
    args = ap.parse_args()
    only = set(x for x in args.ops.split(",") if x)
    lib = _abi.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    name, cus = _abi.device_info()
    MEM_DEVICE = _abi.MEM_DEVICE

    def timed(lib_, fn, reps=10, _t=timed_impl):  # an operator that was not asked for is not run
        return _t(lib_, fn, reps) if (not only or timed.op in only) else (None, None)

    def report(op, workload, ms, kernels, nbytes):
        if ms is None:
            return
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(json.dumps({"op": op, "workload": workload, "ms": ms, "kernels_ms": kernels, "algorithmic_bytes": nbytes, "GBps": gbs, "frac_of_8TBps": gbs / PEAK, "device": name}), flush=True)

    # ---- streaming unary ops on polygon arrays ------------------------------------------------------
    workloads = {
        "2M x 64-vertex polygons": synth.star_polygons(int(2_000_000 * args.scale), 64),
        "8M x 8-vertex polygons": synth.star_polygons(int(8_000_000 * args.scale), 8),
        "1M power-law multipolygons (C5 shape)": synth.powerlaw_multipolygons(int(1_000_000 * args.scale)),
    }
    for wname, a in workloads.items():
        d = dev_array(a, dev)
        n, v = len(a), a.n_coords
        off_bytes = sum(x.nbytes for x in (a.geom_offsets, a.part_offsets, a.ring_offsets) if x is not None)
        out1 = torch.empty(n, dtype=torch.float64, device=dev)
        out2 = torch.empty((n, 2), dtype=torch.float64, device=dev)
        out4 = torch.empty((n, 4), dtype=torch.float64, device=dev)
        outv = torch.empty(n, dtype=torch.uint8, device=dev)
        outxy = torch.empty((v, 2), dtype=torch.float64, device=dev)
        m6 = (C.c_double * 6)(0.5, -0.25, 3.0, 0.25, 0.5, -7.0)
        timed.op = "area"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_area(d.handle, out1.data_ptr(), MEM_DEVICE, s)))
        report("area", wname, ms, k, 16 * v + off_bytes + 8 * n)
        timed.op = "bounds"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_bounds(d.handle, out4.data_ptr(), MEM_DEVICE, s)))
        report("bounds", wname, ms, k, 16 * v + off_bytes + 32 * n)
        timed.op = "centroid"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_centroid(d.handle, out2.data_ptr(), outv.data_ptr(), MEM_DEVICE, s)))
        report("centroid", wname, ms, k, 16 * v + off_bytes + 17 * n)
        timed.op = "affine_transform"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_affine_transform(d.handle, m6, outxy.data_ptr(), MEM_DEVICE, s)))
        report("affine_transform", wname, ms, k, 32 * v)
        timed.op = "euclidean_length"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_euclidean_length(d.handle, out1.data_ptr(), MEM_DEVICE, s)))
        report("euclidean_length", wname, ms, k, 16 * v + off_bytes + 8 * n)
        if wname.startswith("2M"):
            hxy = torch.empty((v + n, 2), dtype=torch.float64, device=dev)
            hoff = torch.empty(n + 1, dtype=torch.int32, device=dev)
            timed.op = "convex_hull"
            ms, k = timed(lib, lambda s: _abi.check(lib.gpk_convex_hull(d.handle, hxy.data_ptr(), hoff.data_ptr(), MEM_DEVICE, s)), reps=3)
            report("convex_hull", wname, ms, k, 32 * v + off_bytes)
            del hxy, hoff
        # GeoArrow -> WKB on the device: coordinates in, WKB bytes out
        nb = C.c_int64(0)
        _abi.check(lib.gpk_geoarray_to_wkb(d.handle, None, None, 0, C.byref(nb), MEM_DEVICE, None))
        wkb = torch.empty(int(nb.value), dtype=torch.uint8, device=dev)
        woff = torch.empty(n + 1, dtype=torch.int32, device=dev)
        timed.op = "to_wkb"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_geoarray_to_wkb(d.handle, woff.data_ptr(), wkb.data_ptr(), int(nb.value), C.byref(nb), MEM_DEVICE, s)), reps=5)
        report("to_wkb", wname, ms, k, 16 * v + off_bytes + int(nb.value) + 4 * n)
        # WKB -> GeoArrow on the device (the bytes are already in HBM)
        def decode(s):
            out, gt = C.c_void_p(), C.c_int32(-1)
            _abi.check(lib.gpk_geoarray_from_wkb(wkb.data_ptr(), woff.data_ptr(), n, None, MEM_DEVICE, s, C.byref(out), C.byref(gt)))
            lib.gpk_geoarray_free(out)
        timed.op = "from_wkb"
        ms, k = timed(lib, decode, reps=3)
        report("from_wkb", wname, ms, k, int(nb.value) + 4 * n + 16 * v + off_bytes)
        del d, out1, out2, out4, outv, outxy, wkb, woff
        torch.cuda.empty_cache()

    # ---- C3: 10M points x 100k linestrings, row-wise distance -----------------------------------------
    npts, nls = int(10_000_000 * args.scale), int(100_000 * args.scale)
    ls = synth.random_linestrings(nls)
    pts = synth.uniform_points(npts)
    dl, dp = dev_array(ls, dev), dev_array(pts, dev)
    out = torch.empty(npts, dtype=torch.float64, device=dev)
    for label, rows in (("rows = i mod L", np.arange(npts, dtype=np.uint32) % nls), ("rows shuffled", np.random.default_rng(1).permutation(np.arange(npts, dtype=np.uint32) % nls))):
        r = torch.from_numpy(rows.astype(np.int32)).to(dev)
        timed.op = "distance"
        ms, k = timed(lib, lambda s: _abi.check(lib.gpk_distance_rowwise(dp.handle, dl.handle, r.data_ptr(), out.data_ptr(), MEM_DEVICE, s)), reps=5)
        # 16N + 4N + 16*V_ls + 4(L+1) + 8N (each distinct byte once)
        report("distance", f"C3: {npts} points x {nls} linestrings ({ls.n_coords} coords), {label}", ms, k, 16 * npts + 4 * npts + 16 * ls.n_coords + 4 * (nls + 1) + 8 * npts)


if __name__ == "__main__":
    main()
