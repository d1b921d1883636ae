#!/usr/bin/env python3
"""The two operators the reference itself benchmarks (criterion): `explode` of 45,000 two-point MultiPoints
(geopolars/benches/explode.rs:10-24) and `translate(10, 10)` over the geometry column of data/cities.arrow
(geopolars/benches/affine.rs:23-26; 202 WKB points — the committed copy of that column is tests/golden/cities.npz).
Each is timed the way the reference's Series -> Series call would run through this backend:

  wkb_to_wkb    WKB column in host memory -> decode on the GPU -> operator -> encode on the GPU -> WKB column in host memory
  device_only   the operator alone on a device-resident handle (what a pipeline that keeps its columns in HBM pays)

plus `translate` at 10M points, where the kernel rather than the call overhead is measured.  One JSON line per case.

    python tools/bench_reference_benches.py > profiles/<round>_reference_benches.jsonl
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geopolars_amd import _abi, synth  # noqa: E402
from geopolars_amd.geoarrow import GeoArrowArray  # noqa: E402
from geopolars_amd.geoseries import GeoSeries  # noqa: E402


def timeit(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6  # microseconds per call


def main():
    lib = _abi.lib()
    name, _ = _abi.device_info()
    out = lambda **kw: print(json.dumps(dict(kw, device=name)), flush=True)

    # ---- explode: 45,000 MultiPoints of two points (0, 0) each --------------------------------------------------------
    n_mp = 45_000
    mp = GeoArrowArray(_abi.GEOM_MULTIPOINT, np.zeros((2 * n_mp, 2)), geom_offsets=np.arange(0, 2 * n_mp + 1, 2, dtype=np.int32))
    wkb_v, wkb_o = mp.to_wkb()

    def explode_wkb():
        s = GeoSeries.from_wkb_device(wkb_v, wkb_o)
        return s.explode().to_wkb()

    v, o = explode_wkb()
    assert len(o) == 2 * n_mp + 1 and len(v) == 2 * n_mp * 21  # 90,000 WKB points
    dev_series = GeoSeries(mp)
    dev_series.device()

    def explode_dev():
        return dev_series.explode()

    out(bench="explode", reference="geopolars/benches/explode.rs:10-24", workload=f"{n_mp} two-point MultiPoints", mode="wkb_to_wkb", us_per_call=timeit(explode_wkb, reps=100))
    out(bench="explode", reference="geopolars/benches/explode.rs:10-24", workload=f"{n_mp} two-point MultiPoints", mode="device_only", us_per_call=timeit(explode_dev))

    # ---- translate(10, 10): the cities column ---------------------------------------------------------------------------
    g = np.load(os.path.join(ROOT, "tests", "golden", "cities.npz"))
    cv, co = g["wkb_values"], g["wkb_offsets"]

    def translate_wkb():
        return GeoSeries.from_wkb_device(cv, co).translate(10.0, 10.0).to_wkb()

    tv, to = translate_wkb()
    got = GeoArrowArray.from_wkb(tv, to).xy
    assert np.array_equal(got, g["xy"] + 10.0)  # bit-exact: x * 1 + y * 0 + 10
    cities = GeoSeries(GeoArrowArray.from_points(g["xy"]))
    cities.device()
    out(bench="translate", reference="geopolars/benches/affine.rs:23-26", workload="data/cities.arrow geometry column (202 WKB points)", mode="wkb_to_wkb", us_per_call=timeit(translate_wkb, reps=100))
    out(bench="translate", reference="geopolars/benches/affine.rs:23-26", workload="data/cities.arrow geometry column (202 points)", mode="device_only",
        us_per_call=timeit(lambda: cities.translate(10.0, 10.0)))

    # ---- translate at a size where the kernel is what is measured ---------------------------------------------------------
    n = 10_000_000
    xy = torch.from_numpy(synth.uniform_points(n).xy).cuda()
    outxy = torch.empty_like(xy)
    from geopolars_amd.geoarrow import DeviceGeoArray

    d = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy)
    m6 = (C.c_double * 6)(1.0, 0.0, 10.0, 0.0, 1.0, 10.0)
    stream = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: _abi.check(lib.gpk_affine_transform(d.handle, m6, outxy.data_ptr(), _abi.MEM_DEVICE, stream)), reps=50, warm=5)
    out(bench="translate", reference="geopolars/benches/affine.rs:23-26", workload=f"{n} points, device-resident in and out", mode="device_only", us_per_call=us,
        GBps=32 * n / (us * 1e-6) / 1e9)


if __name__ == "__main__":
    main()
