#!/usr/bin/env python3
"""Tuning: per-kernel HIP-event times of gpk_geoarray_from_wkb / gpk_geoarray_to_wkb (device buffers) on the three column shapes of
tools/bench_ops.py.   python tools/wkb_time.py [--only powerlaw|small|mid]"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from tools.bench_ops import dev_array

ap = argparse.ArgumentParser(); ap.add_argument("--only", default=""); a = ap.parse_args()
lib = _abi.lib(); dev = torch.device("cuda", 0); MEM_DEVICE = _abi.MEM_DEVICE
shapes = {"mid": lambda: synth.star_polygons(2_000_000, 64), "small": lambda: synth.star_polygons(8_000_000, 8), "powerlaw": lambda: synth.powerlaw_multipolygons(1_000_000)}
NAMES = [b"gpk_wkb_scan", b"gpk_wkb_extent", b"gpk_scan", b"gpk_wkb_fill", b"gpk_wkb_copy_long", b"gpk_wkb_copy", b"gpk_wkb_enc", b"gpk_wkb"]
for key, make in shapes.items():
    if a.only and key != a.only: continue
    host = make(); d = dev_array(host, dev); n = len(host)
    nb = C.c_int64(0)
    woff = torch.empty(n + 1, dtype=torch.int32, device=dev)
    _abi.check(lib.gpk_geoarray_to_wkb(d.handle, None, None, 0, C.byref(nb), MEM_DEVICE, None))
    wkb = torch.empty(int(nb.value), dtype=torch.uint8, device=dev)
    def enc(s): _abi.check(lib.gpk_geoarray_to_wkb(d.handle, woff.data_ptr(), wkb.data_ptr(), int(nb.value), C.byref(nb), MEM_DEVICE, s))
    def dec(s):
        out = C.c_void_p(); gt = C.c_int32(-1)
        _abi.check(lib.gpk_geoarray_from_wkb(wkb.data_ptr(), woff.data_ptr(), n, None, MEM_DEVICE, s, C.byref(out), C.byref(gt)))
        lib.gpk_geoarray_free(out)
    alg = int(nb.value) + 4 * n + 16 * host.n_coords + 4 * (host.n_geoms + host.n_parts + host.n_rings + 3)
    for label, fn in (("to_wkb", enc), ("from_wkb", dec)):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2): fn(s)
        torch.cuda.synchronize(); lib.gpk_profile_reset(); lib.gpk_profile_enable(1)
        for _ in range(5): fn(s)
        torch.cuda.synchronize(); lib.gpk_profile_enable(0)
        parts = {}
        for nm in NAMES:
            m, c = C.c_double(0), C.c_int64(0); lib.gpk_profile_query(nm, C.byref(m), C.byref(c))
            if c.value: parts[nm.decode()] = round(m.value / 5, 4)
        lib.gpk_profile_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn(s)
        e1.record(); torch.cuda.synchronize()
        tot = parts.get("gpk_wkb", 0.0)
        print(f"{key:9s} {label:9s} kernels {tot:.3f} ms = {alg / tot / 1e6 if tot else 0:.0f} GB/s | wall {e0.elapsed_time(e1) / 5:.3f} ms | {parts}", flush=True)
    del d, wkb, woff
