#!/usr/bin/env python3
"""Probe: what bounds gpk_distance on C3?  Same kernel, row maps with different working sets."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from tools.bench_ops import dev_array, timed
lib = _abi.lib(); dev = torch.device("cuda", 0)
npts, nls = 10_000_000, 100_000
ls = synth.random_linestrings(nls); pts = synth.uniform_points(npts)
dl, dp = dev_array(ls, dev), dev_array(pts, dev)
out = torch.empty(npts, dtype=torch.float64, device=dev)
for label, rows in (("i mod 100000", np.arange(npts) % nls), ("i mod 1000 (L2-resident)", np.arange(npts) % 1000), ("i // 100 (each linestring 100 consecutive rows)", np.arange(npts) // 100)):
    r = torch.from_numpy(rows.astype(np.int32)).to(dev)
    ms, k = timed(lib, lambda s: _abi.check(lib.gpk_distance_rowwise(dp.handle, dl.handle, r.data_ptr(), out.data_ptr(), _abi.MEM_DEVICE, s)), reps=5)
    lib.gpk_profile_enable(1)
    _abi.check(lib.gpk_distance_rowwise(dp.handle, dl.handle, r.data_ptr(), out.data_ptr(), _abi.MEM_DEVICE, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); lib.gpk_profile_enable(0)
    parts = {}
    for name in (b"gpk_dist_hist", b"gpk_dist_scatter", b"gpk_dist_batches", b"gpk_scan", b"gpk_distance_grouped", b"gpk_distance"):
        m, c = C.c_double(0), C.c_int64(0); lib.gpk_profile_query(name, C.byref(m), C.byref(c)); parts[name.decode()] = round(m.value, 3)
    lib.gpk_profile_reset()
    print(label, round(ms, 3), "ms (sum of kernels)", parts)
