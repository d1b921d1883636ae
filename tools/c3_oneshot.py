#!/usr/bin/env python3
"""Tuning: C3 row-map build and one-shot distance wall times (ms).   python tools/c3_oneshot.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.geoseries import RowMap
from tools.bench_ops import dev_array
lib = _abi.lib(); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
n, L = 10_000_000, 100_000
ls = dev_array(synth.random_linestrings(L), dev); pts = dev_array(synth.uniform_points(n, seed=3), dev)
rows = (np.arange(n, dtype=np.uint32) % L).astype(np.uint32)
out = torch.empty(n, dtype=torch.float64, device=dev)
for label, r in (("i mod L", rows), ("shuffled", np.random.default_rng(1).permutation(rows))):
    r_dev = torch.from_numpy(r.view(np.int32)).to(dev)
    b, o = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = RowMap.from_device(ls, r_dev, stream=stream); torch.cuda.synchronize(); b.append((time.perf_counter() - t0) * 1e3); m.free()
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _abi.check(lib.gpk_distance_rowwise(pts.handle, ls.handle, r_dev.data_ptr(), out.data_ptr(), _abi.MEM_DEVICE, stream)); torch.cuda.synchronize(); o.append((time.perf_counter() - t0) * 1e3)
    print(f"{label:9s} rowmap_build {min(b[1:]):.3f} ms  one_shot {min(o[1:]):.3f} ms", flush=True)
