#!/usr/bin/env python3
"""Diagnosis: where the wall time of an index build goes (GPK_DEBUG_INDEX=1 prints the phases), and what hipMalloc /
hipFree of large buffers cost on this box.   python tools/idx_build_time.py [n_multipolygons]"""
import ctypes as C, os, sys, time
os.environ["GPK_DEBUG_INDEX"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.dist import GeoBuffers
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.spatial_index import SpatialIndex

hip = C.CDLL("libamdhip64.so")
for mb in (64, 512, 2048):
    p = C.c_void_p()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), C.c_size_t(mb << 20)); t1 = time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(mb << 20)); torch.cuda.synchronize(); t2 = time.perf_counter()
    hip.hipFree(p); t3 = time.perf_counter()
    print(f"hipMalloc {mb} MB: {1e3*(t1-t0):.3f} ms, first memset {1e3*(t2-t1):.3f} ms, hipFree {1e3*(t3-t2):.3f} ms", flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
chunks = max(1, n // 625_000)
host = GeoArrowArray.concat([synth.powerlaw_multipolygons(n // chunks, seed=51 + k, size_n=n) for k in range(chunks)])
right = GeoBuffers.from_host(host, dev).to_device_geoarray(stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = SpatialIndex.from_device(right, stream=stream)
    torch.cuda.synchronize(); print(f"== build {rep}: {1e3*(time.perf_counter()-t0):.1f} ms, {idx.nbytes()/1e9:.2f} GB", file=sys.stderr, flush=True)
    del idx
