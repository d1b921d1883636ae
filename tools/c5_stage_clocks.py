#!/usr/bin/env python3
"""Diagnosis: where the work-groups of pip_tile_kernel (the general tile kernel) spend their time on the C5 workload, from a
GPK_TILE_TRACE build:   GPK_LIB_PATH=geopolars_amd/variants/trace.so python tools/c5_stage_clocks.py [--full]
Per stage: the summed wall-clock time of thread 0 of every work-group (barrier waits included) and its share."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.dist import GeoBuffers
from geopolars_amd.geoarrow import DeviceGeoArray, GeoArrowArray
from geopolars_amd.spatial_index import SpatialIndex, join_pairs_enqueue

lib = _abi.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n, M = 6_250_000, 5_000_000
host = GeoArrowArray.concat([synth.powerlaw_multipolygons(M // 8, seed=51 + k, size_n=M) for k in range(8)])
right = GeoBuffers.from_host(host, dev).to_device_geoarray(stream)
index = SpatialIndex.from_device(right, stream=stream, full="--full" in sys.argv)
xy = torch.from_numpy(synth.uniform_points(n, seed=52).xy).to(dev)
pts = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=stream)
counts = torch.empty(n, dtype=torch.int32, device=dev)
pairs = torch.empty((4 * n, 2), dtype=torch.int32, device=dev)
total = torch.zeros(1, dtype=torch.int64, device=dev)
for i in range(3):
    join_pairs_enqueue(pts, right, index, "within", counts, pairs, total, stream=stream)
torch.cuda.synchronize()
lib.gpk_join_stats_enable(1)
join_pairs_enqueue(pts, right, index, "within", counts, pairs, total, stream=stream)
torch.cuda.synchronize()
lib.gpk_join_stats_enable(0)
buf = (C.c_uint64 * 16)()
lib.gpk_join_trace.argtypes = [C.c_void_p, C.c_int64]
assert lib.gpk_join_trace(buf, 16) == 0
t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[:6] * 10.0  # 100 MHz ticks -> ns
names = ["points + level-1 words + records (stages A-D)", "pushes + entry lists (+ barrier)", "phase 2: scan of the queue", "phase 2: flattened edge pass",
         "phase 2: decide + holes (+ barrier)", "finalize (codes, pool, stores)"]
tile_points = 256 if index.describe()["list_heavy"] else 512
tiles = (n + tile_points - 1) // tile_points
print(f"{tiles} work-groups; summed critical path {t.sum() / 1e6:.1f} ms = {t.sum() / tiles / 1e3:.1f} us per work-group")
for nm, v in zip(names, t):
    print(f"  {nm:50s} {v / tiles / 1e3:7.2f} us per work-group   {100 * v / t.sum():5.1f} %")
