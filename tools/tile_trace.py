#!/usr/bin/env python3
"""Diagnosis: stage times of the lean tile kernel from a GPK_TILE_TRACE build (GPK_LIB_PATH=.../variants/trace.so).
Prints, per stage, the median / mean duration over the sampled tiles, and the tile lifetime."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex, join_pairs_enqueue
lib = _abi.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = 10_000_000
polys = DeviceGeoArray.upload(synth.star_polygons(1000, 64), stream=stream)
index = SpatialIndex.from_device(polys, stream=stream)
sets = []
for r in range(3):
    xy = torch.from_numpy(synth.uniform_points(n, seed=77 + r).xy).to(dev)
    sets.append((DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=stream), torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 2), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)))
for i in range(4):
    p, c, pr, t = sets[i % 3]
    join_pairs_enqueue(p, polys, index, "intersects", c, pr, t, stream=stream)
torch.cuda.synchronize()
lib.gpk_join_stats_enable(1)
p, c, pr, t = sets[1]
join_pairs_enqueue(p, polys, index, "intersects", c, pr, t, stream=stream)
torch.cuda.synchronize()
lib.gpk_join_stats_enable(0)
W = 8 * 8000
buf = (C.c_uint64 * W)()
lib.gpk_join_trace.argtypes = [C.c_void_p, C.c_int64]
assert lib.gpk_join_trace(buf, W) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
print("sampled tiles", len(a))
names = ["entry->points landed (+barrier)", "level-1 words landed", "level-2 records landed", "decide+push+barrier", "exact phase", "barrier 2", "collect+finalize (stores issued)"]
d = np.diff(a, axis=1) * 10.0  # wall clock = 100 MHz -> ns
for i, nm in enumerate(names):
    print(f"{nm:40s} median {np.median(d[:, i]):8.0f} ns   mean {d[:, i].mean():8.0f} ns   p90 {np.percentile(d[:, i], 90):8.0f}")
life = (a[:, 7] - a[:, 0]) * 10.0
print(f"tile lifetime median {np.median(life):.0f} ns mean {life.mean():.0f} ns; kernel span {(a[:, 7].max() - a[:, 0].min()) * 10.0 / 1e3:.1f} us")
start = (a[:, 0] - a[:, 0].min()) * 10.0 / 1e3
print("tile start times (us) percentiles 10/50/90:", np.percentile(start, [10, 50, 90]))
