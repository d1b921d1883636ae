#!/usr/bin/env python3
"""Wall time of the index build for the C2 right side (1000 x 64-vertex polygons), no debug stamps: 20 builds, min / median."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex
stream = torch.cuda.current_stream().cuda_stream
polys = DeviceGeoArray.upload(synth.star_polygons(1000, 64), stream=stream)
ts = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = SpatialIndex.from_device(polys, stream=stream)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    idx.free()
ts.sort()
print(f"C2 index build: min {ts[0]:.3f} ms, median {ts[len(ts)//2]:.3f} ms, max {ts[-1]:.3f} ms over {len(ts)} builds", flush=True)
