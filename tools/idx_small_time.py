#!/usr/bin/env python3
"""Diagnosis: phases of the index build for the C2 right side (1000 x 64-vertex polygons), GPK_DEBUG_INDEX stamps + totals."""
import os, sys, time
os.environ["GPK_DEBUG_INDEX"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex
stream = torch.cuda.current_stream().cuda_stream
polys = DeviceGeoArray.upload(synth.star_polygons(1000, 64), stream=stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = SpatialIndex.from_device(polys, stream=stream)
    torch.cuda.synchronize(); print(f"== build {rep}: {1e3*(time.perf_counter()-t0):.3f} ms", file=sys.stderr, flush=True)
    del idx
os.environ.pop("GPK_DEBUG_INDEX")
