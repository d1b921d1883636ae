#!/usr/bin/env python3
"""Tuning: HIP-event time of gpk_pip_tile / gpk_pip_write for the C2 workload on cold rotating inputs, for builds whose
answers are wrong on purpose (GPK_ABLATE variants: select with GPK_LIB_PATH).  No parity, no bench line.
    GPK_LIB_PATH=geopolars_amd/variants/abl1.so python tools/ablate_time.py"""
import ctypes as C, os, sys
import torch
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
    sets.append((DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=stream), torch.empty(n, dtype=torch.int32, device=dev),
                 torch.empty((n, 2), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)))
def step(i):
    p, c, pr, t = sets[i % 3]
    join_pairs_enqueue(p, polys, index, "intersects", c, pr, t, stream=stream)
for i in range(4):
    step(i)
torch.cuda.synchronize()
lib.gpk_profile_reset(); lib.gpk_profile_enable(1)
for i in range(30):
    step(4 + i)
torch.cuda.synchronize()
lib.gpk_profile_enable(0)
out = []
for name in (b"gpk_pip_tile", b"gpk_pip_write"):
    ms, cnt = C.c_double(0), C.c_int64(0)
    lib.gpk_profile_query(name, C.byref(ms), C.byref(cnt))
    out.append(f"{name.decode()} {1e3 * ms.value / max(cnt.value, 1):.1f} us")
print(os.path.basename(os.environ.get("GPK_LIB_PATH", "base")), " ".join(out), "hits", int(sets[0][3].item()))
