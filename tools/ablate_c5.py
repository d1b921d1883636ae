#!/usr/bin/env python3
"""Tuning: HIP-event time of gpk_pip_tile for the C5 workload (6.25M points within 5M power-law multipolygons) for builds whose
answers are wrong on purpose (GPK_ABLATE variants: select with GPK_LIB_PATH).  No parity, no bench line.
    GPK_LIB_PATH=geopolars_amd/variants/abl1.so python tools/ablate_c5.py"""
import ctypes as C, os, sys
import torch
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
index = SpatialIndex.from_device(right, stream=stream)
xy = torch.from_numpy(synth.uniform_points(n, seed=52).xy).to(dev)
pts = DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=stream)
counts = torch.empty(n, dtype=torch.int32, device=dev)
pairs = torch.empty((4 * n, 2), dtype=torch.int32, device=dev)
total = torch.zeros(1, dtype=torch.int64, device=dev)
for i in range(3):
    join_pairs_enqueue(pts, right, index, "within", counts, pairs, total, stream=stream)
torch.cuda.synchronize()
lib.gpk_profile_reset(); lib.gpk_profile_enable(1)
for i in range(8):
    join_pairs_enqueue(pts, right, index, "within", counts, pairs, total, stream=stream)
torch.cuda.synchronize()
lib.gpk_profile_enable(0)
out = []
for name in (b"gpk_pip_tile", b"gpk_pip_write"):
    ms, cnt = C.c_double(0), C.c_int64(0)
    lib.gpk_profile_query(name, C.byref(ms), C.byref(cnt))
    out.append(f"{name.decode()} {1e3 * ms.value / max(cnt.value, 1):.1f} us")
print(os.path.basename(os.environ.get("GPK_LIB_PATH", "base")), " ".join(out), "hits", int(total.item()), flush=True)
