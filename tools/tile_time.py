#!/usr/bin/env python3
"""Tuning: HIP-event time of the kernels of the C2 join (gpk_join_prep / gpk_pip_tile / gpk_pip_write) on cold rotating inputs,
plus the wall time of a queued step, for the build named by GPK_LIB_PATH (default: the in-tree library).  No parity, no bench line
(bench.py is the measurement of record); environment switches of the library apply (GPK_NO_CHAINS=1: queue kernel on an index
without chains, GPK_TILE_KERNEL=chain: the chain kernel instead of the routed one serves an index with chains).
    GPK_LIB_PATH=geopolars_amd/variants/r2.so python tools/tile_time.py [--polys 1000] [--points 10000000]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex, join_pairs_enqueue

ap = argparse.ArgumentParser()
ap.add_argument("--polys", type=int, default=1000)
ap.add_argument("--verts", type=int, default=64)
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--tag", default="")
ap.add_argument("--frac", type=float, default=1.0, help="points uniform over this fraction of the domain's side (locality experiment: the index lines touched shrink with its square)")
a = ap.parse_args()
lib = _abi.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = a.points
polys = DeviceGeoArray.upload(synth.star_polygons(a.polys, a.verts), stream=stream)
index = SpatialIndex.from_device(polys, stream=stream)
sets = []
for r in range(3):
    xy = torch.from_numpy(synth.uniform_points(n, seed=77 + r, domain=synth.DOMAIN * a.frac).xy).to(dev)
    sets.append((DeviceGeoArray.from_device_buffers(_abi.GEOM_POINT, xy, stream=stream), torch.empty(n, dtype=torch.int32, device=dev),
                 torch.empty((n, 2), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)))
def step(i):
    p, c, pr, t = sets[i % 3]
    join_pairs_enqueue(p, polys, index, "intersects", c, pr, t, stream=stream)
for i in range(4):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    step(4 + i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e6
lib.gpk_profile_reset(); lib.gpk_profile_enable(1)
for i in range(a.steps):
    step(4 + i)
torch.cuda.synchronize()
lib.gpk_profile_enable(0)
out = []
for name in (b"gpk_pip_tile", b"gpk_join_prep", b"gpk_pip_write"):
    ms, cnt = C.c_double(0), C.c_int64(0)
    lib.gpk_profile_query(name, C.byref(ms), C.byref(cnt))
    out.append(f"{name.decode()[4:]} {1e3 * ms.value / max(cnt.value, 1):.1f}")
st = (C.c_int64 * 4)()
lib.gpk_join_stats_enable(1); lib.gpk_join_stats(st, 1); step(0); lib.gpk_join_stats(st, 1); lib.gpk_join_stats_enable(0)
d = index.describe()
name = a.tag or os.path.basename(os.environ.get("GPK_LIB_PATH", "base"))
print(f"{name:14s} step {wall:6.1f} us | " + " ".join(out) + f" | hits {int(sets[0][3].item())} exact {int(st[0])} edges {int(st[1])} deferred {int(st[2])} | R {d['R']} chains {int(d['chains'])} route {int(d['route'])} index {index.nbytes() / 1e6:.1f} MB", flush=True)
