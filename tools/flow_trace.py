#!/usr/bin/env python3
"""Diagnosis: the timeline of pip_tile_flow_kernel (gpk_pipflow.hip) from a GPK_TILE_TRACE build.
    GPK_LIB_PATH=geopolars_amd/variants/trace1.so python tools/flow_trace.py [n]     stage stamps (us after the first wave's entry)
    GPK_LIB_PATH=geopolars_amd/variants/trace2.so python tools/flow_trace.py [n]     + a wave's tile time summed by phase
Stamps (100 MHz wall clock): 0 entry, 1 image built, 2 first tile starts, 3 .. 8 after tile i (trace1) / phase sums (trace2),
9 tiles done + list empty, 10 work-group barrier, 11 place known, 12 pairs issued."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex, join_pairs_enqueue
lib = _abi.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
phases = "trace2" in os.environ.get("GPK_LIB_PATH", "")
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
lib.gpk_join_trace.argtypes = [C.c_void_p, C.c_int64]
W = 8 * 8000
buf = (C.c_uint64 * W)()
lib.gpk_join_stats_enable(1)
lib.gpk_join_trace(buf, W)  # clears
for rep in range(2):
    p, c, pr, t = sets[(1 + rep) % 3]
    join_pairs_enqueue(p, polys, index, "intersects", c, pr, t, stream=stream)
    torch.cuda.synchronize()
    assert lib.gpk_join_trace(buf, W) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    stamps = [0, 1, 2, 9, 10, 11, 12] if phases else [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]
    print(f"rep {rep}: traced waves {len(a)}; kernel span (first entry -> last stamp) {us(a[:, stamps].max()):.1f} us")
    names = {0: "entry", 1: "image built", 2: "first tile starts", 3: "tile 0 done", 4: "tile 1 done", 5: "tile 2 done", 6: "tile 3 done", 7: "tile 4 done", 9: "tiles done",
             10: "wg barrier", 11: "place known", 12: "pairs issued"}
    for i in stamps:
        col = a[:, i]
        col = col[col > 0]
        if len(col):
            v = us(col)
            print(f"  {names[i]:18s} n {len(col):5d}  min {v.min():6.1f}  p10 {np.percentile(v, 10):6.1f}  median {np.median(v):6.1f}  p90 {np.percentile(v, 90):6.1f}  max {v.max():6.1f}")
    if not phases and len(a) >= 1024:  # where the late work-groups sit: barrier time by XCD (blockIdx % 8: the dispatcher deals work-groups round the XCDs)
        full = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 16).astype(np.int64)[: 256 * 4].reshape(256, 4, 16)
        bar = (full[:, :, 10].max(axis=1) - t0) / 100.0
        done = (full[:, :, 9].max(axis=1) - t0) / 100.0
        ent = (full[:, :, 0].min(axis=1) - t0) / 100.0
        print("  work-group barrier by XCD (blockIdx % 8): " + "  ".join(f"{x}: {np.median(bar[x::8]):.1f} / {bar[x::8].max():.1f}" for x in range(8)))
        print("  work-group entry by XCD:                  " + "  ".join(f"{x}: {np.median(ent[x::8]):.1f}" for x in range(8)))
        order = np.argsort(bar)
        print("  the 12 latest work-groups (blockIdx: barrier us): " + ", ".join(f"{int(b)}: {bar[b]:.1f}" for b in order[-12:]))
        print("  barrier by blockIdx quartile: " + "  ".join(f"{np.median(bar[q * 64:(q + 1) * 64]):.1f}" for q in range(4)))
    if phases:
        for i, nm in enumerate(["list walked", "points arrived", "routed + records arrived", "ranked + stored", "drawn", "final flush"]):
            v = a[:, 3 + i] / 100.0
            print(f"  sum over a wave's tiles: {nm:26s} median {np.median(v):6.2f} us  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
lib.gpk_join_stats_enable(0)
