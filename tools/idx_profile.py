#!/usr/bin/env python3
"""Per-kernel time of one gpk_index_build over a C5-shaped column (tuning helper)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import _abi, synth
from geopolars_amd.dist import GeoBuffers
from geopolars_amd.spatial_index import SpatialIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "c5"
lib = _abi.lib(); dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
a = synth.powerlaw_multipolygons(n, seed=51) if kind == "c5" else (synth.star_polygons(n, 64) if kind == "c2" else synth.clustered_polygons(n, seed=42))
d = GeoBuffers.from_host(a, dev).to_device_geoarray(stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = SpatialIndex.from_device(d, stream=stream); torch.cuda.synchronize()
    print("build wall ms", (time.perf_counter() - t0) * 1e3, "bytes", ix.nbytes()); ix.free()
lib.gpk_profile_reset(); lib.gpk_profile_filter(b""); lib.gpk_profile_enable(1)
t0 = time.perf_counter(); ix = SpatialIndex.from_device(d, stream=stream); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
lib.gpk_profile_enable(0)
names = ["gpk_seq_bbox","gpk_bounds_combine","gpk_stats_to_bbox","gpk_index_extent","gpk_index_count","gpk_index_fill","gpk_index_sort","gpk_scan","gpk_one_ring_each","gpk_seq_classify","gpk_seq_long"] + ["gpk_pipidx_" + x for x in ("part_geom","ring_part","ring_rows","slab_count","slab_fill","part_info","mark_count","mark_fill","unique_flags","unique_compact","cell_count","cell_fill","sub_flag","sub_head","sub_work","sub_build","sub_build_fast","sub_build_rest","sub2_build","sub_commit","lrec_count","lrec_assign","lrec_build")]
tot = 0
for nm in names:
    ms, cnt = C.c_double(0), C.c_int64(0); lib.gpk_profile_query(nm.encode(), C.byref(ms), C.byref(cnt))
    if cnt.value: print(f"{nm:32s} {ms.value:8.3f} ms  x{cnt.value}"); tot += ms.value
print("sum of bracketed kernels", tot, "wall (profiled run)", wall)
