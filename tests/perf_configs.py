#!/usr/bin/env python3
"""Single-GPU shares of BASELINE.json configs[3] (C4) and configs[4] (C5) at a chosen scale: wall times of
index build and join through the host-buffer API, with a parity spot check against the CPU oracle on a
prefix (which is why it lives under tests/: the oracle is test infrastructure).  Not the driver's bench; numbers go to
DESIGN.md / profiles/."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import synth
from geopolars_amd.dist import slice_rows
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs
from oracle import pyoracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--c4", type=int, default=1_000_000)
ap.add_argument("--c5-points", type=int, default=6_250_000)
ap.add_argument("--c5-polys", type=int, default=1_000_000)
ap.add_argument("--only", choices=["c4", "c5", "tess", "contains"], default=None)
ap.add_argument("--districts", type=int, default=100_000)
ap.add_argument("--parcels", type=int, default=1_000_000)
a = ap.parse_args()

def t(f):
    t0 = time.perf_counter(); r = f(); return r, (time.perf_counter() - t0) * 1e3

import ctypes as C
from geopolars_amd._abi import lib as _lib
def kernel_ms(f, names):
    """Per-kernel device time (ms) of one call of f, from the library's event profiler."""
    L = _lib()
    L.gpk_profile_reset(); L.gpk_profile_enable(1)
    f()
    L.gpk_profile_enable(0)
    out = {}
    for nm in names:
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.gpk_profile_query(nm.encode(), C.byref(ms), C.byref(cnt))
        if cnt.value: out[nm] = ms.value
    L.gpk_profile_reset()
    return out

# ---- contains join: districts (8-64 vertices) x parcels (4-12 vertices, ~50x smaller) — spatial_index.rs:99-101 ----------
if a.only == "contains":
    D = synth.clustered_polygons(a.districts, seed=71, mean_neighbours=2.0)
    Pc = synth.clustered_polygons(a.parcels, seed=72, mean_neighbours=0.02, min_verts=4, max_verts=12)
    ds, pcs = GeoSeries(D), GeoSeries(Pc)
    ds.device(); pcs.device()
    idx, ms_idx = t(lambda: SpatialIndex(pcs))
    (pairs, counts), ms_join = t(lambda: join_pairs(ds, pcs, "contains", r_index=idx))
    (pairs, counts), ms_join2 = t(lambda: join_pairs(ds, pcs, "contains", r_index=idx))
    kms = kernel_ms(lambda: join_pairs(ds, pcs, "contains", r_index=idx), ["gpk_bbox_cand_count", "gpk_bbox_cand_fill", "gpk_pair_contains", "gpk_pair_count", "gpk_pair_emit"])
    (ipairs, _), ms_int = t(lambda: join_pairs(ds, pcs, "intersects", r_index=idx))
    k = min(a.districts, 5000)
    ep, ec, _ = O.spatial_join(slice_rows(D, 0, k), Pc, "contains", mode=1)
    ok = np.array_equal(counts[:k], ec) and np.array_equal(pairs[: len(ep)], ep)
    print(json.dumps({"config": "contains join", "left": a.districts, "right": a.parcels, "index_build_ms": ms_idx, "join_ms_first": ms_join, "join_ms": ms_join2, "kernel_ms": kms, "pairs": int(len(pairs)), "intersects_pairs": int(len(ipairs)), "intersects_join_ms": ms_int, "parity_prefix_rows": k, "parity": bool(ok)}), flush=True)
    sys.exit(0)
# ---- tessellation: 10M points in 1024 polygons that share every border (administrative-boundary shape) -------------------
if a.only in (None, "tess"):
    T = synth.tessellation(32, 16); P = synth.uniform_points(10_000_000, seed=61)
    ts, ps = GeoSeries(T), GeoSeries(P)
    ts.device(); ps.device()
    idx, ms_idx = t(lambda: SpatialIndex(ts))
    kms = kernel_ms(lambda: join_pairs(ps, ts, "intersects", r_index=idx), ["gpk_pip_tile", "gpk_pip_write"])
    (pairs, counts), ms_join = t(lambda: join_pairs(ps, ts, "intersects", r_index=idx))
    k = 300_000
    ep, ec, _ = O.spatial_join(slice_rows(P, 0, k), T, "intersects", mode=1)
    ok = np.array_equal(counts[:k], ec) and np.array_equal(pairs[: len(ep)], ep)
    print(json.dumps({"config": "tessellation", "points": len(P), "polygons": len(T), "coords": int(T.n_coords), "index_build_ms": ms_idx, "join_ms": ms_join, "kernel_ms": kms, "pairs": int(len(pairs)), "parity_prefix_rows": k, "parity": bool(ok)}), flush=True)
    del ts, ps, idx
    if a.only == "tess":
        sys.exit(0)
# ---- C4: polygon x polygon intersects join ------------------------------------------------------------
if a.only == "c5":
    a.c4 = 1000
L = synth.clustered_polygons(a.c4, seed=41, mean_neighbours=4.0); R = synth.clustered_polygons(a.c4, seed=42, mean_neighbours=4.0)
ls, rs = GeoSeries(L), GeoSeries(R)
ls.device(); rs.device()
idx, ms_idx = t(lambda: SpatialIndex(rs))
(pairs, counts), ms_join = t(lambda: join_pairs(ls, rs, "intersects", r_index=idx))
(pairs, counts), ms_join2 = t(lambda: join_pairs(ls, rs, "intersects", r_index=idx))
kms4 = kernel_ms(lambda: join_pairs(ls, rs, "intersects", r_index=idx), ["gpk_bbox_cand_count", "gpk_bbox_cand_fill", "gpk_pair_refine", "gpk_pair_count", "gpk_pair_emit"])
k = min(a.c4, 20000)
ep, ec, _ = O.spatial_join(slice_rows(L, 0, k), R, "intersects", mode=1)
ok = np.array_equal(counts[:k], ec) and np.array_equal(pairs[: len(ep)], ep)
print(json.dumps({"config": "C4 share", "left": a.c4, "right": a.c4, "index_build_ms": ms_idx, "join_ms_first": ms_join, "join_ms": ms_join2, "kernel_ms": kms4, "pairs": int(len(pairs)), "parity_prefix_rows": k, "parity": bool(ok)}), flush=True)
del ls, rs, idx

# ---- C5: points within power-law multipolygons + area ---------------------------------------------------
if a.only == "c4":
    sys.exit(0)
MP = synth.powerlaw_multipolygons(a.c5_polys, seed=51); P = synth.uniform_points(a.c5_points, seed=52)
ms_, ps = GeoSeries(MP), GeoSeries(P)
ms_.device(); ps.device()
idx, ms_idx = t(lambda: SpatialIndex(ms_))
(pairs, counts), ms_join = t(lambda: join_pairs(ps, ms_, "within", r_index=idx))
(pairs, counts), ms_join2 = t(lambda: join_pairs(ps, ms_, "within", r_index=idx))
kms = kernel_ms(lambda: join_pairs(ps, ms_, "within", r_index=idx), ["gpk_pip_tile", "gpk_pip_write"])
area, ms_area = t(lambda: ms_.area())
area, ms_area2 = t(lambda: ms_.area())
k = min(a.c5_points, 200000)
ep, ec, _ = O.spatial_join(slice_rows(P, 0, k), MP, "within", mode=1)
ok = np.array_equal(counts[:k], ec) and np.array_equal(pairs[: len(ep)], ep)
ea = O.area(MP)
ok_area = bool(np.all(np.abs(area - ea) <= 1e-9 * np.maximum(np.abs(ea), 1e-300)))
print(json.dumps({"config": "C5 share", "points": a.c5_points, "multipolygons": a.c5_polys, "coords": int(MP.n_coords), "index_build_ms": ms_idx, "index_bytes": idx.nbytes(), "join_ms_first": ms_join, "join_ms": ms_join2, "kernel_ms": kms, "pairs": int(len(pairs)), "area_ms": ms_area2, "parity_prefix_rows": k, "parity": bool(ok), "area_parity": ok_area}), flush=True)
