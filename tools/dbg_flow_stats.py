import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
from geopolars_amd import _abi, synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs
lib = _abi.lib()
def around(xy, radius, per, seed):
    rng = np.random.default_rng(seed)
    jit = rng.uniform(-radius, radius, (len(xy), per, 2))
    return np.concatenate([xy, (xy[:, None, :] + jit).reshape(-1, 2)])
polys_l = []
for i in range(20):
    for j in range(20):
        cx, cy = 50.0 * i + 25.0, 50.0 * j + 25.0
        ang = np.linspace(-np.pi / 2, 3 * np.pi / 2, 41)[1:-1]
        ring = [(cx - 0.001, cy - 20.0), (cx - 0.001, cy + 5.0), (cx + 0.001, cy + 5.0), (cx + 0.001, cy - 20.0)]
        ring += [(cx + 20.0 * np.cos(a), cy + 20.0 * np.sin(a)) for a in ang]
        polys_l.append([ring])
polys = GeoArrowArray.from_polygons(polys_l)
slit = np.array([[50.0 * i + 25.0, 50.0 * j + 5.0 + t] for i in range(20) for j in range(20) for t in (3.0, 7.3, 12.9, 24.0, 29.9995, 30.0005)])
pts = GeoArrowArray.from_points(np.concatenate([around(slit, 0.004, 30, 6), synth.uniform_points(20_000, seed=10).xy]))
right = GeoSeries(polys); index = SpatialIndex(right)
print(index.describe(), "n", len(pts.xy))
st = (C.c_int64 * 4)()
lib.gpk_join_stats_enable(1); lib.gpk_join_stats(st, 1)
got = join_pairs(GeoSeries(pts), right, "intersects", r_index=index)
lib.gpk_join_stats(st, 1)
print("pairs", len(got[0]), "stats", list(st))
