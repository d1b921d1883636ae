import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd import synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from oracle import pyoracle as O
ls = synth.random_linestrings(500, min_log2=1.0, max_log2=6.0)
xy, off = ls.xy, ls.geom_offsets
rng = np.random.default_rng(6)
rows, pts, kinds = [], [], []
for t in range(500):
    v = xy[off[t] : off[t + 1]]
    for _ in range(20):
        k = rng.integers(0, len(v) - 1); kind = rng.integers(0, 4)
        a, b = v[k], v[k + 1]
        pts.append(a if kind == 0 else ((a + b) / 2 if kind == 1 else (a + (b - a) * 0.25 if kind == 2 else a + np.array([1e-9, -1e-9]))))
        rows.append(t); kinds.append(kind)
pts = GeoArrowArray.from_points(np.array(pts)); rows = np.array(rows, dtype=np.uint32); kinds=np.array(kinds)
got = GeoSeries(pts).distance(GeoSeries(ls), other_rows=rows)
exp = O.distance_rowwise(pts, ls, rows)
bad = np.nonzero((got == 0) != (exp == 0))[0]
print("mismatches", len(bad), "by kind", np.bincount(kinds[bad], minlength=4), "exp zeros by kind", np.bincount(kinds[exp==0], minlength=4))
for i in bad[:8]: print(i, kinds[i], got[i], exp[i], rows[i])
# row-major kernel for comparison (fewer than 8 rows per target)
got2 = GeoSeries(pts).distance(GeoSeries(GeoArrowArray.concat([ls, synth.random_linestrings(3000, seed=9)])), other_rows=rows)
bad2 = np.nonzero((got2 == 0) != (exp == 0))[0]
print("row-major mismatches", len(bad2))
