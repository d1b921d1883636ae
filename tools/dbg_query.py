import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex
from oracle import pyoracle
pyoracle.build()
rng = np.random.default_rng(11)
mps = []
for i in range(1500):
    parts = []
    for _ in range(int(rng.integers(0, 4))):
        x, y, w = rng.uniform(0, 990), rng.uniform(0, 990), rng.uniform(0.5, 9.0)
        parts.append([[(x, y), (x + w, y), (x + w, y + w), (x, y + w)]])
    mps.append(parts)
arr = GeoArrowArray.from_multipolygons(mps)
arr.validity = np.packbits(rng.uniform(size=len(arr)) > 0.1, bitorder="little")
c = rng.uniform(0, 1000, (300, 2)); half = rng.uniform(0, 200, (300, 2))
boxes = np.column_stack([c - half, c + half])
index = SpatialIndex(GeoSeries(arr), for_points=False)
b = pyoracle.bounds(arr)
for mode in ("contained", "intersecting"):
    ep, ec = pyoracle.envelope_query(arr, boxes, mode)
    gp, gc = index.query_envelopes(boxes, mode)
    bad = np.nonzero(gc != ec)[0]
    print(mode, "bad queries", len(bad), bad[:5])
    for q in bad[:3]:
        e = set(ep[ep[:, 0] == q][:, 1].tolist()); g = set(gp[gp[:, 0] == q][:, 1].tolist())
        print(" q", q, boxes[q], "missing", sorted(e - g)[:5], "extra", sorted(g - e)[:5])
        for j in sorted(e ^ g)[:3]:
            print("   leaf", j, b[j], "valid", bool(arr.is_valid()[j]))
