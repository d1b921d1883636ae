"""debug: where do the fused join's pairs differ from the oracle's (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from geopolars_amd import synth
from geopolars_amd.geoarrow import GeoArrowArray
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs
from oracle import pyoracle
import test_gpu_fused as T
pyoracle.build()

def report(name, pts, polys, base=0):
    right = GeoSeries(polys); index = SpatialIndex(right)
    ep, ec, _ = pyoracle.spatial_join(pts, polys, "intersects", mode=0)
    gp, gc = join_pairs(GeoSeries(pts), right, "intersects", r_index=index, left_row_base=base)
    ep = ep.copy(); ep[:, 0] += base
    ok_c = np.array_equal(gc, ec); ok_p = gp.shape == ep.shape and np.array_equal(gp, ep)
    print(name, "counts", ok_c, "pairs", ok_p, gp.shape, ep.shape, flush=True)
    if not ok_p and gp.shape == ep.shape:
        bad = np.nonzero((gp != ep).any(axis=1))[0]
        print("  bad pairs", len(bad), "first", bad[:5], "last", bad[-5:])
        for b in bad[:6]:
            print("   at", b, "got", gp[b], "want", ep[b], "tile", ep[b][0] // 512, "count", ec[ep[b][0] - base])
        rows = np.unique(ep[bad, 0] - base)
        print("  rows' tiles", np.unique(rows // 512)[:20], "multi-hit tiles", np.unique(np.nonzero(ec > 1)[0] // 512)[:20])

which = sys.argv[1:] or ["stacked", "p100k", "p3m"]
if "tiny" in which:
    report("tiny-600", synth.uniform_points(600, seed=5), synth.star_polygons(1000, 64))
if "p100k" in which:
    report("plain-100k", synth.uniform_points(100_000, seed=5), synth.star_polygons(1000, 64))
if "p3m" in which:
    report("plain-3M", synth.uniform_points(3_000_000, seed=6), synth.star_polygons(1000, 64), base=77)
if "stacked" not in which:
    sys.exit(0)
polys = T._stacked(900, 37)
rng = np.random.default_rng(3)
inside = np.column_stack([rng.uniform(499.0, 505.0, 4000), rng.uniform(959.0, 965.0, 4000)])
pts = np.concatenate([synth.uniform_points(30_000, seed=4).xy, inside])
rng.shuffle(pts)
report("stacked37", GeoArrowArray.from_points(pts), polys)
