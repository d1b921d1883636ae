"""what one index build for C2's right side (1000 x 64-vertex polygons, the default call shape's per-call work) launches and waits for:
GPK_DEBUG_SYNC=1 names every launch on stderr; the wall time of a build without it"""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prog = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from geopolars_amd import synth
from geopolars_amd.geoarrow import DeviceGeoArray
from geopolars_amd.spatial_index import SpatialIndex
polys = DeviceGeoArray.upload(synth.star_polygons(1000, 64))
for i in range(3):
    SpatialIndex.from_device(polys, light=True).free()
torch.cuda.synchronize()
sys.stderr.write("==== build\n")
n = 1 if os.environ.get("GPK_DEBUG_SYNC") else 50
tb = tf = 0.0
for i in range(n):
    t0 = time.perf_counter()
    idx = SpatialIndex.from_device(polys, light=True)
    t1 = time.perf_counter()
    idx.free()
    t2 = time.perf_counter()
    tb += t1 - t0
    tf += t2 - t1
torch.cuda.synchronize()
print("build %%.3f ms + free %%.3f ms" %% (tb / n * 1e3, tf / n * 1e3))
''' % root
r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=dict(os.environ, GPK_DEBUG_SYNC="1"))
names = [l.split("launch ")[1] for l in r.stderr.split("==== build")[-1].splitlines() if "launch " in l]
from collections import Counter
print(len(names), "launches:", dict(Counter(names)))
r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True)
print(r.stdout.strip(), r.stderr[-300:] if r.returncode else "")
