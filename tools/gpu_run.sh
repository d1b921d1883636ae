R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q -k "join or fixtures or assembly or edge" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python - <<'PY'
import time, numpy as np
from geopolars_amd import synth
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd.spatial_index import SpatialIndex, join_pairs
mp = synth.powerlaw_multipolygons(1_000_000, seed=51); pts = synth.uniform_points(6_250_000, seed=52)
ms_, ps = GeoSeries(mp), GeoSeries(pts); ms_.device(); ps.device()
for rep in range(3):
    t0=time.perf_counter(); p,c = join_pairs(ps, ms_, "within"); t1=time.perf_counter()
    print("one-shot join (light index built inside) ms", (t1-t0)*1e3, len(p))
t0=time.perf_counter(); ix=SpatialIndex(ms_); t1=time.perf_counter(); print("full index build ms",(t1-t0)*1e3, ix.nbytes())
for rep in range(2):
    t0=time.perf_counter(); p2,c2 = join_pairs(ps, ms_, "within", r_index=ix); t1=time.perf_counter(); print("join with prebuilt index ms",(t1-t0)*1e3)
assert np.array_equal(p,p2) and np.array_equal(c,c2)
PY
