cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "ops or mixed or golden or fixture or structural or lineal or propert" ) 2>&1 | grep -a "passed\|failed" | tail -2
timeout 300 python tools/bench_ops.py 2>&1 | grep -a '"op"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['op'] in ('area','bounds','centroid','euclidean_length'): print(d['op'], '|', d['workload'][:40], '|', round(d['ms'],3), 'ms', round(d['GBps']/1000,2), 'TB/s')
"
