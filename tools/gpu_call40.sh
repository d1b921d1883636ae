cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -k "wkb or fixture or golden or mixed" ) 2>&1 | grep -a "passed\|failed" | tail -2
timeout 300 python tools/bench_ops.py 2>&1 | grep -a '"op"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'wkb' in d['op']: print(d['op'], '|', d['workload'][:40], '|', round(d['ms'],3), 'ms', round(d['GBps']/1000,2), 'TB/s')
"
