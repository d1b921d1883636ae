cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "distance or c3 or rowmap or lineal" ) 2>&1 | grep -a "passed\|failed" | tail -2
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('c3', round(d['ms_per_step'],4), d['config'].get('kernel_ms_per_step'), d['roofline']['valu'], d['parity'])
"
