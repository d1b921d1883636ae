cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('c2', round(d['ms_per_step'],4), d['roofline'].get('launch_ms'), d['config'].get('kernel_ms'))
"
done
timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('c2 with cpu baseline', round(d['ms_per_step'],4), d['roofline'].get('launch_ms'))
"
rocm-smi --showclocks 2>/dev/null | head -20
