# record heads + fast labelling path; slab entries as coordinate indices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03q
( GPK_SLAB_COPY_MAX_MB=0 timeout 900 python -m pytest tests -m gpu -x -q -k "join or chain or config or index or pip or assembly or contains or edge or mixed or propert" ) > gpurun_out/${T}_tests_vidx.log 2>&1
tail -3 gpurun_out/${T}_tests_vidx.log
( timeout 900 python -m pytest tests -m gpu -x -q -k "join or chain or config or index or pip or assembly or contains or edge or mixed or propert" ) > gpurun_out/${T}_tests.log 2>&1
tail -3 gpurun_out/${T}_tests.log
GPK_DEBUG_INDEX=1 timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5.log 2>&1
grep -a "index build: " gpurun_out/${T}_c5.log | sort | uniq | head -20
grep -a '^{' gpurun_out/${T}_c5.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('c5 step', d['ms_per_step'], 'join', c['join_ms_per_step'], 'idx ms', c['index_build_ms'], 'idx GB', c['index_bytes']/1e9, c['kernel_ms_per_step'], d['parity'])
"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('c2', d['ms_per_step'], d['config'].get('index_build_ms'), d['roofline'].get('launch_ms'))
"
