# slab entries as coordinate indices: parity with the form forced everywhere, then C5 either way
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03p
( GPK_SLAB_COPY_MAX_MB=0 timeout 900 python -m pytest tests -m gpu -x -q -k "join or chain or config or index or pip" ) > gpurun_out/${T}_tests_vidx.log 2>&1
tail -4 gpurun_out/${T}_tests_vidx.log
for mode in default copies; do
  if [ $mode = copies ]; then export GPK_SLAB_COPY_MAX_MB=100000; fi
  GPK_DEBUG_INDEX=1 timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5_$mode.log 2>&1
  grep -a "index build: " gpurun_out/${T}_c5_$mode.log | sort | uniq | head -20
  grep -a '^{' gpurun_out/${T}_c5_$mode.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('$mode', 'step', d['ms_per_step'], 'join', c['join_ms_per_step'], 'idx ms', c['index_build_ms'], 'idx GB', c['index_bytes']/1e9, c['kernel_ms_per_step'], d['parity'])
"
done
unset GPK_SLAB_COPY_MAX_MB
GPK_SLAB_COPY_MAX_MB=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('c2 with index slabs', d['ms_per_step'], d['config'].get('kernel_ms_per_step'))
"
