# full-size parity tests + C5 raster-side sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03o
( time timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "full_size" ) > gpurun_out/${T}_fullsize.log 2>&1
tail -5 gpurun_out/${T}_fullsize.log
for r in 2048 4096 1024; do
  GPK_DEBUG_INDEX=1 GPK_PIP_RMAX=$r timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5_R$r.log 2>&1
  grep -a "level 2\|lean\|index build: " gpurun_out/${T}_c5_R$r.log | awk '{a[$0]++} END{for(k in a) print k}' | sort | head -40
  grep -a '^{' gpurun_out/${T}_c5_R$r.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('R$r', 'step', d['ms_per_step'], 'join', c['join_ms_per_step'], 'idx ms', c['index_build_ms'], 'idx GB', c['index_bytes']/1e9, c['kernel_ms_per_step'])
"
done
