cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03y
for v in plain sorted; do
  extra=""; if [ $v = sorted ]; then extra="--diag-sorted-points"; fi
  timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline $extra > gpurun_out/${T}_c5_$v.log 2>&1
  grep -a '^{' gpurun_out/${T}_c5_$v.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('$v step', round(d['ms_per_step'],3), 'tile', c['kernel_ms_per_step']['gpk_pip_tile'], 'write', c['kernel_ms_per_step']['gpk_pip_write'], d['parity']['bit_exact'], c['index_full_variant']['join_ms'])
"
done
