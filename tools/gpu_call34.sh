cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03h
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/${T}_tests.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests.log | tail -3
python tools/idx_small_plain.py 2>&1 | tail -1
timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5.log 2>&1
grep -a '^{' gpurun_out/${T}_c5.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('c5 step', round(d['ms_per_step'],3), 'idx ms', round(c['index_build_ms'],1), round(c['index_build_again_ms'],1), 'idx GB', c['index_bytes']/1e9, 'tile', c['kernel_ms_per_step']['gpk_pip_tile'], d['parity']['bit_exact'], c['index_full_variant'])
"
timeout 400 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -a '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']; print('c4', round(d['ms_per_step'],3), c['index_build_ms'])
"
timeout 300 python tests/stress_sweep.py 2>&1 | tail -2
