cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03s
( GPK_LIST_RECORDS=0 timeout 900 python -m pytest tests -m gpu -x -q -k "join or chain or config or index or pip or assembly or contains or edge or mixed or propert" ) > gpurun_out/${T}_tests_nolrec.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests_nolrec.log | tail -3
( timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q ) > gpurun_out/${T}_tests_cfg.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests_cfg.log | tail -3
run() { # name, env...
  name=$1; shift
  env "$@" GPK_DEBUG_INDEX=1 timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5_$name.log 2>&1
  grep -a '^{' gpurun_out/${T}_c5_$name.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('$name', 'step', round(d['ms_per_step'],3), 'join', round(c['join_ms_per_step'],3), 'idx ms', round(c['index_build_ms'],1), 'idx GB', c['index_bytes']/1e9, 'tile', c['kernel_ms_per_step']['gpk_pip_tile'], d['parity']['bit_exact'], c['exact_phase'], c['index_full_variant'])
"
}
run default
run nobox GPK_NO_PART_BOX=1 
