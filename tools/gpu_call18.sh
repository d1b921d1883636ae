cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03x
( timeout 900 python -m pytest tests -m gpu -x -q -k "join or chain or config or index or pip or assembly or contains or edge or mixed or propert or fixture" ) > gpurun_out/${T}_tests.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests.log | tail -3
( GPK_LIST_RECORDS=0 timeout 900 python -m pytest tests -m gpu -x -q -k "(join or chain or config or index or pip or assembly or contains or edge or mixed or propert or fixture) and not many_small_parts" ) > gpurun_out/${T}_tests_nolrec.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests_nolrec.log | tail -3
GPK_DEBUG_INDEX=1 timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5.log 2>&1
grep -a "index build: " gpurun_out/${T}_c5.log | sort | uniq | head -14
grep -a '^{' gpurun_out/${T}_c5.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('c5 step', round(d['ms_per_step'],3), 'idx ms', round(c['index_build_ms'],1), round(c['index_build_again_ms'],1), 'idx GB', c['index_bytes']/1e9, 'tile', c['kernel_ms_per_step']['gpk_pip_tile'], d['parity']['bit_exact'], c['exact_phase'], c['index_full_variant'])
"
