cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03r
run() { # name, env...
  name=$1; shift
  env "$@" GPK_DEBUG_INDEX=1 timeout 400 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_c5_$name.log 2>&1
  grep -a "index build: \(list-cell\|level-2\|cells\)" gpurun_out/${T}_c5_$name.log | sort | uniq
  grep -a '^{' gpurun_out/${T}_c5_$name.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('$name', 'step', round(d['ms_per_step'],3), 'join', round(c['join_ms_per_step'],3), 'idx ms', round(c['index_build_ms'],1), 'idx GB', c['index_bytes']/1e9, 'tile', c['kernel_ms_per_step']['gpk_pip_tile'], d['parity']['bit_exact'])
"
}
run R2048 GPK_PIP_RMAX=2048
run R2048_nolrec GPK_PIP_RMAX=2048 GPK_NO_LIST_RECORDS=1
run R4096_nolrec GPK_NO_LIST_RECORDS=1
