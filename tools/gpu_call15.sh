cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r03u8
( timeout 900 python -m pytest tests -m gpu -x -q -k "(join or contains or config or propert or edge or mixed) and not full_size and not c5 and not c3 and not c2" ) > gpurun_out/${T}_tests.log 2>&1
grep -a "passed\|failed" gpurun_out/${T}_tests.log | tail -3
for v in staged; do
  if [ $v = nostage ]; then export GPK_NO_CAND_STAGE=1; fi
  timeout 400 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_c4_$v.log 2>&1
  grep -a '^{' gpurun_out/${T}_c4_$v.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print('$v', round(d['ms_per_step'],3), c['kernel_ms_per_step'], d['parity']['bit_exact'])
"
done
