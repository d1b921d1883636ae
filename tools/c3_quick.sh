R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for lib in "" $R/geopolars_amd/variants/*.so; do
  GPK_BENCH_ABLATION=1 GPK_LIB_PATH=$lib timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 10 --parity-rows 50000 2>&1 | tail -1 > /tmp/l.json
  python -c "
import json; l=json.load(open('/tmp/l.json')); print('$lib', 'step', round(l['ms_per_step'],4), 'kernel', round(l['roofline']['launch_ms'],4), 'err', l['parity'].get('max_rel_err'))"
done
