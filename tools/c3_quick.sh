R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 10 --parity-rows 50000 2>&1 | tail -1 > /tmp/l.json
  python -c "
import json; l=json.load(open('/tmp/l.json')); print('step', round(l['ms_per_step'],4), 'kernel', round(l['roofline']['launch_ms'],4), 'err', l['parity']['max_rel_err'], 'one-shot', l['config'].get('one_shot_ms'))"
done
