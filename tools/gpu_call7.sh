R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_lineal_ops.py -m gpu -x -q ) > $O/c7_tests.log 2>&1; grep -E "passed|failed|Error|error" $O/c7_tests.log | tail -8
timeout 600 python bench.py --config c5 --no-cpu-baseline --steps 5 --warmup 2 2>$O/c7_c5.err | grep '"metric"' > $O/c7_c5.log; python - <<PY
import json
d=json.loads(open("$O/c7_c5.log").read().strip().split("\n")[-1])
print("ms_per_step",d["ms_per_step"],"build",d["config"]["index_build_ms"],"again",d["config"]["index_build_again_ms"])
print(json.dumps(d["config"]["one_shot_right_partitioned"]))
PY
tail -3 $O/c7_c5.err
