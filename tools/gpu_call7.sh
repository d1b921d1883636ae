R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_contains.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_properties.py tests/test_golden_ops.py -q -x ) > $O/r03k_tests.log 2>&1; tail -6 $O/r03k_tests.log
timeout 400 python bench.py --config c4 > $O/r03k_bench_c4.log 2>&1; tail -1 $O/r03k_bench_c4.log | cut -c1-400; grep -o '"kernel_ms": {[^}]*}' $O/r03k_bench_c4.log
