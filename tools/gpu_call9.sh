R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chains.py tests/test_gpu_join.py tests/test_gpu_configs.py tests/test_gpu_contains.py tests/test_gpu_edge_cases.py tests/test_gpu_properties.py tests/test_gpu_fixtures.py -m gpu -x -q ) > $O/c9_tests.log 2>&1; tail -5 $O/c9_tests.log
timeout 400 python tools/idx_profile.py 5000000 c5 2>&1 | grep -E "wall|sub_build|cell_|gpk_scan|sum of" | tail -14
timeout 200 python tools/idx_small_time.py 2>&1 | grep -E "==" | tail -2
timeout 300 python tools/idx_profile.py 200000 c3 2>&1 | grep -E "wall|sub_build|lrec_build|cell_count"
