R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_join.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py tests/test_golden_ops.py -q -x ) > $O/r03m_tests.log 2>&1; tail -5 $O/r03m_tests.log
bash tools/sweep_tile.sh > $O/r03m_sweep.log 2>&1; cat $O/r03m_sweep.log
