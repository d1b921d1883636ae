R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_lineal_ops.py tests/test_gpu_dist.py tests/test_gpu_structural.py -q -x ) > $O/r03j_tests.log 2>&1; tail -12 $O/r03j_tests.log
