R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_concat.py tests/test_gpu_dist.py -m gpu -x -q ) > $O/c6_tests.log 2>&1; tail -25 $O/c6_tests.log
