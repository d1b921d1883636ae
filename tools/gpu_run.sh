R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_lineal_ops.py tests/test_gpu_join.py -x -q ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
