R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_wkb.py tests/test_gpu_fixtures.py -m gpu -x -q ) > $O/c9_tests.log 2>&1; tail -3 $O/c9_tests.log
timeout 500 python tools/wkb_time.py 2>/dev/null | grep -v "^$" | tail -12
