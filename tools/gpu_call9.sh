R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_wkb.py tests/test_gpu_join.py -m gpu -x -q -k "plain_rows or longer_than_its_row or thousands_of_vertices" ) > $O/c9_tests.log 2>&1; tail -30 $O/c9_tests.log
