R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chains.py -m gpu -x -q ) > $O/c3_tests.log 2>&1; tail -5 $O/c3_tests.log
timeout 200 python tools/tile_time.py --tag pack4 2>&1 | tail -1
GPK_STAGE8=1 timeout 200 python tools/tile_time.py --tag stage8 2>&1 | tail -1
