R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chains.py tests/test_gpu_join.py -m gpu -x -q ) > $O/c8_tests.log 2>&1; tail -12 $O/c8_tests.log
timeout 200 python tools/tile_time.py --tag ldshits 2>&1 | tail -1
GPK_FUSED_LDS=0 timeout 200 python tools/tile_time.py --tag staging 2>&1 | tail -1
