# round 4, call 1: the fused point join — parity, then A/B timing against the round-3 library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chains.py -m gpu -x -q ) > $O/c1_tests.log 2>&1; tail -15 $O/c1_tests.log
for v in r03; do GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so timeout 200 python tools/tile_time.py --tag $v 2>&1 | tail -1; done | tee $O/c1_time.log
timeout 200 python tools/tile_time.py --tag fused 2>&1 | tail -1 | tee -a $O/c1_time.log
GPK_TILE_KERNEL=route timeout 200 python tools/tile_time.py --tag new_route 2>&1 | tail -1 | tee -a $O/c1_time.log
( timeout 500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c2_full" ) > $O/c1_c2full.log 2>&1; tail -3 $O/c1_c2full.log
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-600 | tee $O/c1_bench.log
