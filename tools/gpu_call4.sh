R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chains.py tests/test_gpu_join.py -m gpu -x -q ) > $O/c4_tests.log 2>&1; tail -15 $O/c4_tests.log
bash tools/gpu_sweep.sh base subchains
