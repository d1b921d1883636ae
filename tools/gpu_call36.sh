cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c5_full_size" ) 2>&1 | grep -a "passed\|failed\|real\|Error" | tail -5
