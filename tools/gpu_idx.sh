# tuning helper: index-build parity tests + per-kernel times of one C5 build + the C2 right side's build      bash tools/gpu_idx.sh   (through gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_configs.py tests/test_gpu_contains.py tests/test_gpu_edge_cases.py tests/test_gpu_chains.py tests/test_gpu_fused.py -m gpu -x -q ) > $O/idx_tests.log 2>&1; tail -3 $O/idx_tests.log
timeout 400 python tools/idx_profile.py 5000000 c5 2>&1 | grep -E "wall|sub_build|sum of" | tail -6
timeout 200 python tools/idx_small_plain.py 2>&1 | tail -4
timeout 300 python tools/idx_profile.py 200000 c3 2>&1 | grep -E "wall|sub_build|lrec_build"
