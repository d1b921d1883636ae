# A/B of the in-tree library against geopolars_amd/variants/old.so on one box (alternating, 3 rounds), then the headline parity tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do
  timeout 100 python tools/tile_time.py --tag new 2>&1 | tail -1 | cut -c1-140
  GPK_LIB_PATH=geopolars_amd/variants/old.so timeout 100 python tools/tile_time.py --tag old 2>&1 | tail -1 | cut -c1-140
done
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_join.py tests/test_gpu_chains.py -m gpu -x -q 2>&1 | tail -3
