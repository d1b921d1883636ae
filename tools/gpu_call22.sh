cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/ablate_c5.py 2>&1 | tail -1
for k in 7 1; do
  GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/abl$k.so timeout 300 python tools/ablate_c5.py 2>&1 | tail -1
done
( GPK_NO_LEAN=1 GPK_NO_CHAINS=1 timeout 900 python -m pytest tests -m gpu -x -q -k "(join or config or pip or contains or edge or propert) and not chain and not full_size" ) > gpurun_out/r03zb_tests_general.log 2>&1
grep -a "passed\|failed" gpurun_out/r03zb_tests_general.log | tail -3
