R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 500 python -m pytest tests/test_gpu_chains.py -x -q ) > $O/r03b_chains.log 2>&1; tail -15 $O/r03b_chains.log
bash tools/sweep_tile.sh > $O/r03b_sweep.log 2>&1; cat $O/r03b_sweep.log
( time timeout 700 python -m pytest tests -m gpu -q -x ) > $O/r03b_gputests.log 2>&1; tail -8 $O/r03b_gputests.log
timeout 300 python bench.py > $O/r03b_bench_c2.log 2>&1; tail -1 $O/r03b_bench_c2.log | cut -c1-300; grep -o '"roofline": {[^}]*}' $O/r03b_bench_c2.log
