R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 500 python -m pytest tests/test_gpu_chains.py -q ) > $O/r03h_chains.log 2>&1; tail -5 $O/r03h_chains.log
bash tools/sweep_tile.sh > $O/r03h_sweep.log 2>&1; cat $O/r03h_sweep.log
