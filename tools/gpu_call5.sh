R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/c5_gputests.log 2>&1; tail -6 $O/c5_gputests.log
timeout 300 python bench.py 2>&1 | grep '"metric"' > $O/c5_bench.log; cut -c1-1500 $O/c5_bench.log
