R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/r03i_gputests.log 2>&1; tail -5 $O/r03i_gputests.log
timeout 300 python bench.py > $O/r03i_bench_c2.log 2>&1; tail -1 $O/r03i_bench_c2.log | cut -c1-250; grep -o '"roofline": {[^}]*}' $O/r03i_bench_c2.log
bash tools/profile_headline.sh r03i
