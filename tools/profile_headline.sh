# Collects what profiles/ holds for the headline kernel: rocprofv3 kernel stats of the default bench command and the
# separate PMC passes for HBM-side traffic (MI355X_MICROARCH.md: counters in their own runs).  Run on the GPU box:
#   bash tools/profile_headline.sh <tag>     -> gpurun_out/prof_<tag>/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$1; mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
grep -h '"metric"' $OUT/stats.log | cut -c1-260
