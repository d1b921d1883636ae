# Collects what profiles/ holds for the headline kernel (run on the GPU box):
#   bash tools/profile_headline.sh <tag>     -> gpurun_out/prof_<tag>/...  and  profiles/<tag>_pmc_traffic.json
# 1. rocprofv3 kernel stats of the default bench command;
# 2. separate PMC passes (MI355X_MICROARCH.md: counters in their own runs, no trace domains next to --pmc) for HBM-side
#    traffic, L2 hits and the instruction mix of gpk_pip_tile;
# 3. the same FETCH_SIZE / WRITE_SIZE passes over tools/micro/tile_probe (known byte counts) to calibrate the counters;
# 4. tools/pmc_traffic.py folds all of it into one JSON that names the source hash it was measured at.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-join-stats --no-default-shape --parity-rows 20000"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES"; do
  d=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
hipcc --offload-arch=gfx950 -O3 $R/tools/micro/tile_probe.hip -o /tmp/tile_probe > $OUT/probe_build.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/probe_$set -- /tmp/tile_probe > $OUT/probe_$set.log 2>&1 || tail -2 $OUT/probe_$set.log
done
python $R/tools/pmc_traffic.py $OUT $R/profiles/${TAG}_pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; tail -3 $OUT/pmc_traffic.log
cp $R/profiles/${TAG}_pmc_traffic.json $OUT/ 2>/dev/null
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
grep -h '"metric"' $OUT/stats.log | cut -c1-300
