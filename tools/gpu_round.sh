# One GPU-box call that produces the round's evidence: GPU tests, the four bench lines, the headline profile (kernel stats +
# PMC passes + calibrated traffic record), kernel stats of the other three configurations and of the operator bench.
#   bash tools/gpu_round.sh <tag>     (run through gpurun; everything lands under gpurun_out/, profiles/<tag>_pmc_traffic.json is
#   written by tools/profile_headline.sh and copied to gpurun_out/ as well)
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/${TAG}_gputests.log 2>&1; tail -3 $O/${TAG}_gputests.log
( time timeout 600 python tests/stress_sweep.py ) > $O/${TAG}_stress_sweep.log 2>&1; tail -2 $O/${TAG}_stress_sweep.log
bash tools/profile_headline.sh $TAG
bash tools/profile_configs.sh $TAG; cd $R   # traffic + instruction counts of the dominant kernels of C3 / C4 / C5 -> profiles/<tag>_pmc_traffic_configs.json
for c in c2 c3 c4 c5; do
  timeout 400 python bench.py --config $c > $O/${TAG}_bench_$c.log 2>&1; grep -h '"metric"' $O/${TAG}_bench_$c.log | cut -c1-300 || tail -5 $O/${TAG}_bench_$c.log
done
for c in c3 c4 c5; do
  bash tools/prof_stats.sh ${TAG}_$c python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --parity-rows 20000
done
bash tools/prof_stats.sh ${TAG}_ops python $R/tools/bench_ops.py; grep -h '"op"' $O/${TAG}_ops.log > $O/${TAG}_ops.jsonl
# (round 6) the reductions the one-pass form took over, with every column kept on the two-stage form: the same box, back to back
GPK_RING_STREAM=0 timeout 300 python tools/bench_ops.py --ops area,euclidean_length 2>/dev/null | grep -h '"op"' > $O/${TAG}_ops_twostage.jsonl
python tools/bench_reference_benches.py > $O/${TAG}_reference_benches.jsonl 2>/dev/null
# round 6: the headline kernel over column lengths (a shard of the strong-scaled problem) and its launch timeline (a GPK_TILE_TRACE build)
for n in 10000000 5000000 2500000 1250000 625000; do timeout 100 python tools/tile_time.py --points $n --tag "rows $n" 2>&1 | tail -1; done > $O/${TAG}_sizes.txt 2>&1; cat $O/${TAG}_sizes.txt
for v in trace1 trace2; do
  [ -e $R/geopolars_amd/variants/$v.so ] && GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so timeout 100 python tools/flow_trace.py 2>&1 | tail -16 > $O/${TAG}_timeline_$v.txt
done
bash tools/pmc_mem.sh ${TAG} > /dev/null 2>&1; cp $O/pmcm_${TAG}.txt $O/${TAG}_c2_pmc_counters.txt 2>/dev/null
