# One GPU-box call that produces the round's evidence: GPU tests, the four bench lines, the headline profile.
#   bash tools/gpu_round.sh <tag>     (run through gpurun; everything lands under gpurun_out/)
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/${TAG}_gputests.log 2>&1; tail -3 $O/${TAG}_gputests.log
for c in c2 c3 c4 c5; do
  timeout 400 python bench.py --config $c > $O/${TAG}_bench_$c.log 2>&1; grep -h '"metric"' $O/${TAG}_bench_$c.log | cut -c1-400 || tail -5 $O/${TAG}_bench_$c.log
done
bash tools/profile_headline.sh $TAG
