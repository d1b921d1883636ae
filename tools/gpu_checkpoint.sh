# A light evidence checkpoint (GPU tests, the headline bench line, the headline profile):  bash tools/gpu_checkpoint.sh <tag>
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/${TAG}_gputests.log 2>&1; tail -3 $O/${TAG}_gputests.log
bash tools/profile_headline.sh $TAG; cd $R
timeout 400 python bench.py --config c2 > $O/${TAG}_bench_c2.log 2>&1; grep -h '"metric"' $O/${TAG}_bench_c2.log | cut -c1-300 || tail -5 $O/${TAG}_bench_c2.log
