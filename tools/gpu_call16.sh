cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/idx_small_plain.py 2>&1 | tail -1
python tools/idx_small_time.py > gpurun_out/r03v_idx_small.log 2>&1
grep -a "build\|index build:" gpurun_out/r03v_idx_small.log | tail -32
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof -- python $GRAFT_REPO_ROOT/tools/idx_small_plain.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof.log 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof/*/*hip_api_stats.csv | head -1); head -22 $f | cut -c1-150
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof/*/*kernel_stats.csv | head -1); head -60 $f | cut -d, -f1-4 | cut -c1-110
tail -2 $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof.log
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03v_idxprof
