# rocprofv3 kernel stats of one command:  bash tools/prof_stats.sh <tag> <command ...>  -> gpurun_out/<tag>_kernel_stats.csv (+ <tag>.log)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -- "$@" > $R/gpurun_out/$TAG.log 2>&1
f=$(ls $R/gpurun_out/stats_$TAG/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/stats_$TAG
