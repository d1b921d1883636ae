cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r02f; mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-join-stats --parity-rows 20000"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_FETCH_SIZE -- $CMD > $OUT/pmc_FETCH_SIZE.log 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_WRITE_SIZE -- $CMD > $OUT/pmc_WRITE_SIZE.log 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c4_WRITE_SIZE -- python $R/bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline --parity-rows 20000 > $OUT/c4_WRITE_SIZE.log 2>&1
grep -h '"metric"' $OUT/stats.log | cut -c1-200
