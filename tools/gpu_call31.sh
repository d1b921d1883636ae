cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python bench.py --config c2 > gpurun_out/r03f_bench_c2.log 2>&1; grep -h '"metric"' gpurun_out/r03f_bench_c2.log | cut -c1-250
