R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests/test_gpu_join.py tests/test_gpu_contains.py tests/test_gpu_configs.py -q -x ) > $O/r03l_tests.log 2>&1; tail -4 $O/r03l_tests.log
for c in c3 c4 c5; do timeout 400 python bench.py --config $c > $O/r03l_bench_$c.log 2>&1; grep -h '"metric"' $O/r03l_bench_$c.log | cut -c1-200; done
# SQ_INSTS_VALU for C4's refine (roofline.valu)
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/prof_r03l_c4_sq -- python $R/bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline --parity-rows 20000 > $O/prof_r03l_c4_sq.log 2>&1 || tail -2 $O/prof_r03l_c4_sq.log
