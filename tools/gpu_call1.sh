# round-2 GPU call 1: full GPU test suite, the four bench configurations, the lean-kernel sweep, the headline profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json; tail -3 $O/bench_c2.err
bash tools/sweep.sh base nolean lean_p2 lean_p2q128 lean_p2w8 lean_p4w6 lean_p4w8 lean_p8 > $O/sweep.log 2>&1; cat $O/sweep.log
timeout 400 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 400 $O/bench_c3.json; tail -3 $O/bench_c3.err
timeout 500 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 400 $O/bench_c4.json; tail -3 $O/bench_c4.err
bash tools/profile_headline.sh r02a > $O/profile.log 2>&1; tail -5 $O/profile.log
( time timeout 900 python bench.py --config c5 ) > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 400 $O/bench_c5.json; tail -5 $O/bench_c5.err
