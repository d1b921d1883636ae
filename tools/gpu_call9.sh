R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fixtures.py tests/test_gpu_properties.py tests/test_gpu_mixed_columns.py tests/test_gpu_join.py tests/test_gpu_edge_cases.py tests/test_gpu_structural.py -m gpu -x -q ) > $O/c9_tests.log 2>&1; tail -3 $O/c9_tests.log
timeout 500 python tools/bench_ops.py --ops bounds 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['op'], d['workload'][:24], round(d['ms'],3), round(d['GBps']), {k:round(v,3) for k,v in d['kernels_ms'].items()})"
