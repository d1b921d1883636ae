# the streaming reductions: parity tests of the operators, then their rates (one-pass form against the two-stage form, same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edge_cases.py tests/test_gpu_properties.py tests/test_gpu_fixtures.py tests/test_gpu_mixed_columns.py tests/test_gpu_structural.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python tools/bench_ops.py --ops ${OPS:-area,bounds,euclidean_length} 2>&1 | cut -c1-330 > gpurun_out/ops_stream.jsonl
GPK_RING_STREAM=0 timeout 600 python tools/bench_ops.py --ops ${OPS:-area,bounds,euclidean_length} 2>&1 | cut -c1-330 > gpurun_out/ops_twostage.jsonl
python - <<'PY'
import json
for f in ("gpurun_out/ops_stream.jsonl", "gpurun_out/ops_twostage.jsonl"):
    print(f)
    for l in open(f):
        try:
            d = json.loads(l[:l.rindex("}")+1]) if not l.rstrip().endswith("}") else json.loads(l)
        except Exception:
            print("   ", l[:200].rstrip()); continue
        print("   %-18s %-42s %.4f ms  %.0f GB/s  %s" % (d["op"], d["workload"][:42], d["ms"], d["GBps"], {k: round(v, 4) for k, v in d["kernels_ms"].items()}))
PY
