# rates of the streaming reductions for the in-tree library and the variants named in $VARIANTS (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in base $VARIANTS; do
  if [ $v = base ]; then unset GPK_LIB_PATH; else export GPK_LIB_PATH=geopolars_amd/variants/$v.so; fi
  timeout 600 python tools/bench_ops.py --ops ${OPS:-area,euclidean_length} 2>/dev/null > gpurun_out/ops_$v.jsonl
  python - $v <<'PY'
import json, sys
for l in open("gpurun_out/ops_%s.jsonl" % sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print("%-8s %-18s %-38s %.4f ms  %.0f GB/s  %s" % (sys.argv[1], d["op"], d["workload"][:38], d["ms"], d["GBps"], {k[4:]: round(v, 4) for k, v in d["kernels_ms"].items() if k != "wall_ms_per_call"}))
PY
done
