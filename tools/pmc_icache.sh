# instruction-cache and scalar-cache counters of a kernel:  bash tools/pmc_icache.sh <tag> <kernel-substring> <command ...>  -> gpurun_out/pmci_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; KSUB=$2; shift 2; CMD="$@"; OUT=$R/gpurun_out/pmci_$TAG; mkdir -p $OUT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  d=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
python - <<PY > $R/gpurun_out/pmci_$TAG.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; v=v[len(v)//3:]
    print(f"{k:40s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
echo "== $TAG"; cat $R/gpurun_out/pmci_$TAG.txt
rm -rf $OUT
