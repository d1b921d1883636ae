# two PMC passes (instruction mix, waits) over tools/tile_time.py:  bash tools/pmc_quick.sh <tag>   (GPK_FUSED_FORM / GPK_LIB_PATH from the environment)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/pmcq_$TAG; mkdir -p $OUT
CMD="python $R/tools/tile_time.py --steps 6"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  d=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
python - <<PY > $R/gpurun_out/pmcq_$TAG.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pip_tile" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; v=v[len(v)//3:]
    print(f"{k:40s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
echo "== $TAG"; cat $R/gpurun_out/pmcq_$TAG.txt
rm -rf $OUT
