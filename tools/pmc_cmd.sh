# PMC passes over any command:  bash tools/pmc_cmd.sh <tag> <kernel-substring> <command ...>    -> gpurun_out/pmcc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; KSUB=$2; shift 2; CMD="$@"; OUT=$R/gpurun_out/pmcc_$TAG; mkdir -p $OUT
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum"; do
  d=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
python - <<PY > $R/gpurun_out/pmcc_$TAG.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v=agg[k]; v=v[len(v)//3:]
    print(f"{k:40s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
echo "== $TAG"; cat $R/gpurun_out/pmcc_$TAG.txt
rm -rf $OUT
