# PMC passes over tools/tile_time.py for a library variant:  bash tools/pmc_fused.sh <tag> [variant.so]   -> gpurun_out/pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; V=$2; OUT=$R/gpurun_out/pmcf_$TAG; mkdir -p $OUT
[ -n "$V" ] && export GPK_LIB_PATH=$R/geopolars_amd/variants/$V.so
CMD="python $R/tools/tile_time.py --steps 6"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  d=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$d -- $CMD > $OUT/pmc_$d.log 2>&1 || tail -2 $OUT/pmc_$d.log
done
python - <<PY > $R/gpurun_out/pmc_$TAG.txt
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "pip_tile" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): 
    v=agg[k]; v=v[len(v)//3:]  # drop warm-up launches
    print(f"{k:40s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
cat $R/gpurun_out/pmc_$TAG.txt
rm -rf $OUT
