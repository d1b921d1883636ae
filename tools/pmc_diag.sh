# Diagnosis: where do the waves of gpk_pip_tile wait?  Separate PMC passes (no trace domains next to --pmc).
#   bash tools/pmc_diag.sh <tag>   -> gpurun_out/diag_<tag>/summary.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/diag_$1; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 20000"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALL_BY_TC_CYCLES_sum TA_DATA_STALL_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python $R/tools/pmc_extract.py $OUT pip_tile $OUT/summary.csv
python - <<PY
import csv, collections
a = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/summary.csv")): a[r["counter"]].append(float(r["value"]))
for k, v in a.items(): print(f"{k:45s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
