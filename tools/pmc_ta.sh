cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc7ta; mkdir -p $OUT
i=0
for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/base/p$i -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/base.p$i.log 2>&1 || tail -3 $OUT/base.p$i.log
done
