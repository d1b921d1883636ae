# HBM-side traffic counters and the instruction counts of the dominant kernels of C3 / C4 / C5 (run on the GPU box):
#   bash tools/profile_configs.sh <tag>   -> gpurun_out/prof_<tag>_cfg/{c3,c4,c5}_{FETCH_SIZE,WRITE_SIZE}/..., summarised by
#   tools/pmc_configs.py into profiles/<tag>_pmc_traffic_configs.json
# Separate PMC passes, no trace domains next to --pmc (MI355X_MICROARCH.md); short timeouts: a pass that hangs is cut.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/prof_${TAG}_cfg; mkdir -p $OUT
for c in c3 c4 c5; do
  for set in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU"; do
    d=$(echo $set | cut -d' ' -f1)
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${c}_$d -- python $R/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --parity-rows 20000 > $OUT/${c}_$d.log 2>&1 || tail -2 $OUT/${c}_$d.log
  done
done
python $R/tools/pmc_configs.py $OUT $R/profiles/${TAG}_pmc_traffic_configs.json; cp $R/profiles/${TAG}_pmc_traffic_configs.json $OUT/
