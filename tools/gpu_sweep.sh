# A/B timing of library variants on the C2 join:  bash tools/gpu_sweep.sh <variant> ...   ("base" = the in-tree library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in "$@"; do
  if [ "$v" = base ]; then timeout 200 python tools/tile_time.py --tag base 2>&1 | tail -1
  else GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so timeout 200 python tools/tile_time.py --tag $v 2>&1 | tail -1; fi
done | tee $O/sweep_$(date +%H%M%S).log
