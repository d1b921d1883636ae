# Tuning helper: SQ instruction-mix counters for gpk_pip_tile (run on the GPU box).
#   bash tools/pmc_ablate.sh <out-subdir> [variant ...]   ("base" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  if [ $v = base ]; then unset GPK_LIB_PATH; else export GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so; fi
  timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/$v/p1 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$v.p1.log 2>&1
done
