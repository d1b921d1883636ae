# Tuning helper: bench every variant library given on the command line ("base" = in-tree build).
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset GPK_LIB_PATH; else export GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so; fi
  echo $v $(timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}')
done
