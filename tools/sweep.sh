# Tuning helper: bench every variant library given on the command line ("base" = in-tree build, "nolean" / "noroute" = in-tree
# build with the general / the lean tile kernel forced).
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  unset GPK_LIB_PATH GPK_NO_LEAN GPK_NO_ROUTE
  if [ $v = nolean ]; then export GPK_NO_LEAN=1; elif [ $v = noroute ]; then export GPK_NO_ROUTE=1; elif [ $v != base ]; then export GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so; fi
  echo $v $(timeout 150 python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --parity-rows 20000 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|bench.py:.*')
done
unset GPK_LIB_PATH GPK_NO_LEAN GPK_NO_ROUTE
