# Tuning helper: polygon x polygon refine time for every variant library given ("base" = in-tree build).
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset GPK_LIB_PATH; else export GPK_LIB_PATH=$R/geopolars_amd/variants/$v.so; fi
  echo $v $(timeout 200 python tests/perf_configs.py --only c4 2>&1 | grep -o '"gpk_pair_refine": [0-9.]*\|"pairs": [0-9]*\|"parity": [a-z]*')
done
