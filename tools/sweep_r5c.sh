R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for f in pool wave chunked; do
for rep in 1 2; do GPK_FUSED_FORM=$f timeout 200 python tools/tile_time.py --tag $f "$@" 2>&1 | tail -1; done
done
for v in geopolars_amd/variants/*.so; do
  for f in pool wave; do
  GPK_FUSED_FORM=$f GPK_LIB_PATH=$R/$v timeout 200 python tools/tile_time.py --tag $f-$(basename $v .so) "$@" 2>&1 | tail -1
  done
done
