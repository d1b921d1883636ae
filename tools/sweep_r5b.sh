# wave-form variants (GPK_FUSED_FORM=wave for every library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export GPK_FUSED_FORM=wave
for rep in 1 2; do timeout 200 python tools/tile_time.py --tag wave 2>&1 | tail -1; done
for v in geopolars_amd/variants/*.so; do
  GPK_LIB_PATH=$R/$v timeout 200 python tools/tile_time.py --tag $(basename $v .so) "$@" 2>&1 | tail -1
done
