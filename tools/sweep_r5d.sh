R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do timeout 100 python tools/tile_time.py --tag pool 2>&1 | tail -1; done
for v in geopolars_amd/variants/*.so; do
  GPK_LIB_PATH=$R/$v timeout 100 python tools/tile_time.py --tag $(basename $v .so) 2>&1 | tail -1
done
timeout 100 python tools/tile_time.py --tag pool 2>&1 | tail -1
