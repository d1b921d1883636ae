cd $GRAFT_REPO_ROOT
timeout 300 python tools/tile_time.py 2>&1 | tail -1
for v in wrdirect wpt16 wpt4; do GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/$v.so timeout 300 python tools/tile_time.py --tag $v 2>&1 | tail -1; done
