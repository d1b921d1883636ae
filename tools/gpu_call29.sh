cd $GRAFT_REPO_ROOT
GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/trace.so timeout 300 python tools/c5_stage_clocks.py 2>&1 | tail -7
