cd $GRAFT_REPO_ROOT
python tools/ablate_c5.py 2>&1 | tail -1
GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/abl8.so timeout 300 python tools/ablate_c5.py 2>&1 | tail -1
