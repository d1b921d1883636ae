cd $GRAFT_REPO_ROOT
python tools/ablate_c5.py 2>&1 | tail -1
for v in b3 b2 b2p1; do
  GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/$v.so timeout 300 python tools/ablate_c5.py 2>&1 | tail -1
done
