cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/ablate_c5.py 2>&1 | tail -1
for k in 2 3 6 7 1 5; do
  GPK_LIB_PATH=$GRAFT_REPO_ROOT/geopolars_amd/variants/abl$k.so timeout 300 python tools/ablate_c5.py 2>&1 | tail -1
done
