cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mb in default 0; do
  if [ $mb = 0 ]; then export GPK_DEVICE_CACHE_MB=0; fi
  timeout 400 python tools/idx_build_time.py 5000000 > gpurun_out/r03h_idx_$mb.log 2>&1
  echo "== cache $mb"; grep -a "== build\|temporaries\|hipMalloc " gpurun_out/r03h_idx_$mb.log | head -12
done
