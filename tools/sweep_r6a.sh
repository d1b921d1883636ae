R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in 10000000 5000000 2500000 1250000; do
  for p in 8 4 2 1; do
    GPK_FLOW_P=$p timeout 100 python tools/tile_time.py --points $n --tag "n$n-P$p" 2>&1 | tail -1 | cut -c1-110
  done
  GPK_FUSED_FORM=pool timeout 100 python tools/tile_time.py --points $n --tag "n$n-pool" 2>&1 | tail -1 | cut -c1-110
done
