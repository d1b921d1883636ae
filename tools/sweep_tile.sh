# Tuning sweep of the C2 tile kernels (one GPU call): the in-tree build in its three modes (route / chain kernel on an index with
# chains, the queue kernel on an index without), then every variant library under geopolars_amd/variants/ (its name says which kernel
# it is for: r* route, c* chain).  usage: bash tools/sweep_tile.sh [extra args of tools/tile_time.py]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
unset GPK_LIB_PATH GPK_TILE_KERNEL GPK_NO_CHAINS
timeout 200 python tools/tile_time.py --tag route "$@" 2>&1 | tail -1
GPK_TILE_KERNEL=chain timeout 200 python tools/tile_time.py --tag chain "$@" 2>&1 | tail -1
GPK_NO_CHAINS=1 timeout 200 python tools/tile_time.py --tag queue "$@" 2>&1 | tail -1
for v in geopolars_amd/variants/*.so; do
  [ -f $v ] || continue
  b=$(basename $v .so)
  case $b in c*) k=chain;; *) k=route;; esac
  GPK_TILE_KERNEL=$k GPK_LIB_PATH=$R/$v timeout 200 python tools/tile_time.py --tag $b "$@" 2>&1 | tail -1
done
