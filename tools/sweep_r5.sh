# Round-5 tuning sweep of the C2 point join (one GPU call): the fused tests, then HIP-event times of the in-tree build in its forms
# (chunked = default, GPK_FUSED_FORM=wave = round 4) and of every variant library under geopolars_amd/variants/.
#   bash tools/sweep_r5.sh [tests]      ("tests": run the fused / join parity tests first)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
unset GPK_LIB_PATH GPK_TILE_KERNEL GPK_NO_CHAINS GPK_FUSED_FORM
if [ "$1" = "tests" ]; then
  ( timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_join.py tests/test_gpu_chains.py -q ) > $O/r5_fused_tests.log 2>&1; tail -5 $O/r5_fused_tests.log
  shift
fi
for rep in 1 2; do
timeout 200 python tools/tile_time.py --tag chunked "$@" 2>&1 | tail -1
GPK_FUSED_FORM=wave timeout 200 python tools/tile_time.py --tag wave "$@" 2>&1 | tail -1
done
for v in geopolars_amd/variants/*.so; do
  [ -f $v ] || continue
  GPK_LIB_PATH=$R/$v timeout 200 python tools/tile_time.py --tag $(basename $v .so) "$@" 2>&1 | tail -1
done
for n in 512 2097152 4194304 20000000; do
timeout 200 python tools/tile_time.py --tag chunked-$n --points $n 2>&1 | tail -1
[ $n -lt 10000000 ] && GPK_FUSED_FORM=wave timeout 200 python tools/tile_time.py --tag wave-$n --points $n 2>&1 | tail -1
done
