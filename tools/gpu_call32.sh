cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python tests/stress_sweep.py ) > gpurun_out/r03g_stress_sweep.log 2>&1
tail -25 gpurun_out/r03g_stress_sweep.log
