cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 420 python -m pytest tests/test_gpu_plan.py -q -x -k "oversized or two_live" > gpurun_out/r03/plan_test.log 2>&1; echo "plan test rc=$?"; tail -30 gpurun_out/r03/plan_test.log
