cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dp.py > gpurun_out/r03/gputest.log 2>&1; echo "gputest rc=$?"; tail -30 gpurun_out/r03/gputest.log
timeout 600 python bench.py --no_cpu_baseline > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r03/bench1.json; tail -3 gpurun_out/r03/bench1.err
