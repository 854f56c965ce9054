cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_rs or linear_mfma" > gpurun_out/r03/rs_test.log 2>&1; echo "rs test rc=$?"; tail -25 gpurun_out/r03/rs_test.log
timeout 600 python tools/bench_gemm_rs.py > gpurun_out/r03/rs_bench.log 2>&1; cat gpurun_out/r03/rs_bench.log
