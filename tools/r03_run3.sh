cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_rs or linear_mfma" > gpurun_out/r03/rs_test.log 2>&1; echo "rs test rc=$?"; tail -12 gpurun_out/r03/rs_test.log
timeout 600 python tools/bench_gemm_rs.py > gpurun_out/r03/rs_bench.log 2>&1; cat gpurun_out/r03/rs_bench.log
for rt in 1 2; do for t in 3 5; do
  echo "== RT=$rt T=$t"
  MSDE_RS_RT=$rt MSDE_RS_T=$t timeout 300 python tools/bench_gemm_rs.py 3588x300x300 3588x600x300 3588x300x600 3588x128x300 3588x300x128 2>&1 | grep "M="
done; done | tee gpurun_out/r03/rs_sweep.log
