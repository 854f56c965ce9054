cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_rs or linear_mfma" > gpurun_out/r03/rs_test.log 2>&1; echo "rs test rc=$?"; tail -12 gpurun_out/r03/rs_test.log
for nr in 0 1 0 1; do
  echo "== NO_ROTATE=$nr"
  MSDE_RS_NO_ROTATE=$nr timeout 300 python tools/bench_gemm_rs.py 3588x300x300 3588x600x300 3588x300x600 3588x128x300 3588x300x128 3588x728x728 2>&1 | grep "M="
done | tee gpurun_out/r03/rs_rot.log
