cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_rs or linear_mfma" > gpurun_out/r03/rs_test.log 2>&1; echo "rs test rc=$?"; tail -3 gpurun_out/r03/rs_test.log
timeout 300 python tools/bench_gemm_rs.py 3588x300x300 3588x600x300 3588x300x600 3588x128x300 3588x300x128 3588x728x728 2>&1 | grep "M="
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do echo "default $(run X=1)"; done
