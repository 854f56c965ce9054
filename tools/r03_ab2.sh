cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_rs" > gpurun_out/r03/rs_test.log 2>&1; echo "rs test rc=$?"; tail -3 gpurun_out/r03/rs_test.log
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2 3; do
  echo "default            $(run X=1)"
  echo "MSDE_FUSE_GIN=0    $(run MSDE_FUSE_GIN=0)"
done | tee gpurun_out/r03/ab2.log
