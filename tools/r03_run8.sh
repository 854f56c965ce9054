cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dp.py > gpurun_out/r03/gputest.log 2>&1; echo "gputest rc=$?"; tail -15 gpurun_out/r03/gputest.log
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do
  echo "default              $(run X=1)"
  echo "MSDE_FUSE_GIN_APPLY=0 $(run MSDE_FUSE_GIN_APPLY=0)"
done | tee gpurun_out/r03/ab6.log
