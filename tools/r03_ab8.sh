cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no_cpu_baseline --no_configs45 --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
timeout 300 python -m pytest tests/test_gpu_models.py -q -x -k "hipgraph_step_matches_eager or bs256_forward_backward" 2>&1 | tail -2
for i in 1 2 3; do
  echo "SIDE_WGRAD=1 $(run MSDE_SIDE_WGRAD=1)"
  echo "SIDE_WGRAD=0 $(run MSDE_SIDE_WGRAD=0)"
done | tee gpurun_out/r03/ab8.log
echo "SIDE_WGRAD=1 wgs128 $(run MSDE_SIDE_WGRAD=1 MSDE_SIDE_WGRAD_WGS=128)"
