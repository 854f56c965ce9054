cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do
  echo "default            $(run X=1)"
  echo "GEOMETRY_ON_SIDE   $(run MSDE_GEOMETRY_ON_SIDE=1)"
  echo "EARLY_WGRAD_FLUSH  $(run MSDE_EARLY_WGRAD_FLUSH=1)"
  echo "GEO+FLUSH          $(run MSDE_GEOMETRY_ON_SIDE=1 MSDE_EARLY_WGRAD_FLUSH=1)"
done | tee gpurun_out/r03/ab5.log
