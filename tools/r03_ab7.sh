cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2 3; do
  echo "default (lazy fwd only) $(run X=1)"
  echo "MSDE_FUSE_GIN_APPLY=0   $(run MSDE_FUSE_GIN_APPLY=0)"
done | tee gpurun_out/r03/ab7.log
