cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do
  echo "default            $(run X=1)"
  echo "MSDE_FUSE_GIN=0    $(run MSDE_FUSE_GIN=0)"
  echo "FUSE_SCHNET_TAIL=0 $(run MSDE_FUSE_SCHNET_TAIL=0)"
  echo "MSDE_LINEAR=lib    $(run MSDE_LINEAR=lib MSDE_FUSE_GIN=0 MSDE_FUSE_SCHNET_TAIL=0)"
done | tee gpurun_out/r03/ab1.log
