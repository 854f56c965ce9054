cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
run() { env "$@" python bench.py --no_cpu_baseline --steps 60 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for i in 1 2; do
  echo "default            $(run X=1)"
  echo "FWD192 BWD160      $(run MSDE_SIDE_CFFWD_WGS=192 MSDE_SIDE_CFBWD_WGS=160)"
  echo "FWD128 BWD128      $(run MSDE_SIDE_CFFWD_WGS=128 MSDE_SIDE_CFBWD_WGS=128)"
  echo "FWD128 BWD96       $(run MSDE_SIDE_CFFWD_WGS=128 MSDE_SIDE_CFBWD_WGS=96)"
  echo "FWD384 BWD192      $(run MSDE_SIDE_CFFWD_WGS=384 MSDE_SIDE_CFBWD_WGS=192)"
done | tee gpurun_out/r03/ab4.log
