cd /root/repo; mkdir -p gpurun_out
run() { env "$@" python bench.py --no_cpu_baseline --no_configs45 --no_pipeline --steps 300 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2; do
  echo "default $(run X=1)"
  for w in 256 384 768; do echo "CFFWD=$w $(run MSDE_SIDE_CFFWD_WGS=$w)"; done
  for w in 128 224 256; do echo "CFBWD=$w $(run MSDE_SIDE_CFBWD_WGS=$w)"; done
done
