# A/B/C... of environment settings on the headline bench, alternating runs on ONE box.
# usage: tools/ab_multi.sh ROUNDS "ENV1=a ENV2=b" "ENV1=c" ...   (an empty string "" = defaults)
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq 1 $R); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --no_cpu_baseline --no_configs45 --no_pipeline --no_bf16x3 --steps 300 $BENCH_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$cfg] $ms"
  done
done
