# A/B of an environment switch on the headline bench, alternating runs on ONE box.  usage: tools/ab.sh VAR A B [rounds] [bench args]
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; R=${4:-3}; shift 4
for i in $(seq 1 $R); do
  for v in $A $B; do
    ms=$(env $VAR=$v python bench.py --no_cpu_baseline --no_configs45 --no_bf16x3 --no_pipeline --steps 300 "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$VAR=$v $ms"
  done
done
