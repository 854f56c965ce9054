#!/bin/bash
# A/B on ONE box of a module-level boolean with the 3D->2D head on (bench.py --full): tools/ab_flag_full.sh <file> <NAME> [rounds]
cd $GRAFT_REPO_ROOT
F=$1; N=$2; R=${3:-3}
run() { python bench.py --full --no_cpu_baseline --no_configs45 --no_pipeline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$1', j['ms_per_step'], 'node mlp chain', j['roofline_dense_head_node_mlp'].get('us_per_chain'), j['roofline_dense_head_node_mlp'].get('frac'))"; }
for i in $(seq $R); do
  sed -i "s/^$N = False/$N = True/" $F; run "$N=True "
  sed -i "s/^$N = True/$N = False/" $F; run "$N=False"
done
