#!/bin/bash
# A/B on ONE box of a module-level boolean: tools/ab_flag.sh <file> <NAME> [rounds]   (A = True, B = False, alternating)
cd $GRAFT_REPO_ROOT
F=$1; N=$2; R=${3:-3}
run() { python bench.py --no_cpu_baseline --no_configs45 --no_pipeline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$1', j['ms_per_step'], 'tail', j['tail_us'], 'kernels', j['kernels_per_step'], 'family frac', j['roofline'].get('frac'))"; }
for i in $(seq $R); do
  sed -i "s/^$N = False/$N = True/" $F; run "$N=True "
  sed -i "s/^$N = True/$N = False/" $F; run "$N=False"
done
