#!/bin/bash
# Sweep of a module-level constant on ONE box, alternating over the values: tools/ab_value.sh <file> <NAME> <rounds> <bench flags or -> v1 v2 ...
# (the line `NAME = <something>` of <file> is rewritten in the box's copy; each run prints ms/step and the side / main chain ends)
cd $GRAFT_REPO_ROOT
F=$1; N=$2; R=$3; FLAGS=$4; shift 4
[ "$FLAGS" = "-" ] && FLAGS=""
run() { python bench.py $FLAGS --no_cpu_baseline --no_configs45 ${NOPIPE---no_pipeline} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$N = $1', j['ms_per_step'], 'tail_us', j.get('tail_us'), 'after both', j.get('tail_after_both_streams_us'))"; }
for i in $(seq $R); do
  for v in "$@"; do
    sed -i -E "s/^$N = [^ #]+/$N = $v/" $F; run "$v"
  done
done
