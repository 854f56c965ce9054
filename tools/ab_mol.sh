#!/bin/bash
# A/B on ONE box: the headline bench with the score network under autograd on the one-workgroup-per-molecule kernels (A:
# --score_kernel mol) and operator by operator (B: the default), alternating.
cd $GRAFT_REPO_ROOT
run() { python bench.py --no_cpu_baseline --no_configs45 --no_pipeline --score_kernel $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$1', j['ms_per_step'], 'tail', j['tail_us'], 'kernels', j['kernels_per_step'])"; }
for i in 1 2 3; do
  run A_mol mol
  run B_ops ops
done
