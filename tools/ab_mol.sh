#!/bin/bash
# A/B on ONE box: the headline bench with the score network on the one-workgroup-per-molecule kernels (A) and on the operator
# path (B), alternating A B A B.
cd $GRAFT_REPO_ROOT
F=moleculesde_amd/geom3d/sde_2d_to_3d.py
run() { python bench.py --no_cpu_baseline --no_configs45 --no_bf16x3 --no_pipeline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$1', j['ms_per_step'], 'tail', j['tail_us'], 'kernels', j['kernels_per_step'], 'fwd', j['roofline_forward_schnet_sde2d3d'].get('ms'))"; }
for i in 1 2 3 4; do
  sed -i 's/^MOL_KERNEL_TRAIN = False/MOL_KERNEL_TRAIN = True/' $F; run A_mol
  sed -i 's/^MOL_KERNEL_TRAIN = True/MOL_KERNEL_TRAIN = False/' $F; run B_ops
done
