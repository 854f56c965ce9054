#!/bin/bash
mkdir -p gpurun_out/r03u
for i in 1 2; do
MSDE_SCHNET_CHAIN=1 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/c1_$i.log 2>&1
MSDE_SCHNET_CHAIN=0 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/c0_$i.log 2>&1
done
for f in c1_1 c0_1 c1_2 c0_2; do python3 - $f <<'PY'
import json,sys
l=[x for x in open('gpurun_out/r03u/'+sys.argv[1]+'.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline_forward_schnet_sde2d3d']['ms'], d['config']['stream']['ms_per_step_4_resident_batches_own_graphs'])
else: print(sys.argv[1], 'no json')
PY
done
