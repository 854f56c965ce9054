#!/bin/bash
mkdir -p gpurun_out/r03u
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -x > gpurun_out/r03u/t1.log 2>&1
tail -n 6 gpurun_out/r03u/t1.log | cut -c1-300
MSDE_FUSE_FRAME=1 MSDE_FUSE_PAIR_LINEAR=1 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/b_11.log 2>&1
MSDE_FUSE_FRAME=0 MSDE_FUSE_PAIR_LINEAR=0 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/b_00.log 2>&1
MSDE_FUSE_FRAME=1 MSDE_FUSE_PAIR_LINEAR=0 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/b_10.log 2>&1
MSDE_FUSE_FRAME=1 MSDE_FUSE_PAIR_LINEAR=1 timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03u/b_11b.log 2>&1
for f in b_11 b_00 b_10 b_11b; do python3 - $f <<'PY'
import json,sys
l=[x for x in open('gpurun_out/r03u/'+sys.argv[1]+'.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline_forward_schnet_sde2d3d']['ms'])
else: print(sys.argv[1], 'no json')
PY
done
