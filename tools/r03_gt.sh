#!/bin/bash
mkdir -p gpurun_out/r03g
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gin" > gpurun_out/r03g/t1.log 2>&1
tail -n 4 gpurun_out/r03g/t1.log | cut -c1-300
for i in 1 2; do
for v in 1 0; do
MSDE_GT_REGS=$v timeout 400 python bench.py --no_configs45 --no_cpu_baseline --no_pipeline --steps 300 > gpurun_out/r03g/d${v}_$i.log 2>&1
done; done
for v in 1 0; do
MSDE_GT_REGS=$v timeout 400 python bench.py --full --no_configs45 --no_cpu_baseline --no_pipeline --steps 200 > gpurun_out/r03g/f${v}.log 2>&1
done
for f in d1_1 d0_1 d1_2 d0_2 f1 f0; do python3 - $f <<'PY'
import json,sys
l=[x for x in open('gpurun_out/r03g/'+sys.argv[1]+'.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[1], d['ms_per_step'], d['config']['stream']['ms_per_step_4_resident_batches_own_graphs'])
else: print(sys.argv[1], 'no json')
PY
done
