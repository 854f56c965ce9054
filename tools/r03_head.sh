#!/bin/bash
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -x -k "dense or head or full or three or 3Dto2D or losscurve" > gpurun_out/r03h/t1.log 2>&1; tail -n 3 gpurun_out/r03h/t1.log
timeout 600 python bench.py --full --no_cpu_baseline --no_configs45 > gpurun_out/r03h/f1.json 2> gpurun_out/r03h/f1.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03h/f1.json') if l.startswith('{')][-1])
print(d['ms_per_step'], {k:v for k,v in d['config']['stream'].items() if k.startswith('ms_')}); print(d['roofline_dense_head_node_mlp'])
PY
