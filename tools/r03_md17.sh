#!/bin/bash
mkdir -p gpurun_out/r03m
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -s -k "md17" > gpurun_out/r03m/t1.log 2>&1
timeout 600 python - > gpurun_out/r03m/c5.log 2>&1 <<'PY'
import json, torch, bench
print(json.dumps(bench.config5_md17(torch.device("cuda", 0))))
PY
tail -n 12 gpurun_out/r03m/t1.log | cut -c1-400; tail -n 3 gpurun_out/r03m/c5.log | cut -c1-900
