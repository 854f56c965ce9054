#!/bin/bash
mkdir -p gpurun_out/r03p
timeout 600 python -m pytest tests/test_gpu_plan.py -q -x -k "pipeline or two_live" > gpurun_out/r03p/t1.log 2>&1
timeout 600 python bench.py --no_configs45 --no_cpu_baseline --steps 300 > gpurun_out/r03p/b1.log 2>&1
timeout 600 python bench.py --no_configs45 --no_cpu_baseline --steps 300 > gpurun_out/r03p/b2.log 2>&1
timeout 300 python tools/probes/aten_sources.py > gpurun_out/r03p/aten.log 2>&1
tail -n 5 gpurun_out/r03p/t1.log; cut -c1-300 gpurun_out/r03p/b1.log | tail -n 2
