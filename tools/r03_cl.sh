#!/bin/bash
mkdir -p gpurun_out/r03cl
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03cl/t.log 2>&1; tail -n 3 gpurun_out/r03cl/t.log
bash tools/ab.sh MSDE_CL_ON_SIDE 1 0 3 2>&1 | tee gpurun_out/r03cl/ab.log
