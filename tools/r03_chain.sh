#!/bin/bash
mkdir -p gpurun_out/r03c
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_chain" > gpurun_out/r03c/t1.log 2>&1
timeout 300 python tools/bench_gemm_rs.py chain > gpurun_out/r03c/mb8.log 2>&1
MSDE_CHAIN_WAVES=4 timeout 300 python tools/bench_gemm_rs.py chain > gpurun_out/r03c/mb4.log 2>&1
timeout 600 python -m pytest tests/test_gpu_models.py -q -x > gpurun_out/r03c/t2.log 2>&1
MSDE_SCHNET_CHAIN=1 timeout 400 python bench.py --no_configs45 --steps 300 > gpurun_out/r03c/b_chain.log 2>&1
MSDE_SCHNET_CHAIN=0 timeout 400 python bench.py --no_configs45 --steps 300 > gpurun_out/r03c/b_nochain.log 2>&1
tail -n 3 gpurun_out/r03c/t1.log gpurun_out/r03c/t2.log; cat gpurun_out/r03c/mb8.log gpurun_out/r03c/mb4.log
