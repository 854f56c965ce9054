#!/bin/bash
mkdir -p gpurun_out/sweep
for rt in 1 2; do for t in 2 3 5; do
echo "RT=$rt T=$t"; MSDE_RS_RT=$rt MSDE_RS_T=$t timeout 120 python tools/bench_gemm_rs.py 3588x300x300 3588x600x300 3588x300x600 3588x128x300 3588x300x128 2>&1 | grep "M="
done; done > gpurun_out/sweep/sweep.log 2>&1
cat gpurun_out/sweep/sweep.log
