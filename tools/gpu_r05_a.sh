#!/bin/bash
# round 5, visit A: the self-launching `bench.py --gpus 2` on ONE GPU (gloo smoke) + a same-box default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
MSDE_DP_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 3 --no_cpu_baseline --no_configs45 --no_bf16x3 > $O/bench_dp2_gloo_one_gpu.json 2> $O/bench_dp2_gloo_one_gpu.err; echo "dp2 rc=$?"; tail -3 $O/bench_dp2_gloo_one_gpu.err; cut -c1-400 $O/bench_dp2_gloo_one_gpu.json
timeout 600 python3 bench.py --gpus 2 --census_only; echo "census without gloo rc=$? (expected 2)"
timeout 900 python3 bench.py --no_bf16x3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
