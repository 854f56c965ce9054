cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
MSDE_DP_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no_cpu_baseline --no_configs45 > gpurun_out/r03/bench_dp2_gloo.json 2> gpurun_out/r03/bench_dp2_gloo.err; echo "dp2 rc=$?"; tail -3 gpurun_out/r03/bench_dp2_gloo.err; cut -c1-300 gpurun_out/r03/bench_dp2_gloo.json
