cd /root/repo; mkdir -p gpurun_out
timeout 900 python tools/bench_gemm_t2.py --sweep > gpurun_out/t2_sweep.txt 2>&1
tail -3 gpurun_out/t2_sweep.txt
