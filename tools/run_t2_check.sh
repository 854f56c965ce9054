cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_t2 or fused_batchnorm or gemm_rs" 2>&1 | tail -3 > gpurun_out/t2_check.txt
python tools/bench_gemm_t2.py 3588x728x728 3588x728x364 3588x300x600 2>&1 | grep -v amdgpu >> gpurun_out/t2_check.txt
python bench.py --full --no_cpu_baseline --no_configs45 > gpurun_out/bench_full_quick.json 2>/dev/null
