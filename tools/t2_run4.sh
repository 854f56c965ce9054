cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_t2 or fused_batchnorm_chain" 2>&1 | tail -8 > gpurun_out/t2_tests.txt
MSDE_T2=1 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t2_tests.txt
bash tools/ab.sh MSDE_T2 0 1 3 > gpurun_out/t2_ab.txt 2>&1
