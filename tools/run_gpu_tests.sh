cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/gpu_tests.txt
