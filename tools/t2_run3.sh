cd /root/repo; mkdir -p gpurun_out
timeout 300 python tools/t2_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/t2_diag.txt
timeout 300 python tools/t2_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/t2_phases.txt
timeout 900 python tools/bench_gemm_t2.py --sweep 2>&1 | grep -v amdgpu.ids > gpurun_out/t2_sweep.txt
