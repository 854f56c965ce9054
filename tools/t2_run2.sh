cd /root/repo; mkdir -p gpurun_out
timeout 600 python tools/t2_phases.py > gpurun_out/t2_phases.txt 2>&1; tail -3 gpurun_out/t2_phases.txt
