#!/bin/bash
mkdir -p gpurun_out/r03a
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 tools/probes/dp_debug_worker.py $(pwd) > gpurun_out/r03a/dpdbg.log 2>&1
grep "^rank 0 it 0" gpurun_out/r03a/dpdbg.log | grep "parameters with" | cut -c1-3000
