#!/bin/bash
mkdir -p gpurun_out/r03u
timeout 300 python tools/probes/aten_sources.py > gpurun_out/r03u/aten.log 2>&1
grep -v "Warning\|warn\|amdgpu" gpurun_out/r03u/aten.log | cut -c1-200
