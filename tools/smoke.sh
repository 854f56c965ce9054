#!/bin/bash
mkdir -p gpurun_out/tests
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/tests/smoke.log 2>&1; tail -n 5 gpurun_out/tests/smoke.log
