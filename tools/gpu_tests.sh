#!/bin/bash
mkdir -p gpurun_out/tests
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/tests/gputest.log 2>&1
tail -n 8 gpurun_out/tests/gputest.log | cut -c1-300
