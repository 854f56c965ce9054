#!/bin/bash
# Round 6, first visit: the GPU suite, the default bench line (now with config2_full), the 2-rank gloo smoke line, the 1-rank RCCL line.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a
mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu > $O/gputest.log 2>&1; echo "tests rc=$?"; tail -n 5 $O/gputest.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
bash tools/dp2_gloo_smoke.sh r06a; echo "dp2 rc=$?"
timeout 600 python3 bench.py --debug_dp_path --no_cpu_baseline --no_configs45 > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err; echo "rccl 1 rank rc=$?"; cut -c1-300 $O/bench_rccl_1rank.json
tail -5 $O/bench_rccl.err
